// valu_peak2.hip -- issue cost of the VALU / LDS instructions the megakernel's node step, leaf step and shading block are made of,
// measured on the box: cycles per wave64 instruction per SIMD with 1 / 2 / 4 waves resident (round 3: the round-2 table had six
// opcodes and one of them, v_cndmask_b32, came out at 23 cycles -- VERDICT r02 asks whether that is real).
// Every opcode runs as 8 independent chains x 32 in an unrolled loop, no memory traffic; the clock is s_memtime-independent:
// cycles = wall time x the device's reported peak clock, so "2.4" means 2 cycles at the clock the chip really ran at.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak2.hip -o gpurun_out/valu_peak2 && gpurun_out/valu_peak2 > profiles/archive/r03_valu_peak.json
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { auto e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kChains = 8, kUnroll = 32;

// OP(index, name, asm text): %0 = the chain's own register (read + written), %1 = a 64-bit VGPR pair of the chain, %2 / %3 = two
// loop-invariant VGPRs, %4 = an SGPR pair holding a lane mask
#define OPS(X) \
    X(0, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3") \
    X(1, "v_mul_f32", "v_mul_f32 %0, %0, %2") \
    X(2, "v_add_f32", "v_add_f32 %0, %0, %2") \
    X(3, "v_fmac_f32", "v_fmac_f32 %0, %2, %3") \
    X(4, "v_max_f32", "v_max_f32 %0, %0, %2") \
    X(5, "v_min_f32", "v_min_f32 %0, %0, %2") \
    X(6, "v_max3_f32", "v_max3_f32 %0, %0, %2, %3") \
    X(7, "v_min3_f32", "v_min3_f32 %0, %0, %2, %3") \
    X(8, "v_med3_f32", "v_med3_f32 %0, %0, %2, %3") \
    X(9, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %2") \
    X(10, "v_cvt_f32_ubyte3", "v_cvt_f32_ubyte3 %0, %2") \
    X(11, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %2") \
    X(12, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %2") \
    X(13, "v_fma_mix_f32 (f16 lo x f32 + f32)", "v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]") \
    X(14, "v_fma_mix_f32 (f16 hi x f32 + f32)", "v_fma_mix_f32 %0, %2, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
    X(15, "v_cndmask_b32 vcc (vcc set once per trip)", "v_cndmask_b32 %0, %0, %2, vcc") \
    X(16, "v_cndmask_b32 sgpr mask (e64)", "v_cndmask_b32_e64 %0, %0, %2, %4") \
    X(17, "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 vcc, %0, %2") \
    X(18, "v_cmp_lt_f32 sgpr (e64)", "v_cmp_lt_f32_e64 s[40:41], %0, %2") \
    X(19, "v_cmp + v_cndmask pair", "v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc") \
    X(20, "v_and_b32", "v_and_b32 %0, %0, %2") \
    X(21, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3") \
    X(22, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %2") \
    X(23, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0") \
    X(24, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8") \
    X(25, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3") \
    X(26, "v_bfi_b32", "v_bfi_b32 %0, %2, %0, %3") \
    X(27, "v_min_u32", "v_min_u32 %0, %0, %2") \
    X(28, "v_max_u32", "v_max_u32 %0, %0, %2") \
    X(29, "v_min3_u32", "v_min3_u32 %0, %0, %2, %3") \
    X(30, "v_max3_u32", "v_max3_u32 %0, %0, %2, %3") \
    X(31, "v_add_u32", "v_add_u32 %0, %0, %2") \
    X(32, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %2") \
    X(33, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %3") \
    X(34, "v_mad_u64_u32", "v_mad_u64_u32 %1, s[40:41], %2, %3, %1") \
    X(35, "v_lshl_add_u64", "v_lshl_add_u64 %1, %1, 2, %1") \
    X(36, "v_mov_b32", "v_mov_b32 %0, %2") \
    X(37, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(38, "v_pk_fma_f32", "v_pk_fma_f32 %1, %1, %1, %1") \
    X(39, "v_pk_mul_f32", "v_pk_mul_f32 %1, %1, %1") \
    X(40, "v_pk_add_f32", "v_pk_add_f32 %1, %1, %1") \
    X(41, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %2, %3") \
    X(42, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %2") \
    X(43, "v_pk_min_f16", "v_pk_min_f16 %0, %0, %2") \
    X(44, "v_pk_mul_f16", "v_pk_mul_f16 %0, %0, %2") \
    X(45, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %2") \
    X(46, "v_rcp_f32", "v_rcp_f32 %0, %0") \
    X(47, "v_rsq_f32", "v_rsq_f32 %0, %0") \
    X(48, "v_sqrt_f32", "v_sqrt_f32 %0, %0") \
    X(49, "v_exp_f32", "v_exp_f32 %0, %0") \
    X(50, "v_log_f32", "v_log_f32 %0, %0") \
    X(51, "v_sin_f32", "v_sin_f32 %0, %0") \
    X(52, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %0") \
    X(53, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %0, %2") \
    X(54, "v_readfirstlane_b32", "v_readfirstlane_b32 s42, %0") \
    X(55, "v_sub_f32", "v_sub_f32 %0, %0, %2") \
    X(56, "v_xor_b32", "v_xor_b32 %0, %0, %2") \
    X(57, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %2") \
    X(58, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %2") \
    X(59, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %2, 7") \
    X(60, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %2, %0") \
    X(61, "v_cmp_eq_u32 sgpr (e64)", "v_cmp_eq_u32_e64 s[40:41], %0, %2") \
    X(62, "v_cmp_class_f32", "v_cmp_class_f32 vcc, %0, %2") \
    X(63, "v_dot2c_f32_f16", "v_dot2c_f32_f16 %0, %2, %3") \
    X(64, "v_cmp + 2 x v_cndmask (per instruction)", "v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %0, %3, %0, vcc") \
    X(65, "s_mov vcc + v_cndmask vcc (per pair)", "s_mov_b64 vcc, %4\n\tv_cndmask_b32 %0, %0, %2, vcc") \
    X(66, "v_cmp (e64 -> sgpr) + v_cndmask e64 (per instruction)", "v_cmp_lt_f32_e64 s[40:41], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[40:41]") \
    X(67, "v_mul_f32 sdwa src0 BYTE_1 (is SDWA full rate?)", "v_mul_f32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(68, "v_or_b32", "v_or_b32 %0, %0, %2") \
    X(69, "v_cvt_f32_ubyte0 sdwa-free: v_and + v_or (per pair)", "v_and_b32 %0, %0, %2\n\tv_or_b32 %0, %0, %3") \
    X(70, "v_cmp e64 -> sgpr + 2 x v_cndmask e64 (per instruction)", "v_cmp_lt_f32_e64 s[40:41], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[40:41]\n\tv_cndmask_b32_e64 %0, %3, %0, s[40:41]") \
    X(71, "v_cmp vcc + v_cndmask + v_fma + v_cndmask (per instruction)", "v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_fma_f32 %0, %0, %2, %3\n\tv_cndmask_b32 %0, %3, %0, vcc") \
    X(72, "v_cmp vcc + 3 x v_cndmask (per instruction)", "v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %0, %3, %0, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc") \
    X(73, "v_cmp e64 -> sgpr + 3 x v_cndmask e64 (per instruction)", "v_cmp_lt_f32_e64 s[40:41], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[40:41]\n\tv_cndmask_b32_e64 %0, %3, %0, s[40:41]\n\tv_cndmask_b32_e64 %0, %0, %2, s[40:41]") \
    X(74, "v_cmp vcc + 4 x v_fma + 2 x v_cndmask (per instruction)", "v_cmp_lt_f32 vcc, %0, %2\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %0, %3, %0, vcc") \
    X(75, "v_cmp vcc + s_mov sgpr, vcc + 2 x v_cndmask e64 on the copy (per VALU instruction)", "v_cmp_lt_f32 vcc, %0, %2\n\ts_mov_b64 s[40:41], vcc\n\tv_cndmask_b32_e64 %0, %0, %3, s[40:41]\n\tv_cndmask_b32_e64 %0, %3, %0, s[40:41]") \
    X(76, "v_cndmask_b32_e64 with vcc as the explicit mask operand (vcc set once per trip)", "v_cndmask_b32_e64 %0, %0, %2, vcc") \
    X(77, "v_cndmask_b32 vcc + 3 x v_fma (vcc set once per trip; per instruction)", "v_cndmask_b32 %0, %0, %2, vcc\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %0, %0, %2, %3") \
    X(78, "v_cndmask_b32_e64 sgpr mask x 4 on one mask (per instruction)", "v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %0, %3, %0, %4\n\tv_cndmask_b32_e64 %0, %0, %3, %4\n\tv_cndmask_b32_e64 %0, %2, %0, %4")
constexpr int kOps = 79;
constexpr int instr_per_asm(int idx) {
    return idx == 19 || idx == 65 || idx == 66 || idx == 69 ? 2 : (idx == 64 || idx == 70 || idx == 75 ? 3 : (idx == 71 || idx == 72 || idx == 73 || idx == 77 || idx == 78 ? 4 : (idx == 74 ? 7 : 1)));
}
// (an asm statement that clobbers SGPRs makes the compiler put an s_nop behind it: only the opcodes that write one declare it)
constexpr bool writes_sgpr(int idx) { return idx == 17 || idx == 18 || idx == 19 || idx == 34 || idx == 54 || idx == 61 || idx == 62 || idx == 64 || idx == 65 || idx == 66 || (idx >= 70 && idx <= 75); }

template<int KIND>
__global__ __launch_bounds__(64) void valu_kernel(float *out, int trips, float seed) {
    float a[kChains];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[kChains];
    float x = seed * 0.999f + threadIdx.x * 1e-9f, y = seed * 1e-3f;
#pragma unroll
    for (int i = 0; i < kChains; i++) { a[i] = seed + i, p[i] = v2f{seed + i, seed - i}; }
    unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)trips;
    for (int t = 0; t < trips; t++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x), "v"(a[0]) : "vcc");
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
#pragma unroll
            for (int i = 0; i < kChains; i++) {
#define X(idx, name, text) if (KIND == idx) { \
    if (writes_sgpr(idx)) { asm volatile(text : "+v"(a[i]), "+v"(p[i]) : "v"(x), "v"(y), "s"(mask) : "vcc", "s40", "s41", "s42"); } \
    else { asm volatile(text : "+v"(a[i]), "+v"(p[i]) : "v"(x), "v"(y), "s"(mask)); } }
                OPS(X)
#undef X
            }
        }
    }
    float s = x + y;
#pragma unroll
    for (int i = 0; i < kChains; i++) { s += a[i] + p[i].x + p[i].y; }
    if (s == 12345.678f) { out[threadIdx.x] = s; }
}

// LDS throughput of the node step's access patterns: ds_read_b128 (conflict-free), ds_write_b32 / ds_read_b32 with a 1 KiB stride
// between entries (the traversal stack), ds_bpermute_b32
template<int KIND>
__global__ __launch_bounds__(64) void lds_kernel(float *out, int trips, float seed) {
    __shared__ float4 stage[256 * 4];
    const auto lane = threadIdx.x;
    stage[lane] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int idx = static_cast<int>(lane);
    unsigned v = lane;
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            if (KIND == 0) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f q;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(lane * 16u), "n"((k & 3) * 4096));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc.x += 0.f;
                asm volatile("" ::"v"(q));
            }
            if (KIND == 1) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lane * 4u), "v"(v), "n"((k & 15) * 1024) : "memory"); }
            if (KIND == 2) {
                unsigned q;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(q) : "v"(lane * 4u), "n"((k & 15) * 1024));
                asm volatile("" ::"v"(q));
            }
            if (KIND == 3) { asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(v) : "v"(idx * 4)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc.x + v == 12345.678f) { out[lane] = acc.x; }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const double clock_hz = prop.clockRate * 1e3;
    float *out;
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[kOps] = {
#define X(idx, name, text) name,
        OPS(X)
#undef X
    };
    void (*kernels[kOps])(float *, int, float) = {
#define X(idx, name, text) valu_kernel<idx>,
        OPS(X)
#undef X
    };
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"unit\": \"cycles per wave64 instruction per SIMD at the reported peak clock\", \"results\": [\n", prop.name, cus, clock_hz / 1e6);
    bool first = true;
    for (int kind = 0; kind < kOps; kind++) {
        std::printf("%s  {\"op\": \"%s\"", first ? "" : ",\n", names[kind]);
        first = false;
        for (int waves : {1, 2, 4}) {
            const int trips = 4000;
            const int blocks = simds * waves;
            hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(64), 0, 0, out, 50, 1.5f);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(64), 0, 0, out, trips, 1.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_wave = double(trips) * kChains * kUnroll * instr_per_asm(kind);
            const double cycles = ms * 1e-3 * clock_hz / (instr_per_wave * waves);
            std::printf(", \"w%d\": %.2f", waves, cycles);
        }
        std::printf("}");
    }
    const char *lds_names[] = {"ds_read_b128 (conflict-free)", "ds_write_b32 (stack push)", "ds_read_b32 (stack pop)", "ds_bpermute_b32"};
    void (*lds_kernels[])(float *, int, float) = {lds_kernel<0>, lds_kernel<1>, lds_kernel<2>, lds_kernel<3>};
    for (int kind = 0; kind < 4; kind++) {
        std::printf(",\n  {\"op\": \"%s\"", lds_names[kind]);
        for (int waves : {1, 2, 4}) {
            const int trips = 4000;
            const int blocks = simds * waves;
            hipLaunchKernelGGL(lds_kernels[kind], dim3(blocks), dim3(64), 0, 0, out, 50, 1.5f);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(lds_kernels[kind], dim3(blocks), dim3(64), 0, 0, out, trips, 1.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cycles = ms * 1e-3 * clock_hz / (double(trips) * 64 * waves);
            std::printf(", \"w%d\": %.2f", waves, cycles);
        }
        std::printf("}");
    }
    std::printf("\n]}\n");
    return 0;
}
