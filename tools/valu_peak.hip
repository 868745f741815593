// valu_peak.hip -- calibrates the VALU issue rate of a gfx950 SIMD on the box ("2 or 4 cycles per wave64 op?").
// Independent chains of v_fma_f32 / v_pk_fma_f32 / v_cvt_f32_ubyte0 / v_max_f32 / v_cndmask, 1 ... 8 waves per SIMD resident,
// no memory traffic.  Prints wave-level instructions per cycle per SIMD (clock from hipDeviceProp clockRate and, independently,
// from s_memtime / wall clock).   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { auto e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kChains = 8, kUnroll = 32;// 256 instructions per loop trip

template<int KIND>
__global__ __launch_bounds__(64) void valu_kernel(float *out, int trips, float seed) {
    float a[kChains];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[kChains];
    unsigned u = __float_as_uint(seed) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < kChains; i++) { a[i] = seed + i, p[i] = v2f{seed + i, seed - i}; }
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
#pragma unroll
            for (int i = 0; i < kChains; i++) {
                if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed)); }
                if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) % kChains])); }
                if (KIND == 2) { asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u)); }
                if (KIND == 3) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
                if (KIND == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed)); }
                if (KIND == 5) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(u)); }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChains; i++) { s += a[i] + p[i].x + p[i].y; }
    if (s == 12345.678f) { out[threadIdx.x] = s; }
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
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte0", "v_max_f32", "v_cndmask_b32", "v_add_u32"};
    void (*kernels[])(float *, int, float) = {valu_kernel<0>, valu_kernel<1>, valu_kernel<2>, valu_kernel<3>, valu_kernel<4>, valu_kernel<5>};
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f, \"results\": [\n", prop.name, cus, clock_hz / 1e6);
    bool first = true;
    for (int kind = 0; kind < 6; kind++) {
        for (int waves : {1, 2, 4, 8}) {
            const int trips = 20000;
            const int blocks = simds * waves;// one 64-lane block per wave; the dispatcher spreads them over the SIMDs
            hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(64), 0, 0, out, 100, 1.5f);// warm-up
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernels[kind], dim3(blocks), dim3(64), 0, 0, out, trips, 1.5f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double wave_instr = double(blocks) * trips * kChains * kUnroll;
            const double per_simd_per_s = wave_instr / (ms * 1e-3) / simds;
            std::printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_instr_per_s_per_simd\": %.4g, \"wave_instr_per_cycle_per_simd\": %.4f, \"cycles_per_wave_instr\": %.3f}",
                        first ? "" : ",\n", names[kind], waves, ms, per_simd_per_s, per_simd_per_s / clock_hz, clock_hz / per_simd_per_s);
            first = false;
        }
    }
    std::printf("\n]}\n");
    return 0;
}
