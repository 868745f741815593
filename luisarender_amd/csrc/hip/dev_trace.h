// dev_trace.h — BVH4 traversal + triangle intersection for gfx950 (no ray-tracing hardware:
// this is VALU + LDS + memory, one ray per lane, 64 rays per wavefront).
//
// Replaces the reference's Accel::intersect / intersect_any (LuisaCompute, absent submodule;
// call sites src/base/geometry.cpp:218-279) with the published semantics: closest hit returns
// {inst, prim, bary} (src/base/geometry.h:16-28), any-hit returns a bool.
//
// What bounds this loop on CDNA4 is not HBM bytes but the per-CU vector L1 (TCP): a gather where
// every lane touches its own cache line costs one tag lookup per lane per instruction (measured:
// ~1 lane-request/clk/CU, profiles/archive/r01).  So the node fetch is organised around cache lines, not
// lanes:
//   * nodes are 64-byte quantised BVH4 packets (DNodeQ: box origin + per-axis scale + 8-bit child
//     planes + 4 child references) — half the bytes of fp32 child boxes, conservative by
//     construction (lo rounded down, hi rounded up);
//   * each wave fetches the 64 packets its lanes need with FOUR fully coalesced 16-byte-per-lane loads
//     (4 consecutive lanes read one packet = 16 lines per instruction instead of 64) straight into a
//     4 KiB per-wave LDS staging area (gfx950: global_load_lds_dwordx4), organised by quads (below), and
//     every lane reads its own packet back with conflict-free ds_read_b128;
//   * slab tests run directly on the quantised planes: t = fma(q, scale * inv_d, (origin - o) * inv_d),
//     near / far plane words picked by the sign of the ray direction, 24 v_cvt_f32_ubyteN + 24 v_fma_f32
//     per packet;
//   * hits are ordered near->far with a 5-comparator network on packed (t | slot) integer keys;
//   * the traversal stack lives in LDS, [entry][lane] interleaved (conflict-free push/pop); entries
//     beyond kStackLds spill to a per-thread HBM area (lane-coalesced);
//   * every leaf is ONE pre-transformed 48-byte triangle (Moeller-Trumbore, 3 x dwordx4), named by index
//     (the host builder, accel.cpp, may reference a triangle from more than one leaf);
//   * the traversal is RESUMABLE (TravState): the wave leaves the loop as soon as `refill` lanes have
//     finished their rays, those lanes shade and spawn new rays, and everybody re-enters — measured
//     lane occupancy of the loop was 28 % when every lane had to wait for the slowest ray;
//   * a lane traces the shadow ray of a bounce and then the continuation ray back-to-back without
//     leaving the loop.
#pragma once
#include "dev_scene.h"
#include <initializer_list>
#include <type_traits>

namespace lrd {

constexpr uint32_t kBlockThreads = 256u;
constexpr uint32_t kWavesPerBlock = kBlockThreads / 64u;
#ifndef LR_STACK_LDS
#define LR_STACK_LDS 16
#endif
constexpr uint32_t kStackLds = LR_STACK_LDS; // entries per lane kept in LDS
constexpr uint32_t kSpillEntries = 88u;      // HBM overflow entries per lane
constexpr uint32_t kPoolParkedWords = 5u;    // what a pool kernel's lane keeps of its ray in flight on top of its stack across the shading block (megapool_kernel.h)
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kInvalid = 0xffffffffu;

struct Ray {
    f3 o;
    float t_min;
    f3 d;
    float t_max;
};

struct HitRecord {
    uint32_t inst, prim;
    float u, v;
    uint32_t tri;// index of the baked BVH triangle (-> DShadeTri)
};

struct TraceStats {
    uint32_t nodes, tris;
    uint32_t steps, steps_busy;// loop iterations of the wave / iterations in which this lane did work
    uint32_t steps_starved;    // iterations this lane sat out because its pixel had no samples left to start
    uint32_t nodes_empty;      // node visits without a hit child
#ifdef LR_STALL_PROBE
    uint32_t probe_lds;// LDS byte address of this wave's probe words (eight section sums + the last timestamp), 0 = not a sampled wave: THE STALL PROBE below
#endif
};

// THE STALL PROBE (round 6; -DLR_STALL_PROBE, counting kernels only; tools/stall_probe.py).  The box offers no PC sampling and no thread trace
// (profiles/r06a_pc_sampling_unavailable.txt), so the iteration is cut at its waits by hand: an s_memtime at every section boundary, each
// followed by s_waitcnt lgkmcnt(0) (the timestamp itself returns through that counter -- so every LDS access of a section has landed when
// the section's time is taken), the values that cross the boundary pinned on both sides of it.  All marks stand at WAVE-UNIFORM points
// (the divergent regions of the node / leaf step are split around them in the probed flow below); the sums and the last timestamp live in
// nine LDS words per wave, updated by lane 0 (the first form kept them in SGPRs: the loop has none to spare, they were spilled to SCRATCH
// with -amdgpu-spill-sgpr-to-vgpr=0, and every reload's s_waitcnt vmcnt(0) waited for the iteration's gathers as well -- the probed wave
// ran ten times slower and the time piled up wherever the reloads stood: profiles/r06b_stall_probe_first_attempt.txt); lane 0 adds them to
// lrhip_counters::probe after every traversal call.  ONE WAVE IN 36 takes the timestamps (TraceStats::probe_lds != 0).
// Sections (cycles of a wave, whatever it waited for inside them):
//   issue     address arithmetic + the fetch requests of the iteration (packets, and -- fused flow -- the leaf triangles)
//   vm_wait   s_waitcnt vmcnt(0) of trav_fetch_wait: the iteration's gather(s) on their way
//   leaf      the triangle test, its pop
//   packet    the lane's three ds_read_b128 of its staged packet, until they have returned
//   slab      24 cvt + 24 fma + min / max + the sort network: arithmetic only -- what it takes beyond its priced issue cycles is the wave
//             waiting for an issue slot (the SIMD's other waves)
//   chain     the four reference reads, pushes / pop, lgkmcnt(0) of trav_packets_done
//   tail      end of the iteration: votes, turnover of ended rays, exit tests
//   leaf_wait serial flow only (ALPHA pool kernels, one-path kernels): the leaf step's own wait for its triangle
enum : uint32_t { kProbeIssue, kProbeVmWait, kProbeLeaf, kProbePacket, kProbeSlab, kProbeChain, kProbeTail, kProbeLeafWait, kProbeSlots };
#ifdef LR_STALL_PROBE
LR_D uint32_t probe_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return static_cast<uint32_t>(t);
}
typedef __attribute__((address_space(3))) uint32_t probe_lds_u32;
constexpr uint32_t kProbeWords = 16u;// LDS words per wave: [0..7] section sums, [8] the last timestamp
LR_D void probe_start(TraceStats &s) {
    if (s.probe_lds != 0u) {
        const auto t = probe_now();
        if ((threadIdx.x & 63u) == 0u) { reinterpret_cast<probe_lds_u32 *>(static_cast<uintptr_t>(s.probe_lds))[8] = t; }
    }
}
LR_D void probe_mark(TraceStats &s, uint32_t slot) {
    if (s.probe_lds != 0u) {
        const auto t = probe_now();
        if ((threadIdx.x & 63u) == 0u) {
            const auto words = reinterpret_cast<probe_lds_u32 *>(static_cast<uintptr_t>(s.probe_lds));
            const auto last = words[8];
            words[8] = t;
            words[slot] += t - last;
        }
    }
}
template<typename T>
LR_D int probe_pin_one(T &x) { asm volatile("" : "+v"(x)); return 0; }
template<typename... T>
LR_D void probe_pin(T &...x) { (void)std::initializer_list<int>{0, probe_pin_one(x)...}; }
// LR_STALL_PROBE = 1: the SHIPPED flow with four marks (requests out / gathers landed / walk done / iteration done): what the wait for the
// iteration's gathers is of an iteration, undisturbed.  = 2: the split flow with every section (its own cost: it serialises the two gathers
// and runs ~4x slower per iteration -- its arithmetic sections are what it is read for, not its waits).
#define LR_MARK(slot, ...) do { if (COUNT) { probe_pin(__VA_ARGS__); probe_mark(stats, slot); probe_pin(__VA_ARGS__); } } while (0)
#define LR_PIN(...) do { if (COUNT) { probe_pin(__VA_ARGS__); } } while (0)// (values that must not be touched before this point: loads whose wait belongs to the NEXT section)
#if LR_STALL_PROBE >= 2
#define LR_MARK2(slot, ...) LR_MARK(slot, __VA_ARGS__)
#define LR_PIN2(...) LR_PIN(__VA_ARGS__)
#else
#define LR_MARK2(slot, ...) do { } while (0)
#define LR_PIN2(...) do { } while (0)
#endif
#else
#define LR_MARK(slot, ...) do { } while (0)
#define LR_PIN(...) do { } while (0)
#define LR_MARK2(slot, ...) do { } while (0)
#define LR_PIN2(...) do { } while (0)
#endif

struct TraversalStack {
    uint32_t *lds;      // &stack[0][tid]; stride kBlockThreads
    uint32_t *spill;    // &spill[0][gtid]; stride total_threads
    uint32_t spill_stride;
    float4 *stage;      // this wave's 4 KiB staging area (256 x float4)
    // explicit address spaces: with generic pointers the compiler folds the two paths into one flat_load/flat_store
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(1))) uint32_t global_u32;
    LR_D void push(uint32_t sp, uint32_t v) const {
        if (sp < kStackLds) { *(lds_u32 *)(lds + sp * kBlockThreads) = v; }
        else { *(global_u32 *)(spill + static_cast<size_t>(sp - kStackLds) * spill_stride) = v; }
    }
    // the same for entries known to lie in LDS (the wave-level `deep` test of trace_steps): no per-lane range check, no branch
    LR_D void push_lds(uint32_t sp, uint32_t v) const { *(lds_u32 *)(lds + sp * kBlockThreads) = v; }
    LR_D uint32_t pop_lds(uint32_t sp) const { return *(lds_u32 *)(lds + sp * kBlockThreads); }
    LR_D uint32_t pop(uint32_t sp) const {
        uint32_t v;
        if (sp < kStackLds) { v = *(lds_u32 *)(lds + sp * kBlockThreads); }
        else { v = *(global_u32 *)(spill + static_cast<size_t>(sp - kStackLds) * spill_stride); }
        return v;
    }
};

// Round 3 (profiles/archive/r03_valu_peak.json: on gfx950 only v_fma / v_mul / v_add / v_and / v_xor / v_mov issue every 2 cycles per
// wave64; every min / max / cvt / cndmask / cmp / shift / 64-bit add / DPP / packed op takes 4, LDS-crossing ds_bpermute 25):
//   * the packet fetch is organised by QUADS: load j of lane l fetches quarter (l & 3) of the packet of lane (l & ~3) + j, whose node
//     index comes over a DPP quad_perm broadcast (VALU) instead of a ds_bpermute (LDS); addresses are 32-bit offsets from the scalar
//     table base (global_load_lds saddr form), the LDS destinations are wave-uniform SGPRs.  The four 1 KiB regions are 1040 bytes
//     apart, which staggers them over the banks: every lane's ds_read_b128 are conflict-free without an XOR swizzle;
//   * the reference of a sorted child is read back from the staged packet (ds_read_b32 at slot * 4) instead of a three-v_cndmask
//     select per push; the four child words never enter the VGPRs;
//   * one wave-level test per iteration decides whether any lane could reach the HBM overflow area of the stack; if not, every push
//     and pop of the iteration is a bare LDS access.
// A/B of the three against the round-2 forms (ds_bpermute + XOR swizzle, v_cndmask selects, per-entry range checks), C2 at 256 spp:
// 813 -> 828 (quad fetch) / 820 (child reads) / 835 (both) -> 844 (stack test) Msamples/s; profiles/archive/r03b_ab_quad_fetch.txt.
constexpr uint32_t kStageRegion = 65u;     // float4 per load region of the wave's staging area (64 + one float4 of bank stagger)
constexpr uint32_t kStageWave = 4u * kStageRegion;          // float4 per wave

LR_D void cswap(uint32_t &a, uint32_t &b) {
    auto lo = min(a, b), hi = max(a, b);
    a = lo, b = hi;
}

// 1 / d with |result| capped at 1e30: the quantised slab test multiplies plane indices by
// scale * inv_d, and 0 * inf would poison axis-parallel rays with NaNs
LR_D f3 safe_inverse(f3 d) {
    auto one = [](float x) {
        auto r = 1.0f / x;
        return fabsf(r) < 1e30f ? r : copysignf(1e30f, x);
    };
    return mk3(one(d.x), one(d.y), one(d.z));
}

// fmaxf / fminf make the compiler canonicalise every operand that is not the result of an arithmetic instruction (a v_max_f32 x, x
// per ray bound per node step); the slab test's operands are never signalling NaNs, so it names its instructions itself
LR_D float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
LR_D float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
LR_D float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
LR_D float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

LR_D float ubyte_to_float(uint32_t v, int byte) {// v_cvt_f32_ubyteN
    return static_cast<float>((v >> (8 * byte)) & 0xffu);
}

// Resumable per-lane traversal state.  It survives the shading block of OTHER lanes: a wave leaves the
// traversal loop as soon as enough lanes have finished their rays (they go and shade / spawn new
// rays) while the remaining lanes keep cur/sp/stack and continue afterwards — the persistent-threads
// "dynamic fetch" scheme, inside one megakernel.
// kPhasePendingAlpha (a flag on top of Shadow / Closest, ALPHA kernels only): the lane stands at a leaf whose triangle it hit, on
// an instance that may be non-opaque; the candidate (pend_t, pend_u, pend_v) waits for the stochastic alpha test, which runs
// OUTSIDE the traversal loop (resolve_pending_alpha, dev_shade.h) -- see trace_steps.
enum : uint32_t { kPhaseIdle = 0u, kPhaseShadow = 1u, kPhaseClosest = 2u, kPhasePendingAlpha = 4u };
struct TravState {
    f3 o, d;// (1 / d is recomputed on entry to trace_steps: three v_rcp per call instead of three registers live across shading)
    float t_min, t_max;
    uint32_t cur, sp;
    uint32_t phase;
    HitRecord hit;
    bool occluded;
    float pend_t, pend_u, pend_v;// ALPHA kernels: the candidate hit awaiting its alpha test (dead registers elsewhere)
};

LR_D void trav_begin(TravState &tr, const Ray &r, uint32_t phase) {
    tr.o = r.o, tr.d = r.d;
    tr.t_min = r.t_min, tr.t_max = r.t_max;
    tr.cur = 0u, tr.sp = 0u;// root
    tr.phase = phase;
}

// Geometry::_alpha_skip (geometry.cpp:165-192), defined in dev_shade.h next to the texture code
LR_D bool alpha_skip(const DScene &scene, uint32_t inst_id, uint32_t prim, float u, float v);

// What every step of a traversal loop needs of its wave: addresses that do not change over the call.
//
// THE LOOP'S OWN BOOKKEEPING (round 5).  The pipes are full (SQ_ACTIVE_INST_VALU 0.998, round 4), so what an iteration costs is its VALU
// issue cycles, and of the ~960 a pool iteration took, ~110 were not arithmetic of the walk at all:
//   * every wave vote on a COMPOUND predicate (`live && x`) made the compiler materialise the predicate in a VGPR and compare it with
//     zero again (v_cndmask_b32_e64 v, 0, 1, s[..] + v_cmp_ne_u32: 8.4 cycles, five of them per iteration).  Inside the loops a lane's state
//     is therefore carried by `cur` alone, so that every vote is ONE compare:  (int) cur >= 0  the lane stands at an inner node,
//     (int) cur < -2  at a leaf,  cur == kInvalid (-1)  its ray has ended (the job's next ray / the turnover is due),  cur == kCurIdle (-2)
//     it has no ray.  `phase` still says shadow / closest (and parked, ALPHA) -- it is read where a ray ends, not in every iteration;
//   * the stack pointer is the LDS BYTE ADDRESS of the next free entry (`spb`, entries 1 KiB apart: [entry][lane]) instead of an entry
//     index: a push is a store and a full-rate v_add, a pop a v_add and a load -- five half-rate v_lshl_add_u32 per iteration gone -- and
//     "is the stack empty" / "could a node step reach the HBM overflow area" are compares against two wave-uniform bounds (SGPRs): a
//     lane's offset within an entry row is < 1 KiB, so spb - (address of the wave's first lane) orders like the entry index;
//   * the four packet addresses of the cooperative fetch are one shift and four v_or_b32 with a DPP quad_perm operand (half rate, one
//     instruction each) instead of four v_mov_b32 dpp + four v_lshl_or_b32.
// Same walk, same order of every lane's operations: films and counters are bit-identical to round 4's (tests/test_gpu_pool.py twins).
constexpr uint32_t kCurIdle = 0xfffffffeu;
// ALPHA kernels: a lane whose candidate hit waits for its alpha test keeps its leaf, marked with bit 30 (a leaf names its triangle in 27
// bits): such a `cur` is neither an inner node (negative) nor a leaf to test (>= kCurParked as an int) -- the lane sits out until the wave
// has LR_ALPHA_BATCH of them (or nothing else to do) and leaves the loop for the tests.  Round 5: the wave used to leave for EVERY candidate.
constexpr uint32_t kCurParked = 0xc0000000u;
constexpr uint32_t kCurParkBit = 0x40000000u;// what a lane ORs into the leaf it stands at when its candidate waits
constexpr uint32_t kLeafIndexBits = 27u;     // a leaf names its triangle in the low 27 bits (lrhip_upload_scene bounds the table)
constexpr uint32_t kLeafIndexMask = (1u << kLeafIndexBits) - 1u;
static_assert(kCurParkBit > kLeafIndexMask && (kLeafFlag | kCurParkBit) == kCurParked, "the park bit lies above a leaf's triangle index and below the leaf flag");
#ifndef LR_ALPHA_BATCH
#define LR_ALPHA_BATCH 8
#endif
// whether the wave should leave a traversal loop for the alpha tests of its parked candidates
LR_D bool alpha_tests_due(const uint32_t phase, const uint32_t cur) {
    const auto parked = lr_ballot((phase & kPhasePendingAlpha) != 0u);
    if (parked == 0ull) { return false; }
    return static_cast<uint32_t>(__popcll(parked)) >= static_cast<uint32_t>(LR_ALPHA_BATCH) || !lr_any(cur < kCurParked);
}
constexpr uint32_t kStackStride = kBlockThreads * 4u;// bytes between two entries of a lane's LDS stack
struct TravLane {
    const float4 *tris;
    const char *node_base[4];  // the packet table's base for load j of the cooperative fetch: 1040 j bytes BELOW the table (trav_node_fetch)
    uint32_t quarter;          // byte offset of this lane's quarter of a packet (cooperative fetch below)
    const float4 *mine;        // where this lane finds its own packet in the wave's staging area
    uint32_t lds_base;         // LDS byte address of entry 0 of this lane's stack
    uint32_t s_empty, s_deep;  // wave-uniform: spb > s_empty <=> the stack holds an entry; spb > s_deep <=> sp + 3 > kStackLds
    LR_D static TravLane make(const DScene &scene, const TraversalStack &stack) {
        const auto lane = threadIdx.x & 63u;
        // load j: lane l fetches quarter (l & 3) of the packet of lane (l & ~3) + j into region j, float4 slot l.  Lane o therefore finds
        // its own packet in region (o & 3), slots 4 (o >> 2) .. + 3, in order
        const auto base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((TraversalStack::lds_u32 *)stack.lds));
        const auto first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base - lane * 4u)));// (whatever lane is first: the wave's row origin)
        static_assert(kStackLds >= 3u, "the wave-level overflow test assumes at least three LDS entries");
        const auto nodes = reinterpret_cast<const char *>(scene.nodes);
        constexpr auto region = static_cast<ptrdiff_t>(kStageRegion * 16u);
        auto b1 = nodes - region, b2 = nodes - 2 * region, b3 = nodes - 3 * region;
        asm("" : "+s"(b1)); asm("" : "+s"(b2)); asm("" : "+s"(b3));// (four scalar bases: left to itself the compiler re-derives them from one with 64-bit vector adds, per load)
        return TravLane{reinterpret_cast<const float4 *>(scene.bvh_tris), {nodes, b1, b2, b3}, (lane & 3u) << 4u,
                        stack.stage + (lane & 3u) * kStageRegion + (lane >> 2u) * 4u, base,
                        first + 252u, first + (kStackLds - 3u) * kStackStride + 252u};
    }
    LR_D uint32_t spb_of(uint32_t sp) const { return lds_base + sp * kStackStride; }
    LR_D uint32_t sp_of(uint32_t spb) const { return (spb - lds_base) / kStackStride; }
};
// stack accesses by byte address (entries known to lie in LDS), and the general forms (`deep` iterations: the entry may be in the overflow area)
LR_D void spb_store(uint32_t spb, uint32_t v) { *(TraversalStack::lds_u32 *)static_cast<uintptr_t>(spb) = v; }
LR_D uint32_t spb_load(uint32_t spb) { return *(TraversalStack::lds_u32 *)static_cast<uintptr_t>(spb); }
LR_D void trav_push(const TraversalStack &stack, const TravLane &tl, uint32_t &spb, uint32_t v, bool deep) {
    if (deep) { stack.push(tl.sp_of(spb), v); }
    else { spb_store(spb, v); }
    spb += kStackStride;
}
LR_D uint32_t trav_pop(const TraversalStack &stack, const TravLane &tl, uint32_t &spb, bool deep) {// the next entry, kInvalid if there is none
    auto v = kInvalid;
    if (spb > tl.s_empty) {
        spb -= kStackStride;
        v = deep ? stack.pop(tl.sp_of(spb)) : spb_load(spb);
    }
    return v;
}

// ---- node step of the WAVE (every lane calls; `is_inner` lanes test the packet of tr.cur): cooperative packet fetch, quantised slab
// tests, near -> far ordering, pushes.  `deep`: some lane may reach the HBM overflow area of the stack in this iteration.
// the fetch half of the node step: the packets of the lanes at inner nodes are on their way into the wave's staging area
LR_D void trav_node_fetch(const TraversalStack &stack, const TravLane &tl, const TravState &tr, bool is_inner) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void global_void;
    // ---- cooperative packet fetch: 4 coalesced dwordx4 loads -> LDS (global_load_lds_dwordx4: no trip through the VGPRs)
    // -> 4 ds_read_b128 per lane.  Four consecutive lanes read one 64-byte packet: 16 lines per instruction, not 64
    const auto want = (is_inner ? tr.cur : 0u) << 6u;
#define LR_FETCH(j) { \
        const auto w = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(want), (j) * 0x55, 0xf, 0xf, true)) | tl.quarter; /* quad_perm:[j,j,j,j] */ \
        /* the instruction's immediate offset moves BOTH addresses: the four regions of the staging area are named by it (one M0 for all four loads), and the table base of load j is that much lower */ \
        __builtin_amdgcn_global_load_lds((global_void *)(tl.node_base[j] + w), (lds_void *)stack.stage, 16, (j) * (kStageRegion * 16u), 0); }
    LR_FETCH(0) LR_FETCH(1) LR_FETCH(2) LR_FETCH(3)
#undef LR_FETCH
}
// every load of the iteration has landed: the packets are in the LDS
LR_D void trav_fetch_wait() {
    __builtin_amdgcn_s_waitcnt(0);// vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The slab tests of one staged packet and the near -> far order of its children: key[i] = (float_bits(t_near) & ~3) | slot, ascending, a
// missed child's key is kInvalid (negative as an int; a valid key is a non-negative float's bits).
LR_D void trav_slab_sort_q(float4 q0, float4 q1, float4 q2, const TravState &tr, f3 inv, uint32_t (&key)[4]) {
#ifdef LR_PROBE_NODE
    {// sensitivity probe: LR_PROBE_NODE extra dependent VALU ops per node step
        float dummy = tr.t_min;
#pragma unroll
        for (auto i = 0; i < LR_PROBE_NODE; i++) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy)); }
        asm volatile("" ::"v"(dummy));
    }
#endif
    // packet: q0 = (origin.xyz, scale.x)  q1 = (lo_x4, lo_y4, lo_z4, hi_x4) bytes
    //         q2 = (hi_y4, hi_z4, scale.y, scale.z)  q3 = child[4]
    // An EMPTY slot has inverted planes (lo 255, hi 0) and names the scene's sentinel leaf (a triangle nothing hits,
    // lrhip.hip: quantise_node), so it needs no test of its own: it fails the slab test wherever the node has an extent
    // and costs one wasted triangle test where it has none.
    auto ax = q0.w * inv.x, ay = q2.z * inv.y, az = q2.w * inv.z;
    auto bx = (q0.x - tr.o.x) * inv.x, by = (q0.y - tr.o.y) * inv.y, bz = (q0.z - tr.o.z) * inv.z;
    auto lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
    auto hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    // near / far plane words by the sign of the ray direction (scale >= 0): no per-child min/max per axis.  (Bit selects on
    // a per-ray sign mask, six v_bfi_b32 instead of three v_cmp + six v_cndmask, were measured in round 3 -- the issue-cost
    // table has the second v_cndmask behind one v_cmp at ~14 cycles -- and changed nothing: 836.3 / 836.7 vs 834.8 / 836.3.)
    auto nx = inv.x < 0.f ? hix : lox, fx = inv.x < 0.f ? lox : hix;
    auto ny = inv.y < 0.f ? hiy : loy, fy = inv.y < 0.f ? loy : hiy;
    auto nz = inv.z < 0.f ? hiz : loz, fz = inv.z < 0.f ? loz : hiz;
    // (round 4, two more forms of these selects, after profiles/archive/r04h_cndmask_forms.json priced a v_cndmask_b32_e32 right behind another
    // one at 19 cycles: one v_swap_b32 under EXEC per axis -- compare, s_and_saveexec, branch, swap, restore -- lost 2.7 % (965 -> 939);
    // per-ray sign masks in SGPRs, remade at every turnover, and six v_cndmask_b32_e64 on them, no compare: 961 vs 961, the
    // one-path-per-lane kernel 841 vs 848.  Here each pair sits right behind its own v_cmp, which is the cheap case)
#pragma unroll
    for (auto i = 0; i < 4; i++) {// 24 v_cvt_f32_ubyteN + 24 v_fma_f32 (a v_pk_fma_f32 issues no faster than two of them)
        auto tn = vmax3(fmaf(ubyte_to_float(nx, i), ax, bx), fmaf(ubyte_to_float(ny, i), ay, by),
                        vmax2(fmaf(ubyte_to_float(nz, i), az, bz), tr.t_min));
        auto tf = vmin3(fmaf(ubyte_to_float(fx, i), ax, bx), fmaf(ubyte_to_float(fy, i), ay, by),
                        vmin2(fmaf(ubyte_to_float(fz, i), az, bz), tr.t_max));
        auto h = tn <= tf * 1.0000004f;
        key[i] = h ? ((__float_as_uint(tn) & 0xfffffffcu) | static_cast<uint32_t>(i)) : kInvalid;
    }
    // (the slot kept in the key as a byte offset, slot * 4 in four key bits, saves the shift: measured, no change)
    // near -> far: 5-comparator network on (float_bits(t) & ~3) | slot keys (t >= 0)
    cswap(key[0], key[1]);
    cswap(key[2], key[3]);
    cswap(key[0], key[2]);
#if !defined(LR_SORT_COMPARATORS) || LR_SORT_COMPARATORS >= 4// (experiment: 3 = the nearest child first, the others as they fall; tools/bvh_sim.cpp BVH_SIM_PARTIAL_SORT: +1.05 % steps per ray on C2; measured 966 against 989 Msamples/s)
    cswap(key[1], key[3]);
#endif
#if !defined(LR_SORT_COMPARATORS) || LR_SORT_COMPARATORS >= 5
    cswap(key[1], key[2]);
#endif
}
LR_D void trav_slab_sort(const float4 *mine, const TravState &tr, f3 inv, uint32_t (&key)[4]) { trav_slab_sort_q(mine[0], mine[1], mine[2], tr, inv, key); }
// the staged packets are read until here: no lane's next fetch may land before every lane's reads have returned
// WAVE PRIORITIES (round 5; s_setprio: which of a SIMD's ready waves issues first).  What a lane's walk waits for is a chain -- the sorted
// children, the reference reads, the pushes and the pop, the next iteration's addresses, its fetch requests -- and the sooner a wave's
// requests are out, the more of the memory's latency its SIMD neighbour's arithmetic covers.  So that chain runs at priority 3, the
// arithmetic between a fetch and the next chain (the wait, the triangle test, the slab tests and the sort) at 0, the shading block at 2.
// C2 pool kernel 1044 -> 1079 Msamples/s, C3 1026 -> 1052, films bit-identical (profiles/r05zf_setprio.txt).  On the way: the whole
// loop at 3 over a shading block at 0: -1 %; the shading block at 3 over the loop at 0: +0.5 %; only the fetch requests at 3: +2.0 %;
// the chain starting before the sort network, or the leaf's pop raised as well: -0.3 % each; the shading block at 1 / 2 / 3: the same.
#ifndef LR_WAVE_PRIORITIES
#define LR_WAVE_PRIORITIES 1
#endif
LR_D void prio_chain() { if (LR_WAVE_PRIORITIES) { __builtin_amdgcn_s_setprio(3); } }
LR_D void prio_tests() { if (LR_WAVE_PRIORITIES) { __builtin_amdgcn_s_setprio(0); } }
LR_D void prio_shade() { if (LR_WAVE_PRIORITIES) { __builtin_amdgcn_s_setprio(2); } }
LR_D void trav_packets_done() {
    __builtin_amdgcn_s_waitcnt(0xc07f);// lgkmcnt(0) (vmcnt / expcnt untouched)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the pushes / pop of a node step: far -> near so that the nearest is popped first; the nearest stays in `cur`
template<bool D>
LR_D void trav_node_tail(const TraversalStack &stack, const TravLane &tl, TravState &tr, uint32_t &spb, const uint32_t (&key)[4], uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
    // (a lane that hit nothing pushes nothing: its pop goes out with the reference reads, not behind the pushes -- round 5: +0.8 % on the
    // one-path kernel, +1.2 % on the Cornell box)
    auto popped = kInvalid;
    if (static_cast<int>(key[0]) < 0) { popped = trav_pop(stack, tl, spb, D); }
    if (static_cast<int>(key[3]) >= 0) { trav_push(stack, tl, spb, r3, D); }
    if (static_cast<int>(key[2]) >= 0) { trav_push(stack, tl, spb, r2, D); }
    if (static_cast<int>(key[1]) >= 0) { trav_push(stack, tl, spb, r1, D); }
    tr.cur = static_cast<int>(key[0]) >= 0 ? r0 : popped;
}
template<bool COUNT, bool FETCHED = false>
LR_D void trav_node_step(const TraversalStack &stack, const TravLane &tl, TravState &tr, uint32_t &spb, f3 inv, bool is_inner, bool deep, TraceStats &stats) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    if (!FETCHED) {
        prio_chain();
        trav_node_fetch(stack, tl, tr, is_inner);
        LR_MARK(kProbeIssue);
        prio_tests();
        trav_fetch_wait();
        LR_MARK(kProbeVmWait);
    }
    const auto child_words = reinterpret_cast<lds_cu32 *>((lds_void *)(tl.mine + 3));// q3 = child[4] stays in LDS
    auto ref_of = [&](uint32_t k) { return child_words[k & 3u]; };// ds_read_b32 from the staged packet
#if defined(LR_STALL_PROBE) && LR_STALL_PROBE >= 2
    if (COUNT) {// the probed flow: the same operations, the divergent region split at the section boundaries (marks stand at wave-uniform points)
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        if (is_inner) { q0 = tl.mine[0], q1 = tl.mine[1], q2 = tl.mine[2]; }
        LR_MARK(kProbePacket, q0.x, q1.x, q2.x);
        uint32_t key[4] = {kInvalid, kInvalid, kInvalid, kInvalid};
        if (is_inner) {
            stats.nodes++;
            trav_slab_sort_q(q0, q1, q2, tr, inv, key);
        }
        LR_MARK(kProbeSlab, key[0], key[1], key[2], key[3]);
        if (is_inner) {
            prio_chain();
            const auto r0 = ref_of(key[0]), r1 = ref_of(key[1]), r2 = ref_of(key[2]), r3 = ref_of(key[3]);
            if (deep) { trav_node_tail<true>(stack, tl, tr, spb, key, r0, r1, r2, r3); }
            else { trav_node_tail<false>(stack, tl, tr, spb, key, r0, r1, r2, r3); }
        }
        trav_packets_done();
        LR_MARK(kProbeChain, tr.cur, spb);
        return;
    }
#endif
    if (is_inner) {
        if (COUNT) { stats.nodes++; }
        uint32_t key[4];
        trav_slab_sort(tl.mine, tr, inv, key);
        prio_chain();
        // (MEASURED, NOT KEPT, round 5: the four references through the registers -- a fourth ds_read_b128, two compares and three selects per
        // reference instead of a dependent LDS round trip: 533 against 1012 Msamples/s, profiles/r05n_refs_in_vgprs.txt; runs of v_cndmask on one VCC again)
        // push far -> near so that the nearest is popped first; keep the nearest in `cur`.  (ONE wave-level branch on `deep` around the
        // lot, not one per access: a wave's scalar and branch instructions cost it issue slots like its vector ones)
        // the four references are requested together -- one LDS round trip instead of up to four in a row (a missed child's key reads slot 3:
        // harmless); round 5, with the loop lighter: +0.5 % on both kernel families, profiles/r05r_refs_batched_turnover.txt
        const auto r0 = ref_of(key[0]), r1 = ref_of(key[1]), r2 = ref_of(key[2]), r3 = ref_of(key[3]);
        if (deep) { trav_node_tail<true>(stack, tl, tr, spb, key, r0, r1, r2, r3); }
        else { trav_node_tail<false>(stack, tl, tr, spb, key, r0, r1, r2, r3); }
#ifndef LR_TRACE_PROBE
        if (COUNT && key[0] == kInvalid) { stats.nodes_empty++; }
#endif
    }
    trav_packets_done();
}

// ---- leaf step of a lane standing at a leaf: Moeller-Trumbore on ONE pre-transformed triangle (3 x dwordx4).  The host builds
// one-triangle leaves (accel.cpp): with the wave's lanes at different depths a leaf loop runs for the longest leaf
// of the wave every step, and at 4 triangles per leaf that cost more than the extra level of boxes
// (measured on C2: 422 -> 537 Msamples/s, tris/ray 12.3 -> 3.4, nodes/ray 19.2 -> 21.6).
//
// MEASURED IN ROUND 4 AND NOT KEPT (the code is gone in round 5; profiles/archive/r04d_leaf_batching.txt, r04e_fused_fetch.txt):
//   * leaf batching -- a lane that arrives at a leaf keeps it in a register, goes on with its stack, and the wave runs the leaf step when
//     12 ... 40 lanes hold one: the pool kernel 898 -> 889 / 891 / 875 / 835 / 767 Msamples/s, the one-path kernel 850 -> 820 ... 497; lanes
//     simply WAITING at their leaf until 12 of them do: +0.6 %;
//   * fused fetch -- the node packets and the leaf triangles of an iteration requested together and waited for once: +-1 %.
// The leaf step's price is its triangle fetch's latency, which the SIMD's other three waves cover; the instructions such schemes save
// are fewer than the votes, extra pops and longer live ranges they add to EVERY iteration.
// LR_LEAF_TAIL_ONE_COMPARE: what the leaf test does with a hit, written around ONE compare of the phase (pool kernels without the alpha test: three
// VALU instructions per iteration, C2 1102 -> 1115, C3 1100 -> 1110, C4 1229 -> 1235 Msamples/s, films bit-identical; the one-path kernels gain
// nothing and keep the round 1-5 form; the ALPHA kernels' allocation takes it badly -- <5128> 32 -> 65 spilled VGPRs: profiles/r06zq_leaf_tail.txt)
#ifndef LR_LEAF_TAIL_ONE_COMPARE
#if defined(LR_VARIANT) && ((LR_VARIANT) & 4096)
#define LR_LEAF_TAIL_ONE_COMPARE 1
#else
#define LR_LEAF_TAIL_ONE_COMPARE 0
#endif
#endif
struct LeafTriangle { float4 a, b, c; };
LR_D LeafTriangle trav_leaf_fetch(const TravLane &tl, uint32_t ref) {
    // (a 32-bit byte offset from the scalar table base: 48 B x 2^27 triangles does not fit 32 bits, 48 B x the 89 M a 4 GB table holds does -- lrhip_upload_scene refuses more)
    auto tb = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tl.tris) + (ref & kLeafIndexMask) * 48u);
    // (non-temporal loads here -- a leaf's triangle is touched once -- were measured in round 4: C2 854 -> 761 Msamples/s)
    return LeafTriangle{tb[0], tb[1], tb[2]};
}
// the triangle test of a lane at leaf `ref`; returns true if the ray has found an occluder (a shadow ray's hit: the rest of its stack is dropped)
template<bool COUNT, bool ALPHA>
LR_D bool trav_leaf_test(TravState &tr, uint32_t ref, const LeafTriangle &tri, TraceStats &stats) {
    auto found = false;
    auto a = tri.a, b = tri.b, c = tri.c;
    if (COUNT) { stats.tris++; }
#ifdef LR_PROBE_LEAF
    {
        float dummy = tr.t_min;
#pragma unroll
        for (auto i = 0; i < LR_PROBE_LEAF; i++) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy)); }
        asm volatile("" ::"v"(dummy));
    }
#endif
    auto flags = __float_as_uint(c.w);
    f3 p0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
#ifdef LR_EXACT_LEAF
    // (`make ieee` and the volumetric kernels: the oracle's arithmetic -- no contraction, exact division)
    auto pvec = cross(tr.d, e2);
    auto det = dot(e1, pvec);
    auto inv_det = 1.f / det;
    auto tvec = tr.o - p0;
    auto u = dot(tvec, pvec) * inv_det;
    auto qvec = cross(tvec, e1);
    auto v = dot(tr.d, qvec) * inv_det;
    auto t = dot(e2, qvec) * inv_det;
    auto uv_sum = u + v;
#else
    // THE FUSED MULTIPLY-ADDS ARE WRITTEN OUT (round 6).  Left to -ffp-contract=fast the compiler chose which product of a sum to fuse per
    // INSTANTIATION of this function -- the serial and the fused flow, a counting twin, a build with one register initialised differently --
    // and a hit point that differs in its last bit sends the path elsewhere within a few bounces: films of such builds differed by 1e-3
    // (profiles/r06zn_triangle_test_written_out.txt).  One evaluation order now, whatever the context: cross products as fma(a, b, -(c * d)),
    // dot products x first, then y, then z.
    const auto crossf = [](f3 p, f3 q) {
        return mk3(__builtin_fmaf(p.y, q.z, -(p.z * q.y)), __builtin_fmaf(p.z, q.x, -(p.x * q.z)), __builtin_fmaf(p.x, q.y, -(p.y * q.x)));
    };
    const auto dotf = [](f3 p, f3 q) { return __builtin_fmaf(p.z, q.z, __builtin_fmaf(p.y, q.y, p.x * q.x)); };
    float u, v, t, uv_sum, det;
    {
#pragma clang fp contract(off)
        const auto pvec = crossf(tr.d, e2);
        det = dotf(e1, pvec);
        const auto inv_det = __builtin_amdgcn_rcpf(det);// (a denormal det is a degenerate triangle either way)
        const auto tvec = mk3(tr.o.x - p0.x, tr.o.y - p0.y, tr.o.z - p0.z);
        u = dotf(tvec, pvec) * inv_det;
        const auto qvec = crossf(tvec, e1);
        v = dotf(tr.d, qvec) * inv_det;
        t = dotf(e2, qvec) * inv_det;
        uv_sum = u + v;
    }
#endif
    // (det == 0: 1 / det is infinite, u and v come out infinite or NaN, and one of the tests below fails by itself -- +inf + +inf > 1, -inf fails the
    // minimum, inf - inf and 0 * inf are NaN: the determinant need not be asked.  u >= 0 && v >= 0 as ONE compare of their minimum: a NaN in one
    // of them makes v_min return the other, but then u + v is NaN and fails too.  One VALU and two scalar instructions per iteration: C2 1124 -> 1130,
    // C3 1113 -> 1119, C5 626 -> 630 Msamples/s, films bit-identical: profiles/r06zt_leaf_test_compares.txt)
    (void)det;
    auto ok = fminf(u, v) >= 0.f && uv_sum <= 1.f && t > tr.t_min && t < tr.t_max && (flags & 1u);
    if (ALPHA && ok && (flags & 2u) == 0u) {// park the candidate: the alpha test runs outside this loop
        tr.pend_t = t, tr.pend_u = u, tr.pend_v = v;
        tr.phase |= kPhasePendingAlpha;
        ok = false;
    }
    if constexpr (!ALPHA && LR_LEAF_TAIL_ONE_COMPARE != 0) {
        // (a lane at a leaf traces a shadow ray or a closest-hit ray -- kPhaseShadow / kPhaseClosest, nothing else where no alpha test parks candidates:
        // ONE compare serves both uses of the phase)
        const auto shadow = tr.phase == kPhaseShadow;
        if (ok) {
            tr.t_max = t;
            if (!shadow) {
                tr.hit.inst = __float_as_uint(a.w), tr.hit.prim = __float_as_uint(b.w);
                tr.hit.u = u, tr.hit.v = v;
                tr.hit.tri = ref & kLeafIndexMask;
            }
        }
        found = ok && shadow;
        if (found) { tr.occluded = true; }
        return found;
    }
    if (ok) {
        tr.t_max = t;
        found = true;
        if (tr.phase == kPhaseClosest) {
            tr.hit.inst = __float_as_uint(a.w), tr.hit.prim = __float_as_uint(b.w);
            tr.hit.u = u, tr.hit.v = v;
            tr.hit.tri = ref & kLeafIndexMask;
        }
    }
    if (tr.phase == kPhaseShadow && found) {
        tr.occluded = true;
        return true;
    }
    return false;
}
template<bool COUNT, bool ALPHA>
LR_D void trav_leaf_step(const TraversalStack &stack, const TravLane &tl, TravState &tr, uint32_t &spb, bool deep, TraceStats &stats) {
    if (LR_WAVE_PRIORITIES == 2) { prio_chain(); }
    const auto tri = trav_leaf_fetch(tl, tr.cur);
    // what the lane goes on with is read from its stack while the triangle is on its way (a lane that finds an occluder, or parks a candidate
    // for its alpha test, does not take it): one LDS round trip off the iteration's critical path -- round 5: the one-path kernel +1.5 % on
    // C2, +3.5 % on the Cornell box (profiles/r05s_early_pops.txt).  The pool kernels' fused flow (trav_iteration) pops behind the test:
    // there the same move LOST 7.5 %.
    auto spb_next = spb;
    const auto next = trav_pop(stack, tl, spb_next, deep);
    prio_tests();
    if (trav_leaf_test<COUNT, ALPHA>(tr, tr.cur, tri, stats)) { spb = tl.lds_base, tr.cur = kInvalid; }// any-hit: drop the rest of the stack
    else if (!ALPHA || !(tr.phase & kPhasePendingAlpha)) { tr.cur = next, spb = spb_next; }
    else { tr.cur |= kCurParkBit; }// (a parked lane keeps its leaf, marked: kCurParked)
}

// ONE ITERATION's walk for the wave: the lanes at inner nodes test their packets, the lanes at leaves (the ones that were, and the ones the
// node step has just sent there) their triangles.
//
// MEASURED IN ROUND 5 AND NOT KEPT (profiles/r05d_pipelined_iteration.txt): a PIPELINED iteration -- the four child references and the pop of a
// lane that hit nothing read from the LDS together and waited for once, the triangle loads of every lane at a leaf issued BEFORE the
// pushes, what a leaf lane goes on with read while its triangle is on its way.  Five LDS round trips shorter per iteration, films
// bit-identical, and SLOWER: the pool kernel 959 against 1005 Msamples/s (C2, 256 spp), the one-path kernel 837 against 898, the Cornell box
// 3662 against 4029.  It executes more instructions (five exec regions per iteration instead of two, unconditional address arithmetic),
// and a wave issues at most one instruction of ANY kind every ~4.5 cycles: what an iteration costs a wave is its instruction count --
// scalar and branch instructions included -- as much as the round trips it waits for.
template<bool COUNT, bool ALPHA, bool FUSED = false>
LR_D void trav_iteration(const TraversalStack &stack, const TravLane &tl, TravState &tr, uint32_t &spb, f3 inv, TraceStats &stats) {
    const auto is_inner = static_cast<int>(tr.cur) >= 0;
    // one WAVE-LEVEL test per iteration decides whether any lane could touch the HBM overflow area of the stack in this iteration
    // (a lane at an inner node pushes at most three entries); if none can -- nearly always -- every push and pop of the
    // iteration is a bare LDS access instead of a compare + branch + access per entry (round 3: +1 %)
    const auto deep = lr_any(spb > tl.s_deep);
    if (FUSED) {// (experiment LR_POOL_FUSED_FETCH: both gathers of the iteration requested up front, one wait; a lane the node step sends to a leaf tests it in the NEXT iteration)
        const auto is_leaf = static_cast<int>(tr.cur) < static_cast<int>(kCurParked);
        const auto any_inner = lr_any(is_inner);
        prio_chain();
        if (any_inner) { trav_node_fetch(stack, tl, tr, is_inner); }
        // (the lanes that fetch no triangle never read one: their registers need a DEFINED value, not a particular one -- an empty asm "writes" them
        // instead of the ten v_mov_b32 per iteration of `LeafTriangle tri{}`: 259 -> 250 VALU instructions in the pool loop, C2 1091 -> 1101, C3 1087 ->
        // 1100, C4 1224 -> 1235 Msamples/s, films bit-identical now that trav_leaf_test's arithmetic is written out: profiles/r06zn_triangle_test_written_out.txt)
        LeafTriangle tri;
        asm volatile("" : "=v"(tri.a.x), "=v"(tri.a.y), "=v"(tri.a.z), "=v"(tri.a.w), "=v"(tri.b.x), "=v"(tri.b.y), "=v"(tri.b.z), "=v"(tri.b.w),
                          "=v"(tri.c.x), "=v"(tri.c.y), "=v"(tri.c.z), "=v"(tri.c.w));
        if (is_leaf) { tri = trav_leaf_fetch(tl, tr.cur); }
        LR_MARK(kProbeIssue);
        prio_tests();
        trav_fetch_wait();
        LR_MARK(kProbeVmWait);
        LR_PIN2(tri.a.x, tri.b.x, tri.c.x, tri.c.w);
        if (is_leaf) {
            if (trav_leaf_test<COUNT, ALPHA>(tr, tr.cur, tri, stats)) { spb = tl.lds_base; }
            if (!ALPHA || !(tr.phase & kPhasePendingAlpha)) { tr.cur = trav_pop(stack, tl, spb, deep); }
            else { tr.cur |= kCurParkBit; }
        }
        LR_MARK2(kProbeLeaf, tr.cur, spb);
        if (any_inner) { trav_node_step<COUNT, true>(stack, tl, tr, spb, inv, is_inner, deep, stats); }
#if defined(LR_STALL_PROBE) && LR_STALL_PROBE < 2
        LR_MARK(kProbeChain, tr.cur, spb);// (level 1: everything of the walk behind the wait -- triangle test, slab tests, sort, pushes / pops)
#endif
        return;
    }
    if (lr_any(is_inner)) { trav_node_step<COUNT>(stack, tl, tr, spb, inv, is_inner, deep, stats); }
#if defined(LR_STALL_PROBE) && LR_STALL_PROBE >= 2
    if (COUNT) {// the serial flow's leaf step, split at its wait (trav_leaf_step below is what ships)
        const auto at_leaf = static_cast<int>(tr.cur) < static_cast<int>(kCurParked);
        LeafTriangle tri{};
        auto spb_next = spb;
        auto next = kInvalid;
        if (at_leaf) {
            tri = trav_leaf_fetch(tl, tr.cur);
            next = trav_pop(stack, tl, spb_next, deep);
        }
        LR_MARK(kProbeIssue);
        prio_tests();
        __builtin_amdgcn_s_waitcnt(0);
        LR_MARK(kProbeLeafWait);
        LR_PIN(tri.a.x, tri.b.x, tri.c.x, tri.c.w, next);
        if (at_leaf) {
            if (trav_leaf_test<COUNT, ALPHA>(tr, tr.cur, tri, stats)) { spb = tl.lds_base, tr.cur = kInvalid; }
            else if (!ALPHA || !(tr.phase & kPhasePendingAlpha)) { tr.cur = next, spb = spb_next; }
            else { tr.cur |= kCurParkBit; }
        }
        LR_MARK(kProbeLeaf, tr.cur, spb);
        return;
    }
#endif
    if (static_cast<int>(tr.cur) < static_cast<int>(kCurParked)) {// at a leaf (ALPHA: not one that waits for its alpha test)
        trav_leaf_step<COUNT, ALPHA>(stack, tl, tr, spb, deep, stats);
    }
#if defined(LR_STALL_PROBE) && LR_STALL_PROBE < 2
    LR_MARK(kProbeChain, tr.cur, spb);// (level 1, serial flow: slab tests, sort, pushes / pops AND the leaf step with its own wait for the triangle)
#endif
}

// Runs traversal steps for the whole wave until no lane has a ray in flight or at least `refill`
// lanes have finished theirs.  A lane in kPhaseShadow that finishes switches to `next_closest`
// (if has_next) without leaving the loop.  Must be called by all 64 lanes.
// ALPHA: candidate hits on maybe-non-opaque instances (triangle flag bit 1 clear) pass through the
// stochastic alpha test before they are committed, for closest-hit and any-hit rays alike
// (Geometry::trace_closest / trace_any ray-query branch, geometry.cpp:248-279).  The test (uv interpolation, a hash, a texture
// lookup) is NOT in this loop: a lane with such a candidate parks it (kPhasePendingAlpha), the wave leaves the loop, the caller
// resolves the parked candidates (resolve_pending_alpha) and calls again with the same `idle_at_entry`.  Round 2: with the test
// inlined here the loop of the ALPHA variants carried the texture code's registers and calls, and a scene with 2 % alpha-tested
// triangles ran at 390 instead of 519 Msamples/s (tools/c5_ablation.py).
template<bool COUNT, bool ALPHA>
LR_D void trace_steps(const DScene &scene, const TraversalStack &stack, TravState &tr, bool has_next,
                      const Ray &next_closest, int refill, TraceStats &stats, bool idle_at_entry) {
    const auto tl = TravLane::make(scene, stack);
    auto inv = safe_inverse(tr.d);
    auto spb = tl.spb_of(tr.sp);
    if (tr.phase == kPhaseIdle) { tr.cur = kCurIdle; }// (the loop reads a lane's state off `cur`: see TravLane)
    const auto fresh = ~lr_ballot(idle_at_entry);// lanes that came with a ray: the ones whose end the caller counts
    {// (nothing in flight, or -- back from an alpha test -- enough rays ended already: the tests at the loop's end, which only run when a ray ends)
        const auto idle = lr_ballot(tr.cur == kCurIdle);
        if (idle == ~0ull || __popcll(idle & fresh) >= refill) { return; }
    }
    for (;;) {
#ifdef LR_TRACE_PROBE// (section cycles of the loop in the counting build: node step -> nodes_empty, end of iteration -> trace_steps_starved; lane 0 reports)
        if (COUNT) { stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u; }
        const auto probe_t0 = __builtin_readcyclecounter();
#else
        if (COUNT) { stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u, stats.steps_starved += idle_at_entry ? 1u : 0u; }
#endif
        trav_iteration<COUNT, ALPHA>(stack, tl, tr, spb, inv, stats);
#ifdef LR_TRACE_PROBE
        const auto probe_t1 = __builtin_readcyclecounter();
#endif
#ifdef LR_TRACE_PROBE
        const auto probe_t2 = __builtin_readcyclecounter();
        if (COUNT && (threadIdx.x & 63u) == 0u) {
            stats.nodes_empty += static_cast<uint32_t>(probe_t1 - probe_t0);
            stats.steps_starved += static_cast<uint32_t>(__builtin_readcyclecounter() - probe_t2);
        }
#endif
        if (ALPHA && alpha_tests_due(tr.phase, tr.cur)) { break; }
        // ---- ray finished: switch from the shadow ray to the closest-hit ray, or go idle.  (Nothing the tests below look at changes
        // in an iteration in which no ray ended.)
        const auto ended = tr.cur == kInvalid;
        if (!lr_any(ended)) { continue; }
        if (ended) {
            if (tr.phase == kPhaseShadow && has_next) {
                trav_begin(tr, next_closest, kPhaseClosest);
                inv = safe_inverse(tr.d);
                spb = tl.lds_base;
            } else {
                tr.phase = kPhaseIdle, tr.cur = kCurIdle;
            }
        }
        const auto idle = lr_ballot(tr.cur == kCurIdle);
        if (idle == ~0ull) { break; }
        if (__popcll(idle & fresh) >= refill) { break; }
    }
    prio_shade();
    tr.sp = tl.sp_of(spb);
}

}// namespace lrd
