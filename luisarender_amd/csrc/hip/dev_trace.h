// dev_trace.h — BVH4 traversal + triangle intersection for gfx950 (no ray-tracing hardware:
// this is VALU + LDS + memory, one ray per lane, 64 rays per wavefront).
//
// Replaces the reference's Accel::intersect / intersect_any (LuisaCompute, absent submodule;
// call sites src/base/geometry.cpp:218-279) with the published semantics: closest hit returns
// {inst, prim, bary} (src/base/geometry.h:16-28), any-hit returns a bool.
//
// What bounds this loop on CDNA4 is not HBM bytes but the per-CU vector L1 (TCP): a gather where
// every lane touches its own cache line costs one tag lookup per lane per instruction (measured:
// ~1 lane-request/clk/CU, profiles/r01).  So the node fetch is organised around cache lines, not
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

namespace lrd {

constexpr uint32_t kBlockThreads = 256u;
constexpr uint32_t kWavesPerBlock = kBlockThreads / 64u;
#ifndef LR_STACK_LDS
#define LR_STACK_LDS 16
#endif
constexpr uint32_t kStackLds = LR_STACK_LDS; // entries per lane kept in LDS
constexpr uint32_t kSpillEntries = 88u;      // HBM overflow entries per lane
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
};

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

// Round 3 (profiles/r03_valu_peak.json: on gfx950 only v_fma / v_mul / v_add / v_and / v_xor / v_mov issue every 2 cycles per
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
// 813 -> 828 (quad fetch) / 820 (child reads) / 835 (both) -> 844 (stack test) Msamples/s; profiles/r03b_ab_quad_fetch.txt.
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
struct TravLane {
    const float4 *tris;
    const char *node_bytes;
    uint32_t quarter;          // byte offset of this lane's quarter of a packet (cooperative fetch below)
    const float4 *mine;        // where this lane finds its own packet in the wave's staging area
    LR_D static TravLane make(const DScene &scene, const TraversalStack &stack) {
        const auto lane = threadIdx.x & 63u;
        // load j: lane l fetches quarter (l & 3) of the packet of lane (l & ~3) + j into region j, float4 slot l.  Lane o therefore finds
        // its own packet in region (o & 3), slots 4 (o >> 2) .. + 3, in order
        return TravLane{reinterpret_cast<const float4 *>(scene.bvh_tris), reinterpret_cast<const char *>(scene.nodes), (lane & 3u) << 4u,
                        stack.stage + (lane & 3u) * kStageRegion + (lane >> 2u) * 4u};
    }
};

// ---- node step of the WAVE (every lane calls; `is_inner` lanes test the packet of tr.cur): cooperative packet fetch, quantised slab
// tests, near -> far ordering, pushes.  `deep`: some lane may reach the HBM overflow area of the stack in this iteration.
// the fetch half of the node step: the packets of the lanes at inner nodes are on their way into the wave's staging area
LR_D void trav_node_fetch(const TraversalStack &stack, const TravLane &tl, const TravState &tr, bool is_inner) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void global_void;
    // ---- cooperative packet fetch: 4 coalesced dwordx4 loads -> LDS (global_load_lds_dwordx4: no trip through the VGPRs)
    // -> 4 ds_read_b128 per lane.  Four consecutive lanes read one 64-byte packet: 16 lines per instruction, not 64
    auto want = is_inner ? tr.cur : 0u;
#define LR_FETCH(j) { \
        const auto w = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(want), (j) * 0x55, 0xf, 0xf, false)); /* quad_perm:[j,j,j,j] */ \
        __builtin_amdgcn_global_load_lds((global_void *)(tl.node_bytes + ((w << 6u) | tl.quarter)), (lds_void *)(stack.stage + (j) * kStageRegion), 16, 0, 0); }
    LR_FETCH(0) LR_FETCH(1) LR_FETCH(2) LR_FETCH(3)
#undef LR_FETCH
}
// every load of the iteration has landed: the packets are in the LDS (and the triangles of the lanes at leaves in their registers)
LR_D void trav_fetch_wait() {
    __builtin_amdgcn_s_waitcnt(0);// vmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template<bool COUNT, bool FETCHED = false>
LR_D void trav_node_step(const TraversalStack &stack, const TravLane &tl, TravState &tr, f3 inv, bool is_inner, bool deep, TraceStats &stats) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    if (!FETCHED) {
        trav_node_fetch(stack, tl, tr, is_inner);
        trav_fetch_wait();
    }
    const auto mine = tl.mine;
    auto q0 = mine[0], q1 = mine[1], q2 = mine[2];
    const auto child_words = reinterpret_cast<lds_cu32 *>((lds_void *)(mine + 3));// q3 = child[4] stays in LDS
    if (is_inner) {
        if (COUNT) { stats.nodes++; }
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
        uint32_t key[4];
        // near / far plane words by the sign of the ray direction (scale >= 0): no per-child min/max per axis.  (Bit selects on
        // a per-ray sign mask, six v_bfi_b32 instead of three v_cmp + six v_cndmask, were measured in round 3 -- the issue-cost
        // table has the second v_cndmask behind one v_cmp at ~14 cycles -- and changed nothing: 836.3 / 836.7 vs 834.8 / 836.3.)
        auto nx = inv.x < 0.f ? hix : lox, fx = inv.x < 0.f ? lox : hix;
        auto ny = inv.y < 0.f ? hiy : loy, fy = inv.y < 0.f ? loy : hiy;
        auto nz = inv.z < 0.f ? hiz : loz, fz = inv.z < 0.f ? loz : hiz;
        // (round 4, two more forms of these selects, after profiles/r04h_cndmask_forms.json priced a v_cndmask_b32_e32 right behind another
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
        auto ref_of = [&](uint32_t k) { return child_words[k & 3u]; };// ds_read_b32 from the staged packet
        // near -> far: 5-comparator network on (float_bits(t) & ~3) | slot keys (t >= 0)
        cswap(key[0], key[1]);
        cswap(key[2], key[3]);
        cswap(key[0], key[2]);
        cswap(key[1], key[3]);
        cswap(key[1], key[2]);
        // push far -> near so that the nearest is popped first; keep the nearest in `cur`
        if (deep) {
            if (key[3] != kInvalid) { stack.push(tr.sp++, ref_of(key[3])); }
            if (key[2] != kInvalid) { stack.push(tr.sp++, ref_of(key[2])); }
            if (key[1] != kInvalid) { stack.push(tr.sp++, ref_of(key[1])); }
        } else {
            if (key[3] != kInvalid) { stack.push_lds(tr.sp++, ref_of(key[3])); }
            if (key[2] != kInvalid) { stack.push_lds(tr.sp++, ref_of(key[2])); }
            if (key[1] != kInvalid) { stack.push_lds(tr.sp++, ref_of(key[1])); }
        }
#ifndef LR_TRACE_PROBE
        if (COUNT && key[0] == kInvalid) { stats.nodes_empty++; }
#endif
        if (key[0] != kInvalid) { tr.cur = ref_of(key[0]); }
        else if (tr.sp > 0u) { tr.cur = deep ? stack.pop(--tr.sp) : stack.pop_lds(--tr.sp); }
        else { tr.cur = kInvalid; }
    }
    // the staged packets are read until here: no lane's next fetch may land before every lane's reads have returned
    __builtin_amdgcn_s_waitcnt(0xc07f);// lgkmcnt(0) (vmcnt / expcnt untouched)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- leaf step of a lane standing at a leaf: Moeller-Trumbore on ONE pre-transformed triangle (3 x dwordx4).  The host builds
// one-triangle leaves (accel.cpp): with the wave's lanes at different depths a leaf loop runs for the longest leaf
// of the wave every step, and at 4 triangles per leaf that cost more than the extra level of boxes
// (measured on C2: 422 -> 537 Msamples/s, tris/ray 12.3 -> 3.4, nodes/ray 19.2 -> 21.6).
//
// LEAF BATCHING (round 4, LR_LEAF_BATCH > 0).  With one-triangle leaves a ray takes one leaf step per six node steps, so in a loop
// that runs "node step, then leaf step" every iteration the leaf step works for 0.13 of the lanes that have a ray -- and costs 2050
// cycles of the iteration's 5100 (section probes, profiles/r04c).  A lane that arrives at a leaf therefore POSTPONES it: it keeps the
// leaf in a register (`leaf`) and goes on with the next entry of its stack; the wave runs the leaf step when LR_LEAF_BATCH lanes hold a
// postponed leaf (or no lane has an inner node left), and a lane that reaches a second leaf before that waits.  A lane's leaves are
// still tested in the order it found them, so hits (and ties) come out as before; what changes is that the boxes between a postponed
// leaf and its test are culled against the t_max of before that test: a few more node visits, the same results.  Outside the loop
// nothing is postponed: trav_unpostpone puts the lane back where it stood (the entry it went on with returns to the stack).
// MEASURED, and OFF by default: a model of the loop with the probes' section costs promised 1.16-1.29x on the traversal; on the box
// (C2, 256 spp, Msamples/s) the pool kernel went 898 -> 889 / 891 / 875 / 835 / 767 at LR_LEAF_BATCH 12 / 16 / 24 / 32 / 40 and the
// one-path-per-lane kernel 850 -> 820 / 804 / 720 / 569 / 497 (profiles/r04d_leaf_batching.txt): same films, same rays, 0.5 % fewer
// boxes, steps per ray 20.8 against the minimum of 19.6 -- the lanes do not wait long, the iterations simply do not get cheaper.  The
// leaf step's price is its triangle fetch's latency, which the other three waves of the SIMD cover; the instructions it saves are
// fewer than the ballots, the extra pop and the longer live ranges that batching adds to EVERY iteration.
struct LeafTriangle { float4 a, b, c; };
LR_D LeafTriangle trav_leaf_fetch(const TravLane &tl, uint32_t ref) {
    auto tb = tl.tris + static_cast<size_t>(ref & ((1u << 27u) - 1u)) * 3u;
    // (non-temporal loads here -- a leaf's triangle is touched once -- were measured in round 4: C2 854 -> 761 Msamples/s)
    return LeafTriangle{tb[0], tb[1], tb[2]};
}
template<bool COUNT, bool ALPHA, bool POSTPONED = false>
LR_D void trav_leaf_step(const TraversalStack &stack, const TravLane &tl, TravState &tr, bool deep, TraceStats &stats, uint32_t *leaf = nullptr,
                         const LeafTriangle *fetched = nullptr) {
    auto found = false;
    const auto ref = POSTPONED ? *leaf : tr.cur;
    {
        const auto tri = fetched != nullptr ? *fetched : trav_leaf_fetch(tl, ref);
        auto a = tri.a, b = tri.b, c = tri.c;
#ifdef LR_LEAF_FULL_QUADS// (a kernel that never reads hit.inst / hit.prim gets two 12-byte loads here: keep them 16-byte ones)
        asm volatile("" ::"v"(a.w), "v"(b.w));
#endif
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
        auto pvec = cross(tr.d, e2);
        auto det = dot(e1, pvec);
#ifdef LR_EXACT_LEAF
        auto inv_det = 1.f / det;// (`make ieee`: the experiment build with the oracle's arithmetic)
#else
        auto inv_det = __builtin_amdgcn_rcpf(det);// (a denormal det is a degenerate triangle either way)
#endif
        auto tvec = tr.o - p0;
        auto u = dot(tvec, pvec) * inv_det;
        auto qvec = cross(tvec, e1);
        auto v = dot(tr.d, qvec) * inv_det;
        auto t = dot(e2, qvec) * inv_det;
        auto ok = det != 0.f && u >= 0.f && v >= 0.f && u + v <= 1.f && t > tr.t_min && t < tr.t_max && (flags & 1u);
        if (ALPHA && ok && (flags & 2u) == 0u) {// park the candidate: the alpha test runs outside this loop
            tr.pend_t = t, tr.pend_u = u, tr.pend_v = v;
            tr.phase |= kPhasePendingAlpha;
            ok = false;
        }
        if (ok) {
            tr.t_max = t;
            found = true;
            if (tr.phase == kPhaseClosest) {
                tr.hit.inst = __float_as_uint(a.w), tr.hit.prim = __float_as_uint(b.w);
                tr.hit.u = u, tr.hit.v = v;
                tr.hit.tri = ref & ((1u << 27u) - 1u);
            }
        }
    }
    if (tr.phase == kPhaseShadow && found) {
        tr.occluded = true;
        tr.sp = 0u;// any-hit: drop the rest of the stack
        if (POSTPONED) { tr.cur = kInvalid; }
    }
    if (!ALPHA || !(tr.phase & kPhasePendingAlpha)) {// (a parked lane stays at its leaf)
        if (POSTPONED) { *leaf = kInvalid; }
        else { tr.cur = tr.sp > 0u ? (deep ? stack.pop(--tr.sp) : stack.pop_lds(--tr.sp)) : kInvalid; }
    }
}

#ifndef LR_LEAF_BATCH
#define LR_LEAF_BATCH 0// lanes with a postponed leaf that make the wave run the leaf step; 0: a leaf step every iteration (rounds 1-3)
#endif
// a lane that stands at a leaf and has none postponed keeps it for later and goes on with its stack
LR_D void trav_postpone(const TraversalStack &stack, TravState &tr, uint32_t &leaf, bool live, bool deep) {
    if (live && leaf == kInvalid && tr.cur != kInvalid && (tr.cur & kLeafFlag) != 0u) {
        leaf = tr.cur;
        tr.cur = tr.sp > 0u ? (deep ? stack.pop(--tr.sp) : stack.pop_lds(--tr.sp)) : kInvalid;
    }
}
// ... and back (on the way out of a traversal loop): the lane stands at its postponed leaf, where it went on to is the top of its stack
LR_D void trav_unpostpone(const TraversalStack &stack, TravState &tr, uint32_t &leaf) {
    if (leaf != kInvalid) {
        if (tr.cur != kInvalid) { stack.push(tr.sp++, tr.cur); }
        tr.cur = leaf;
        leaf = kInvalid;
    }
}
// ONE ITERATION of a traversal loop for the wave (LR_FUSED_FETCH, round 4): every lane with a ray stands at an inner node or at a leaf, and
// both kinds of step begin with a dependent gather -- the node's packet, the leaf's triangle.  Rounds 1-3 ran "node step, then leaf
// step": two round trips to memory in a row, the second one (1450-2050 cycles by the section probes, for 80 VALU instructions) on
// behalf of the 0.13 of the lanes that stand at a leaf.  Here both gathers are issued up front and waited for ONCE; then the leaf
// lanes test their triangle and the node lanes their packet.
// MEASURED, and OFF by default (profiles/r04e_fused_fetch.txt, Msamples/s with / without): one path per lane C2 861 / 853 (256 spp),
// 908 / 895 (1024), C3 810 / 803, C4 894 / 881, C5 421 / 417 -- but the Cornell box 3410 / 3757; the pool kernels C2 892 / 898, C3 890 /
// 897, C4 904 / 898, C5 447 / 449, Cornell 3018 / 3230.  One per cent on the kernels that no longer run the large scenes, nothing on the
// ones that do: the wave's own round trips are not what the loop waits for -- the SIMD's other three waves cover them either way.
#ifndef LR_FUSED_FETCH
#define LR_FUSED_FETCH 0
#endif
template<bool COUNT, bool ALPHA>
LR_D void trav_step_fused(const TraversalStack &stack, const TravLane &tl, TravState &tr, f3 inv, bool live, bool deep, TraceStats &stats) {
    const auto is_inner = live && tr.cur != kInvalid && (tr.cur & kLeafFlag) == 0u;
    const auto is_leaf = live && tr.cur != kInvalid && (tr.cur & kLeafFlag) != 0u;
    const auto any_inner = __any(is_inner);
    if (any_inner) { trav_node_fetch(stack, tl, tr, is_inner); }
    LeafTriangle tri;
    if (is_leaf) { tri = trav_leaf_fetch(tl, tr.cur); }
    trav_fetch_wait();
    if (is_leaf) { trav_leaf_step<COUNT, ALPHA>(stack, tl, tr, deep, stats, nullptr, &tri); }
    if (any_inner) { trav_node_step<COUNT, true>(stack, tl, tr, inv, is_inner, deep, stats); }
}

// one iteration's leaf work of the wave; returns with `leaf` tested where the wave's leaf step was due
template<bool COUNT, bool ALPHA>
LR_D void trav_leaves(const TraversalStack &stack, const TravLane &tl, TravState &tr, uint32_t &leaf, bool live, bool deep, TraceStats &stats) {
#if LR_LEAF_BATCH > 0
    trav_postpone(stack, tr, leaf, live, deep);
    const auto holds = live && leaf != kInvalid;
    const auto waiting = __ballot(holds);
    if (waiting == 0ull) { return; }
    if (static_cast<uint32_t>(__popcll(waiting)) < static_cast<uint32_t>(LR_LEAF_BATCH) && __any(live && tr.cur != kInvalid && (tr.cur & kLeafFlag) == 0u)) { return; }
    if (holds) { trav_leaf_step<COUNT, ALPHA, true>(stack, tl, tr, deep, stats, &leaf); }
#elif defined(LR_LEAF_WAIT) && LR_LEAF_WAIT > 1// (the plain form: lanes at a leaf WAIT until LR_LEAF_WAIT of them do, or no lane has an inner node to go on with)
    const auto at_leaf = live && tr.cur != kInvalid && (tr.cur & kLeafFlag) != 0u;
    const auto n_leaf = static_cast<uint32_t>(__popcll(__ballot(at_leaf)));
    if (n_leaf == 0u) { return; }
    if (n_leaf < static_cast<uint32_t>(LR_LEAF_WAIT) && __any(live && tr.cur != kInvalid && (tr.cur & kLeafFlag) == 0u)) { return; }
    if (at_leaf) { trav_leaf_step<COUNT, ALPHA>(stack, tl, tr, deep, stats); }
#else
    if (live && tr.cur != kInvalid && (tr.cur & kLeafFlag) != 0u) { trav_leaf_step<COUNT, ALPHA>(stack, tl, tr, deep, stats); }
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
    auto leaf = kInvalid;// the lane's postponed leaf (LEAF BATCHING above)
    for (;;) {
#ifdef LR_TRACE_PROBE// (section cycles of the loop in the counting build: node step -> nodes_empty, end of iteration -> trace_steps_starved; lane 0 reports)
        if (COUNT) { stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u; }
        const auto probe_t0 = __builtin_readcyclecounter();
#else
        if (COUNT) { stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u, stats.steps_starved += idle_at_entry ? 1u : 0u; }
#endif
        auto live = ALPHA ? (tr.phase == kPhaseShadow || tr.phase == kPhaseClosest) : tr.phase != kPhaseIdle;// (not parked)
        auto is_inner = live && tr.cur != kInvalid && !(tr.cur & kLeafFlag);
        // one WAVE-LEVEL test per iteration decides whether any lane could touch the HBM overflow area of the stack in this iteration
        // (a lane at an inner node pushes at most three entries); if none can -- nearly always -- every push and pop of the
        // iteration is a bare LDS access instead of a compare + branch + access per entry (round 3: +1 %)
        const auto deep = __any(live && tr.sp + 3u > kStackLds);
#if LR_FUSED_FETCH && LR_LEAF_BATCH == 0
        trav_step_fused<COUNT, ALPHA>(stack, tl, tr, inv, live, deep, stats);
#else
        if (__any(is_inner)) { trav_node_step<COUNT>(stack, tl, tr, inv, is_inner, deep, stats); }
#endif
#ifdef LR_TRACE_PROBE
        const auto probe_t1 = __builtin_readcyclecounter();
#endif
#if !(LR_FUSED_FETCH && LR_LEAF_BATCH == 0)
        trav_leaves<COUNT, ALPHA>(stack, tl, tr, leaf, live, deep, stats);
#endif
#ifdef LR_TRACE_PROBE
        const auto probe_t2 = __builtin_readcyclecounter();
#endif
        // ---- ray finished: switch from the shadow ray to the closest-hit ray, or go idle
        if (live && tr.cur == kInvalid && leaf == kInvalid) {
            if (tr.phase == kPhaseShadow && has_next) {
                trav_begin(tr, next_closest, kPhaseClosest);
                inv = safe_inverse(tr.d);
            } else {
                tr.phase = kPhaseIdle;
            }
        }
        auto in_flight = __ballot(tr.phase != kPhaseIdle);
#ifdef LR_TRACE_PROBE
        if (COUNT && (threadIdx.x & 63u) == 0u) {
            stats.nodes_empty += static_cast<uint32_t>(probe_t1 - probe_t0);
            stats.steps_starved += static_cast<uint32_t>(__builtin_readcyclecounter() - probe_t2);
        }
#endif
        if (in_flight == 0ull) { break; }
        if (ALPHA && __any((tr.phase & kPhasePendingAlpha) != 0u)) { break; }
        auto finished = __ballot(tr.phase == kPhaseIdle && !idle_at_entry);
        if (__popcll(finished) >= refill) { break; }
    }
    trav_unpostpone(stack, tr, leaf);
}

}// namespace lrd
