// megapool_kernel.h — round 4: the persistent-threads megakernel with a PATH POOL: two path contexts per lane.
//
// megapath_kernel.h (rounds 1-3) binds one path to one lane: a lane whose ray has finished waits, idle, until enough of its
// neighbours have finished too (the refill threshold), and the shading block then runs for the ~half of the wave that has
// something to shade.  Measured on the C2 stand-in (profiles/archive/r03am_*): 56 % of the traversal loop's lane-steps and 47 % of the
// shading block's lanes did useful work while the VALU pipes were ~full -- the machine was busy computing masked-off lanes.
//
// Here a wave holds 128 paths, two CONTEXTS per lane.  While one context's rays are traced the other one waits -- for the shading
// block with its hit, or for the lane with the rays of its next job -- and when a lane's job ends it switches to its other
// context's rays INSIDE the traversal loop, registers to registers:
//   * what the traversal loop needs of a waiting context -- the job's rays (shadow ray + next path segment, 15 words) or its hit (4
//     words) -- lives in REGISTERS, in fixed roles: `cur` is the context whose ray the lane traces, `oth` the one that waits, and a
//     lane that goes on to its other context's job EXCHANGES the two with 19 v_swap_b32: no memory, no LDS, no queue.  Rays that
//     ended wait until LR_POOL_TURNOVER_LANES lanes have one: the turnover code then runs once for all of them;
//   * what only the shading block needs of a path -- throughput, radiance, NEE term, bsdf pdf, depth, sampler position, pixel: 16
//     words -- lives in a per-thread record in global memory, [context][quad][thread]: four coalesced 16-byte loads when the
//     context is shaded, four stores when it leaves, ONE round trip per shading batch;
//   * the wave leaves the traversal loop when LR_POOL_SHADE_LANES lanes hold a context to shade (or LR_POOL_IDLE_LANES of them have
//     nothing left to trace); a lane shades ONE context per batch -- its other one -- and EVERYTHING ELSE LEAVES THE REGISTERS for the
//     duration of the block: the current context's rays go to the packet staging area of the LDS (idle outside the traversal
//     loop), what is left of the ray in flight (hit so far, t_max, node, phase / stack depth) on top of the lane's own traversal
//     stack.  This is what makes the scheme pay: with those six words in registers the block spilled 37 VGPRs on its hot path, the
//     scratch traffic evicted the BVH's top levels from the 32 KB vector L1, and a node step took 4200 cycles instead of 2100
//     (HISTORY.md appendix, section 4.1c; profiles/archive/r04c_*, r04g_*).
// Measured (profiles/r04_final_schedulers.txt, kernel time of the one-path-per-lane kernel / this one, same build, same box): C2 1.08
// at 1024 spp, C3 1.18, C4 1.06, C5 (wavefront mode) 1.10 at 64 spp; a Cornell box 0.88 -- lrhip.hip: wants_pool picks this kernel
// from ~100 thousand triangles up, earlier for deep paths at few samples per pixel, later for shallow ones (lrhip.hip: pool_auto_triangles; profiles/r05j_scheduler_sweep.txt).
//
// MEASURED AND NOT KEPT (profiles/archive/r04a_*): 128 path SLOTS per wave shared by all lanes -- records of 128 B in global memory, ray and
// shade queues of slot numbers in LDS, lanes fetching their next job from the ray queue inside the loop.  It filled the lanes (0.93 /
// 0.79, 23 % fewer VALU instructions per sample, films equal to 6e-8) and was no faster: C2 830 against 854 Msamples/s at 256 spp,
// C1 2420 against 4820.  Every job turnover read and wrote slot records whose lines the 4 MiB L2 of an XCD had long dropped (8 MiB of
// slots per XCD, the BVH streaming through): 2.4 TB/s of extra fabric traffic, and -- loads return in order -- every one of those
// reads held up the whole wave's next packet fetch (wave cycles waiting 0.50 -> 0.63).  Requesting the rays an iteration ahead
// (+5 %), full-line stores through the LDS (-4 %), non-temporal hints (-29 %), fewer waves (-19 %) did not change the picture: the
// rays must not leave the chip.
//
// FILM.  With lanes no longer bound to pixels and WORK ITEMS OVERLAPPING inside a wave (when an item's sample queue runs dry the wave
// takes the next item at once; the old item's last paths finish beside the new item's first -- no drain), the order of a pixel's
// adds is no longer a function of its item alone.  So the sums are made order-independent instead: radiance is accumulated in 64-bit
// FIXED POINT (dev_wavefront.h: radiance_to_fixed), per wave in an LDS copy of the item's tile (ds_add_u64), flushed to the frame's
// fixed-point sums (WfArgs::accum, global atomics) when the wave leaves the item; a straggler that finishes after its item was
// flushed adds to the frame's sums directly.  Integer adds are associative: films are bit-reproducible run to run, under any tile
// sharding, any grid size and any work-item partition -- a stronger guarantee than rounds 1-3 gave (identical chunking required).
// Sample counts: the item's samples are counted at the flush (film.w += samples per pixel, exact in fp32), a rejected sample
// (NaN / Inf, color.cpp:110-113) takes its count back.
//
// The estimator is the reference's MegakernelPathTracingInstance::Li (src/integrators/mega_path.cpp:49-156) exactly as in
// megapath_kernel.h -- the shading block below is that file's, working on a context instead of the lane -- and the wavefront-mode
// roles of that kernel (kFeatWf camera pass: heavy hits parked for heavy_kernel.h; kFeatCont: continuation records instead of
// camera samples) carry over unchanged.
#pragma once
#include "dev_wavefront.h"

namespace lrd {

#ifndef LR_MIN_WAVES
#define LR_MIN_WAVES 4
#endif
// -DLR_POOL_SINGLE: the lane's second context never takes a path (a diagnostic: this kernel's machinery under the round 1-3 scheduling)
#ifndef LR_POOL_SHADE_LANES
#define LR_POOL_SHADE_LANES 56// lanes with a context to shade that end the traversal loop
#endif
#ifndef LR_POOL_IDLE_LANES
#define LR_POOL_IDLE_LANES 16 // ... or lanes with a context to shade and nothing left to trace
#endif
#ifndef LR_POOL_TURNOVER_LANES
// A ray that ended waits until this many lanes have one (or no lane has anything left to traverse): the turnover code of the loop --
// results into the context, exchange of the contexts, the next ray into the traversal state, 1 / d -- then runs once for all of them,
// every seventh iteration or so instead of in seven of ten.  C2, 256 spp: 918 Msamples/s at 1 (no waiting), 924 at 5 (on the build
// before), 955 at 8, 954 at 12, 945 at 16, 916 at 24 (profiles/archive/r04f_turnover_and_stack.txt).  Round 5, the loop lighter: 12 is 0.3-0.6 % ahead of
// 8 in four A/Bs (r05a, r05d, r05l, r05r), 6 and 4 behind.
#define LR_POOL_TURNOVER_LANES 12
#endif
#ifndef LR_POOL_PARK_ON_STACK
#define LR_POOL_PARK_ON_STACK 1// the five parked words of the ray in flight go on top of the lane's traversal stack (0: an LDS area of their own, LR_STACK_LDS <= 11)
#endif
#ifndef LR_POOL_FUSED_FETCH
// The pool kernels request BOTH gathers of an iteration -- the packets of the lanes at inner nodes, the triangles of the lanes at leaves -- up
// front and wait once (dev_trace.h: trav_iteration<.., FUSED>); a lane the node step sends to a leaf tests it in the next iteration (5 %
// more iterations, each with one round trip instead of two in a row).  Round 4 measured this at +-1 % and left it off; with the loop a
// fifth lighter the round trips weigh more: C2 1022.5 -> 1043.2 at 1024 spp, C3 1024.6 -> 1033.3, C4 1015.4 -> 1031.4,
// films BIT-IDENTICAL (both flows share trav_leaf_test; profiles/r05o_pool_fused_fetch.txt).  The one-path-per-lane kernels keep the serial
// flow: their scenes' walks are a handful of steps out of the L1, and the extra iterations cost the Cornell box 7 %.  So do the ALPHA pool
// kernels (alpha-tested traversal): fused, the alpha-only stand-in runs at 750 instead of 859 Msamples/s and C5 at 504 instead of 567
// (profiles/r05q_fused_fetch_alpha_kernels.txt); the wavefront passes WITHOUT alpha gain 0.7 %.
#define LR_POOL_FUSED_FETCH 1
#endif
#ifndef LR_POOL_FUSED_ALPHA
#define LR_POOL_FUSED_ALPHA 0// (the ALPHA pool kernels under the fused flow: see above)
#endif
#ifndef LR_POOL_STATE_LEAN
#define LR_POOL_STATE_LEAN 1
#endif
#ifndef LR_POOL_RAY_INIT
#define LR_POOL_RAY_INIT if (!mine)
#endif
constexpr uint32_t kPoolSlots = 128u;// paths per wave: two contexts per lane

// ---- path state of one context in global memory: kPoolQuads float4 at [context][quad][thread]
//   0  nee.xyz | pdf_bsdf        1  beta.xyz | depth (16) | pixel in tile (6) << 16
//   2  Li.xyz | sampler word 0   3  pixel index (frame) | work item of the path | sampler words 1-2
//   4  sampler word 3 (generic sampler only)
template<bool GENERIC>
constexpr uint32_t pool_quads() { return GENERIC ? 5u : 4u; }

// ---- one of a lane's two path contexts as the traversal loop sees it (registers)
enum : uint32_t {
    kCtxShadow = 1u, kCtxClosest = 2u,     // rays of the job still to trace
    kCtxHadShadow = 4u, kCtxHadClosest = 8u,// what the job held (the shading block consumes the results accordingly)
    kCtxDone = 16u,                        // the job is traced: hit / occluded are valid, the context waits for the shading block
    kCtxOccluded = 32u,
    kCtxSide = 64u,                        // which of the thread's two state records belongs to this context
    kCtxOpen = 128u,                       // holds a path (else: empty, takes the next sample)
};
struct PathCtx {
    f3 so, sd;        // shadow ray (t_min 0)
    float s_tmax;
    f3 no, nd;        // the path's next segment; kept until the shading block reads them back as the segment that was traced
    float n_tmin, n_tmax;
    uint32_t tri;     // hit of the segment (kInvalid: miss)
    float u, v;
    uint32_t flags;
};
// the first ray of a context's job into the lane's traversal state
LR_D void ctx_start(PathCtx &c, TravState &tr) {
    // (written so that the compiler moves under EXEC instead of selecting: eight v_cndmask_b32_e32 in a row on one VCC cost 19 cycles
    // EACH on this chip -- tools/valu_peak2.hip, profiles/archive/r04h_cndmask_forms.json: 18.7 back to back, 2.2 with other VALU work
    // between them, 4.2 in the VOP3 form -- and a job turnover was mostly that: C2 956 -> 965 Msamples/s)
    tr.o = c.no, tr.d = c.nd, tr.t_min = c.n_tmin, tr.t_max = c.n_tmax;
    tr.phase = kPhaseClosest;
    if ((c.flags & kCtxShadow) != 0u) {
        asm volatile("");
        tr.o = c.so, tr.d = c.sd, tr.t_min = 0.f, tr.t_max = c.s_tmax;
        tr.phase = kPhaseShadow;
        c.flags &= ~kCtxShadow;
    } else {
        asm volatile("");
        c.flags &= ~kCtxClosest;
    }
    tr.cur = 0u, tr.sp = 0u;// root
    tr.hit.tri = kInvalid, tr.hit.u = 0.f, tr.hit.v = 0.f;
    tr.occluded = false;
}

// The lane's two contexts are `cur` -- the one whose ray the lane traces (or traced last) -- and `oth`, the one that waits.  A lane
// that goes on to its other context's job EXCHANGES the two: nineteen v_swap_b32, no copy through a third register, no per-field
// select on a "which one" bit in the loop (the first form of this kernel: 60 instructions per turnover where this one has 20).
LR_D void swap_words(float &x, float &y) { asm volatile("v_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
LR_D void swap_words(uint32_t &x, uint32_t &y) { asm volatile("v_swap_b32 %0, %1" : "+v"(x), "+v"(y)); }
LR_D void ctx_swap(PathCtx &a, PathCtx &b) {
    swap_words(a.so.x, b.so.x), swap_words(a.so.y, b.so.y), swap_words(a.so.z, b.so.z), swap_words(a.s_tmax, b.s_tmax);
    swap_words(a.sd.x, b.sd.x), swap_words(a.sd.y, b.sd.y), swap_words(a.sd.z, b.sd.z);
    swap_words(a.no.x, b.no.x), swap_words(a.no.y, b.no.y), swap_words(a.no.z, b.no.z), swap_words(a.n_tmin, b.n_tmin);
    swap_words(a.nd.x, b.nd.x), swap_words(a.nd.y, b.nd.y), swap_words(a.nd.z, b.nd.z), swap_words(a.n_tmax, b.n_tmax);
    swap_words(a.tri, b.tri), swap_words(a.u, b.u), swap_words(a.v, b.v), swap_words(a.flags, b.flags);
}

LR_D uint32_t lane_rank(unsigned long long mask) {// lanes of `mask` below this one
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32u), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

constexpr uint32_t kCtxRays = kCtxShadow | kCtxClosest;
// flags of a context that the shading block has work for: a traced job, or (while the launch has samples left) no path at all
LR_D bool ctx_shadeable(uint32_t flags, bool samples_left) {
#ifdef LR_POOL_SINGLE// (experiment: the second context never takes a path -- the round 1-3 scheduling in this kernel's clothes)
    if ((flags & kCtxSide) != 0u) { samples_left = false; }
#endif
    return (flags & kCtxDone) != 0u || (samples_left && (flags & kCtxOpen) == 0u);
}
// whether the shading block is due: enough lanes hold a context for it, or enough of them have nothing left to trace, or nothing is in
// flight at all.  Only an idle lane's current context can be shaded.
LR_D bool pool_shade_due(uint32_t phase, uint32_t cur_flags, uint32_t oth_flags, bool samples_left) {
    const auto idle = phase == kPhaseIdle;
    const auto s = lr_ballot(ctx_shadeable(oth_flags, samples_left) || (idle && ctx_shadeable(cur_flags, samples_left)));
    if (s == 0ull) { return false; }
    const auto idle_mask = lr_ballot(idle);
    return static_cast<uint32_t>(__popcll(s)) >= static_cast<uint32_t>(LR_POOL_SHADE_LANES) ||
           static_cast<uint32_t>(__popcll(s & idle_mask)) >= static_cast<uint32_t>(LR_POOL_IDLE_LANES) || idle_mask == ~0ull;
}

// The traversal loop of the pool kernel: dev_trace.h's node and leaf steps; a lane whose job ends switches to its other context's
// rays without leaving the loop.  Returns false when the shading block is due or nothing is in flight; ALPHA: true when LR_ALPHA_BATCH lanes
// hold a candidate hit for the alpha test (dev_shade.h: resolve_pending_alpha) and the wave is to come straight back.  Must be called by all 64 lanes.
template<bool COUNT, bool ALPHA>
LR_D bool pool_trace(const DScene &scene, const TraversalStack &stack, TravState &tr, PathCtx &cur, PathCtx &oth, bool samples_left, TraceStats &stats) {
    const auto tl = TravLane::make(scene, stack);
    auto inv = safe_inverse(tr.d);
    auto spb = tl.spb_of(tr.sp);
    if (tr.phase == kPhaseIdle) { tr.cur = kCurIdle; }// (inside the loop a lane's state is read off `cur`: dev_trace.h, TravLane)
    auto for_alpha = false;// (ALPHA: the wave leaves for the alpha tests of its parked candidates and comes straight back)
#ifdef LR_STALL_PROBE
    if (COUNT) { probe_start(stats); }
#endif
    for (;;) {
        LR_MARK(kProbeTail, tr.cur);// (dev_trace.h, THE STALL PROBE: what the end of the previous iteration took)
        if (COUNT) {
            stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u;
#ifndef LR_TRACE_PROBE
            stats.steps_starved += tr.phase == kPhaseIdle && ((cur.flags | oth.flags) & kCtxOpen) == 0u ? 1u : 0u;// no path left to hold
#endif
        }
#ifdef LR_TRACE_PROBE// (section cycles of the loop in the counting build: the iteration's walk -> nodes_empty, end of iteration -> trace_steps_starved; lane 0 reports)
        const auto probe_t0 = __builtin_readcyclecounter();
#endif
        trav_iteration<COUNT, ALPHA, LR_POOL_FUSED_FETCH != 0 && (!ALPHA || LR_POOL_FUSED_ALPHA != 0)>(stack, tl, tr, spb, inv, stats);
#ifdef LR_TRACE_PROBE
        const auto probe_t1 = __builtin_readcyclecounter();
        const auto probe_t2 = probe_t1;
#endif
        if (ALPHA && alpha_tests_due(tr.phase, tr.cur)) { for_alpha = true; break; }
        // ---- rays that ended: the job's next ray, the other context's job, or idle -- once LR_POOL_TURNOVER_LANES lanes wait with one, or no
        // lane has anything left to traverse.  (Nothing below changes in an iteration without a turnover: the exit tests run behind one.)
        const auto ended = tr.cur == kInvalid;
        const auto n_ended = static_cast<uint32_t>(__popcll(lr_ballot(ended)));
#ifdef LR_TRACE_PROBE
        if (COUNT && (threadIdx.x & 63u) == 0u) {
            stats.nodes_empty += static_cast<uint32_t>(probe_t1 - probe_t0);
            stats.steps_starved += static_cast<uint32_t>(__builtin_readcyclecounter() - probe_t2);
        }
#endif
        if (n_ended == 0u) { continue; }
        if (n_ended < static_cast<uint32_t>(LR_POOL_TURNOVER_LANES) && lr_any(tr.cur < kCurParked)) { continue; }
        if (ended) {
            if (tr.phase == kPhaseShadow) {
                if (tr.occluded) { cur.flags |= kCtxOccluded; }
            } else {
                cur.tri = tr.hit.tri, cur.u = tr.hit.u, cur.v = tr.hit.v;
            }
            tr.phase = kPhaseIdle, tr.cur = kCurIdle;
            if ((cur.flags & kCtxRays) == 0u) {// the job is complete: on to the other context's, if it waits with one
                cur.flags |= kCtxDone;
                if ((oth.flags & kCtxRays) != 0u) { ctx_swap(cur, oth); }
            }
            if ((cur.flags & kCtxRays) != 0u) {
                ctx_start(cur, tr);
                inv = safe_inverse(tr.d);
                spb = tl.lds_base;
            }
        }
        // (what the shading block has to do only changes when a lane runs out of rays: the tests are skipped otherwise -- round 4: +1.9 % on C2)
        if (!lr_any(ended && tr.cur == kCurIdle)) { continue; }
        if (lr_ballot(tr.cur != kCurIdle) == 0ull) { break; }
        if (pool_shade_due(tr.phase, cur.flags, oth.flags, samples_left)) { break; }
    }
    LR_MARK(kProbeTail, tr.cur);
    prio_shade();
    tr.sp = tl.sp_of(spb);
    return for_alpha;
}

// PADDED kernels: a stretch of a vertex's draws -- one number, two numbers, and a fourth where `four` -- is ONE real call, so that the hashes' and the
// permutation's temporaries are not the shading block's: <20482> 16 -> 7 spilled VGPRs, C2 under PaddedSobol 974 -> 1004 Msamples/s at 256 spp, the
// camera class 1067 -> 1111, films bit-identical (profiles/r06za_padded_draws_out_of_line.txt; LR_PADDED_DRAWS_OUT_OF_LINE=0 restores the inline draws).
// The same for the run-time generic sampler (state in; numbers and state out: eight return registers): Sobol 886 -> 880, PCG32 980 -> 968 -- not kept.
#ifndef LR_PADDED_DRAWS_OUT_OF_LINE
#define LR_PADDED_DRAWS_OUT_OF_LINE 1
#endif
struct PaddedDraws {
    float a, b, c, d;
};
[[maybe_unused]] static __device__ __noinline__ PaddedDraws padded_draws(const DScene *scene, uint32_t sample_index, uint32_t pixel, uint32_t dimension, bool four) {
    PathSampler<true> sampler{};
    uint32_t words[kWfSamplerWordsMax] = {sample_index, 0u, dimension, pixel};
    sampler.restore(*scene, words);
    PaddedDraws r;
    r.a = sampler.next_1d();
    const auto u = sampler.next_2d();
    r.b = u.x, r.c = u.y;
    r.d = four ? sampler.next_1d() : 0.f;
    return r;
}

// LR_LIGHT_OUT_OF_LINE: the light sample of a vertex -- selection, the chain light instance -> instance -> alias table -> triangle -> vertices,
// emission, environment sampling, the shadow ray -- as ONE real call with its thirteen inputs by value, in the pool kernels that hold the Disney
// closure: their shading block is the one that spills (<12308> 67 -> 44 VGPRs), and the call's temporaries are not its allocation's any more.
// Camera class 1166 -> 1227 Msamples/s at 64 spp, films bit-identical; the kernels without Disney LOSE (bedroom class <4100> 1092 -> 1015, 12 -> 27
// spilled; the kitchen class' wavefront passes 640 -> 627) and keep the inline form (profiles/r06zd_light_sample_out_of_line.txt).
#ifndef LR_LIGHT_OUT_OF_LINE
#if defined(LR_VARIANT) && ((LR_VARIANT) & 16) && !((LR_VARIANT) & (96 | 256))
#define LR_LIGHT_OUT_OF_LINE 1
#else
#define LR_LIGHT_OUT_OF_LINE 0
#endif
#endif
template<bool ENV>
[[maybe_unused]] static __device__ __noinline__ LightPick sample_one_light_call(const DScene *scene, f3 p, f3 ng, f3 ns, uint32_t offset_bits, float u0, float u1, float u2) {
    SurfacePoint it{};
    it.p = p, it.ng = ng, it.shading.n = ns, it.offset_bits = offset_bits;
    return sample_one_light<ENV>(*scene, it, u0, f2{u1, u2});
}

template<uint32_t F>
__global__ __launch_bounds__(kBlockThreads, LR_MIN_WAVES) void megapool_kernel(DScenePtr scene_ptr, RenderArgs args) {
    const DScene &scene = *(const DScene *)scene_ptr;
    constexpr bool COUNT = (F & kFeatCount) != 0u, PCG = (F & kFeatGeneric) != 0u, ENV = (F & kFeatEnv) != 0u,
                   ALPHA = (F & kFeatAlpha) != 0u, DISNEY = (F & kFeatDisney) != 0u, WF = (F & kFeatWf) != 0u, CONT = (F & kFeatCont) != 0u;
    static_assert((F & kFeatPool) != 0u, "a pool variant");
    static_assert(kPoolParkedWords == 5u, "the shading block parks five words on top of a lane's stack (below): lrhip_upload_scene bounds the BVH depth with this constant");
    static_assert((F & (kFeatMix | kFeatLayered | kFeatAux | kFeatVpt | kFeatNest)) == 0u, "the pool scheduler exists for the lean kernels (closures inline)");
    static_assert(!WF || !DISNEY, "a wavefront variant is a lean kernel: the heavy closures live in heavy_kernel.h");
    static_assert(!CONT || WF, "the continuation pass exists in wavefront mode only");
    constexpr uint32_t SAMPLER_WORDS = PathSampler<PCG>::kSavedWords;
    // PADDED (round 6): the generic sampler's kind known at compile time to be PaddedSobol (LR_ONLY_SAMPLER).  Its stream position is (sample
    // index, pixel, dimension): the first two never change along a path -- written with quad 3 when the path starts -- and the dimension is a
    // function of the depth (two for the pixel, two for a thin lens, six per vertex, one more from the Russian-roulette depth on): nothing
    // is written back where a vertex's numbers are drawn, and the record is four quads like the Independent sampler's.  In wavefront mode a parked
    // path takes the position along in the generic sampler's four words (sample index, 0, dimension, pixel): the heavy kernels draw a vertex's numbers
    // in the same pattern, so the dimension a continuation record comes back with IS the derived one and only (sample index, pixel) return to the record.
#ifdef LR_ONLY_SAMPLER
    constexpr bool PADDED = PCG && (LR_ONLY_SAMPLER) == LR_SAMPLER_PADDED_SOBOL;
#else
    constexpr bool PADDED = false;
#endif
    static_assert(PADDED == ((F & kFeatPadded) != 0u) || (F & kFeatPadded) == 0u, "a kFeatPadded kernel: generic sampler, LR_ONLY_SAMPLER = PaddedSobol");
    constexpr uint32_t QUADS = PADDED ? 4u : pool_quads<PCG>();
    __shared__ uint32_t s_stack[kStackLds * kBlockThreads];
    __shared__ float4 s_stage[kWavesPerBlock * kStageWave];// 4 KiB of node packets per wave
#if LR_POOL_PARK_ON_STACK == 0
    // what a lane keeps of its ray in flight across the shading block: five words, [word][thread] (the LDS an 11-entry stack leaves, see below)
    __shared__ uint32_t s_park[5u * kBlockThreads];
    static_assert(kStackLds + 5u <= 16u, "the parking area is carved out of the round 1-3 stack: compile such pool variants with LR_STACK_LDS <= 11");
#endif
    __shared__ unsigned long long s_film[CONT ? 1u : kWavesPerBlock * 192u];// per-wave tile accumulators, fixed point [pixel][rgb]
#ifdef LR_STALL_PROBE
    __shared__ uint32_t s_probe[kWavesPerBlock * kProbeWords];
    if ((threadIdx.x & 63u) < kProbeWords) { s_probe[(threadIdx.x >> 6u) * kProbeWords + (threadIdx.x & 63u)] = 0u; }// (every wave its own words)
    // one wave in 36 takes the timestamps and reports (dev_trace.h: THE STALL PROBE): wave (block / 9) % 4 of every ninth block -- spread over the
    // XCDs (blocks go round the eight of them) and over the SIMDs.  (With EVERY wave adding its shading sections to the four global counters after
    // every batch -- 80 million atomics on one cache line per frame -- one L2 channel was the bottleneck of the whole kernel: every gather that
    // crossed it queued behind them, and the probed kernel ran 3.5 times slower in all its waves.)
    const auto probe_wave = (F & kFeatCount) != 0u && (blockIdx.x % 9u) == 0u && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6u) == ((blockIdx.x / 9u) & 3u);
#endif
    const auto tid = threadIdx.x;
    const auto lane = tid & 63u;
    const auto gtid = blockIdx.x * kBlockThreads + tid;
    const auto wave_in_block = __builtin_amdgcn_readfirstlane(tid >> 6u);
    TraversalStack stack{s_stack + tid, args.spill + gtid, args.total_threads, s_stage + wave_in_block * kStageWave};
    const auto film_tile = s_film + (CONT ? 0u : wave_in_block * 192u);
    // path state of this thread's two contexts: quad q of context c at args.pool[(c * QUADS + q) * total_threads + gtid]
    const auto state_of = [&](uint32_t side, uint32_t quad) { return args.pool + static_cast<size_t>(side * QUADS + quad) * args.total_threads + gtid; };
    // (as 16-byte quads.  Word by word -- no register tuples for the allocator to place -- was tried: 46 -> 60 spilled VGPRs)
    const auto state_load = [&](uint32_t side, uint32_t quad) { return *state_of(side, quad); };
    const auto state_store = [&](uint32_t side, uint32_t quad, float4 v) { *state_of(side, quad) = v; };
    DCounters local{};
    const auto t_wave = COUNT ? __builtin_readcyclecounter() : 0ull;

    // the continuation pass (CONT) works through the records the heavy kernel wrote this round (megapath_kernel.h: same item sizing)
    const auto cont_total = CONT ? min(scene.wf.counts[kWfCountCont], scene.wf.capacity) : 0u;
    const auto cont_waves = gridDim.x * kWavesPerBlock;
    const auto item_records = CONT ? min(kWfItemRecords, max(64u, ((cont_total + cont_waves - 1u) / cont_waves + 63u) & ~63u)) : 1u;
    const auto item_count = CONT ? (cont_total + item_records - 1u) / item_records : args.item_count;
    const auto cont_queue = wf_cont_queue(scene);

    // ---- the wave's work item (wave-uniform) and its queues
    auto item = kInvalid;          // current work item; kInvalid before the first and after the last
    auto items_left = true;
    auto q_next = 0u, q_total = 0u;// the item's sample queue: k = 64 * (s - s_begin) + pixel_in_tile (CONT: record item * item_records + k)
    auto s_begin = 0u, s_count = 0u, tx = 0u, ty = 0u;
    if (!CONT) { film_tile[lane * 3u] = 0ull, film_tile[lane * 3u + 1u] = 0ull, film_tile[lane * 3u + 2u] = 0ull; }
    // ---- the lane: its ray in flight, the context the ray belongs to (`cur`) and the lane's other context (`oth`)
    TravState tr{};
    tr.phase = kPhaseIdle;
    PathCtx cur{}, oth{};
    cur.flags = 0u, oth.flags = kCtxSide;

    // the wave leaves its work item: the tile's sums join the frame's, every sample of the item is counted
    auto flush_tile = [&]() {
        if (CONT || item == kInvalid) { return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const auto a0 = film_tile[lane * 3u], a1 = film_tile[lane * 3u + 1u], a2 = film_tile[lane * 3u + 2u];
        film_tile[lane * 3u] = 0ull, film_tile[lane * 3u + 1u] = 0ull, film_tile[lane * 3u + 2u] = 0ull;
        const auto wx = tx * 8u + (lane & 7u), wy = ty * 8u + (lane >> 3u);
        if (wx < scene.camera.width && wy < scene.camera.height) {
            const auto index = wy * scene.camera.width + wx;
            const auto acc = scene.wf.accum + static_cast<size_t>(index) * 3u;
            if (a0 != 0ull) { atomicAdd(acc + 0, a0); }
            if (a1 != 0ull) { atomicAdd(acc + 1, a1); }
            if (a2 != 0ull) { atomicAdd(acc + 2, a2); }
            if (s_count != 0u) { atomicAdd(&args.film[index].w, static_cast<float>(s_count)); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // the wave's next work item (the tile of the one it leaves goes to the film first)
    auto take_item = [&]() {
        flush_tile();
        item = next_item(CONT ? scene.wf.counts + kWfWorkCont : args.work_counter, item_count, lane);
        q_next = 0u, q_total = 0u, s_count = 0u;
        if (item == kInvalid) {
            items_left = false;
            return;
        }
        if (CONT) {
            q_total = min(item_records, cont_total - item * item_records);
        } else {
            const auto range = item_range(args, item);
            const auto tile = args.tile_begin + range.tile_index * args.tile_stride;
            ty = tile / args.tiles_x, tx = (tile - ty * args.tiles_x + ty) % args.tiles_x;// row ty is rotated by ty (lrhip.h)
            s_begin = range.s_begin, s_count = range.s_end > range.s_begin ? range.s_end - range.s_begin : 0u;
            q_total = s_count * 64u;
        }
    };

    for (;;) {
        // ==== (A) shading batches, while they are due (pool_shade_due: enough lanes hold a context to shade)
        for (;;) {
            const auto samples_left = items_left || q_next < q_total;
            if (!pool_shade_due(tr.phase, cur.flags, oth.flags, samples_left)) { break; }
            const auto t_shade = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (ALPHA) { tr.pend_t = 0.f, tr.pend_u = 0.f, tr.pend_v = 0.f; }// (no candidate waits for its alpha test out here: three registers the block need not carry)
            // ---- the context this lane shades is its OTHER one; an idle lane whose other context has nothing for the shading block
            // offers its current one
            if (tr.phase == kPhaseIdle && !ctx_shadeable(oth.flags, samples_left)) { ctx_swap(cur, oth); }
            // (a context that waits with the rays of its job is none of the shading block's business)
#ifdef LR_POOL_SINGLE
            const auto mine = (oth.flags & kCtxDone) != 0u || ((oth.flags & kCtxOpen) == 0u && (oth.flags & kCtxSide) == 0u);
#else
            const auto mine = (oth.flags & kCtxDone) != 0u || (oth.flags & kCtxOpen) == 0u;
#endif
            // ---- Of the lane's two contexts the shading block needs the traced segment and the hit of the one it shades, and it
            // produces that context's next rays in `shadow` / `ray` (which start as the rays the context holds: where it is not `mine`
            // they come out as they went in).  Everything else must leave the registers the shading block is allocated in (314 spilled
            // VGPRs otherwise): the CURRENT context's rays -- origin and direction of the ray in flight are among them -- wait in the
            // packet staging area of the LDS (idle outside the traversal loop; 16 words per lane, [quad][lane]).
            {
                const auto stage = stack.stage;
                stage[lane] = make_float4(cur.so.x, cur.so.y, cur.so.z, cur.s_tmax);
                stage[64u + lane] = make_float4(cur.sd.x, cur.sd.y, cur.sd.z, cur.n_tmin);
                stage[128u + lane] = make_float4(cur.no.x, cur.no.y, cur.no.z, cur.n_tmax);
                // SIX more words leave the registers -- what is left of the ray in flight: its hit so far (one hit besides the shaded context's
                // is alive in a lane: the running one, or an idle lane's current context's finished one), t_max, the node it stands at, and
                // one word of small things (traversal phase 0-2, occluded 3, stack depth 5-11, flags of the current context 12-19, of the
                // other one 20-27).  With these six in registers the block spills 37 VGPRs on its hot path, without them 12 off it (round 4,
                // the measurement behind section 4.1c of DESIGN.md): they are the difference between a traversal loop with and without its
                // L1.  One goes into the spare word of the staging area, five into the LDS the pool kernels' stack gives up (11 entries
                // per lane instead of 16).
                const auto keep_word = (tr.phase & 7u) | (tr.occluded ? 8u : 0u) | (tr.sp << 5u) | (cur.flags << 12u) | (oth.flags << 20u);
                stage[192u + lane] = make_float4(cur.nd.x, cur.nd.y, cur.nd.z, __uint_as_float(keep_word));
#if LR_POOL_PARK_ON_STACK// ... on top of the lane's own traversal stack (the entries above sp; beyond the LDS part: the overflow area in HBM)
                stack.push(tr.sp + 0u, tr.phase != kPhaseIdle ? tr.hit.tri : cur.tri);
                stack.push(tr.sp + 1u, __float_as_uint(tr.phase != kPhaseIdle ? tr.hit.u : cur.u));
                stack.push(tr.sp + 2u, __float_as_uint(tr.phase != kPhaseIdle ? tr.hit.v : cur.v));
                stack.push(tr.sp + 3u, __float_as_uint(tr.t_max));
                stack.push(tr.sp + 4u, tr.cur);
#else
                const auto park = s_park + tid;
                park[0u * kBlockThreads] = tr.phase != kPhaseIdle ? tr.hit.tri : cur.tri;
                park[1u * kBlockThreads] = __float_as_uint(tr.phase != kPhaseIdle ? tr.hit.u : cur.u);
                park[2u * kBlockThreads] = __float_as_uint(tr.phase != kPhaseIdle ? tr.hit.v : cur.v);
                park[3u * kBlockThreads] = __float_as_uint(tr.t_max);
                park[4u * kBlockThreads] = tr.cur;
#endif
                asm volatile("" ::: "memory");// (the values must not be forwarded to the loads at the end of the block: they are to LEAVE the registers)
            }
            // ---- the context's path
            PathSampler<PCG> sampler{};
            Ray ray{}, shadow{};
            LR_POOL_RAY_INIT {
                shadow.o = oth.so, shadow.d = oth.sd, shadow.t_min = 0.f, shadow.t_max = oth.s_tmax;
                ray.o = oth.no, ray.d = oth.nd, ray.t_min = oth.n_tmin, ray.t_max = oth.n_tmax;
            }
            f3 beta = mk3(0.f), Li = mk3(0.f), nee = mk3(0.f);
            auto pdf_bsdf = 1e16f;
            auto dp = 0u, pixel_index = 0u, path_item = kInvalid;// dp: depth | pixel in tile << 16 (one register across the block)
            const auto side = (oth.flags & kCtxSide) != 0u ? 1u : 0u;
            auto path_open = (oth.flags & kCtxDone) != 0u;// (else: an empty context, or one that is not `mine`)
            const auto traced_shadow = path_open && (oth.flags & kCtxHadShadow) != 0u, traced_closest = path_open && (oth.flags & kCtxHadClosest) != 0u;
            const auto occluded = (oth.flags & kCtxOccluded) != 0u;
            const auto ro = oth.no, rd = oth.nd;// the path segment the job traced
            const auto hit_tri = oth.tri;
            const auto hit_u = oth.u, hit_v = oth.v;
            // What the block reads of the record, and when (round 5).  Quads 0-2 -- NEE term, pdf, throughput, depth, radiance, sampler word 0
            // -- where the context's path is taken up.  Quad 3's pixel and work item never change along a path: written when the path starts,
            // read where it ends or is parked (LR_POOL_STATE_LEAN: +1.7 % on C2; 0 restores the read at every vertex).  The GENERIC sampler's
            // other three words (quads 3 and 4) are read where the vertex's random numbers are drawn and written back right there: outside
            // that stretch the sampler holds no register (sampler_take / sampler_leave below).
            constexpr bool LEAN_STATE = LR_POOL_STATE_LEAN != 0;
            auto word0 = 0u;// the sampler's first state word (Independent: its only one)
            const auto load_ids = [&]() {
                const auto q3 = state_load(side, 3u);
                pixel_index = __float_as_uint(q3.x), path_item = __float_as_uint(q3.y);
            };
            // generic sampler: the stream position comes into the registers ...
            const auto sampler_take = [&](uint32_t stage) {// stage: 0 the light's three numbers, 1 the closure's three or four
                if (PADDED) {
                    const auto q3 = state_load(side, 3u);
                    const auto depth = dp & 0xffffu;
                    const auto first_rr = scene.rr_depth == 0u ? 0u : scene.rr_depth - 1u;// the first depth that draws a roulette number
                    const auto dimension = (scene.camera.kind == LR_CAMERA_THIN_LENS ? 4u : 2u) + 6u * depth + (depth > first_rr ? depth - first_rr : 0u) + 3u * stage;
                    uint32_t words[kWfSamplerWordsMax] = {__float_as_uint(q3.z), 0u, dimension, __float_as_uint(q3.w)};
                    sampler.restore(scene, words);
                    return;
                }
                const auto q3 = state_load(side, 3u);
                uint32_t words[kWfSamplerWordsMax] = {word0, __float_as_uint(q3.z), __float_as_uint(q3.w), __float_as_uint(state_load(side, QUADS - 1u).x)};
                sampler.restore(scene, words);
            };
            // ... and leaves them again (`Li` is final where a vertex's numbers are drawn: quad 2 is complete)
            const auto sampler_leave = [&]() {
                if (PADDED) { return; }
                uint32_t words[kWfSamplerWordsMax] = {0u, 0u, 0u, 0u};
                sampler.save(words);
                word0 = words[0];
                state_store(side, 2u, make_float4(Li.x, Li.y, Li.z, __uint_as_float(words[0])));
                const auto q3 = state_load(side, 3u);
                state_store(side, 3u, make_float4(q3.x, q3.y, __uint_as_float(words[1]), __uint_as_float(words[2])));
                state_store(side, QUADS - 1u, make_float4(__uint_as_float(words[3]), 0.f, 0.f, 0.f));
            };
            auto sampler_left = false;// (PCG: quads 2-4 of this context are up to date already)
            if (path_open) {
                const auto q0 = state_load(side, 0u), q1 = state_load(side, 1u), q2 = state_load(side, 2u);
                nee = mk3(q0.x, q0.y, q0.z), pdf_bsdf = q0.w;
                beta = mk3(q1.x, q1.y, q1.z);
                const auto packed = __float_as_uint(q1.w);
                dp = packed & 0x3fffffu;
                Li = mk3(q2.x, q2.y, q2.z);
                word0 = __float_as_uint(q2.w);
                if (!LEAN_STATE) { load_ids(); }
                if (!PCG) {
                    uint32_t words[kWfSamplerWordsMax] = {word0, 0u, 0u, 0u};
                    sampler.restore(scene, words);
                }
            }
            auto want_shadow = false, want_closest = false;
            unsigned long long t_closure_sum = 0ull;// (COUNT: wave cycles inside the closure section of this batch; lanes agree)
#ifdef LR_STALL_PROBE
            uint32_t t_lobe = 0u, t_eval = 0u, t_sample = 0u, t_hit = 0u, t_light = 0u;// (sections of the block, lrhip_counters::probe[8..11] + [7] is the traversal's)
            const auto probe_clock = []() { return static_cast<uint32_t>(__builtin_readcyclecounter()); };
            const auto t_block = probe_clock();
#endif
            auto park_kind = kInvalid;// WF: closure kind (0 Disney, 1 Mix, 2 Layered) of the heavy surface this path just reached
            if (path_open) {
                if (traced_shadow) {// direct lighting of the bounce that spawned the shadow ray, mega_path.cpp:124-130
                    if (!occluded) { Li += nee; }
                }
                if (traced_closest) {// one iteration of the reference's depth loop, mega_path.cpp:63-154
                    if (COUNT) { local.shade_busy++; }
                    const auto wo = -rd;
                    const auto hit_valid = hit_tri != kInvalid;
                    // GENERIC SAMPLER (PCG32 / Sobol / PaddedSobol, dev_shade.h): every draw is a three-way dispatch around hashes, a permutation
                    // and table walks, and the sampler's four state words were alive through the whole block: <4098> wanted 158 VGPRs and
                    // spilled 33 on its hot path in round 4.  Now the sampler is in the registers for two short stretches only -- the light's
                    // three numbers, then the closure's three or four behind the evaluation (sampler_take / sampler_leave): 154 wanted, 17
                    // spilled, PaddedSobol C2 820 -> 875+ Msamples/s at 256 spp, films bit-identical (profiles/r05f).  MEASURED, NOT KEPT: all of
                    // a vertex's numbers drawn in one stretch where the first is used (20 spilled), or before the hit is even reconstructed,
                    // where next to nothing else is alive (162 wanted, 24 spilled: seven floats alive through the block cost more than the
                    // draws' temporaries at its peak).
                    auto u_light_selection = 0.f, u_lobe = 0.f, u_rr = 0.f;
                    f2 u_light_surface{0.f, 0.f}, u_bsdf{0.f, 0.f};
                    const auto rr = (dp & 0xffffu) + 1u >= scene.rr_depth;// Russian roulette, mega_path.cpp:148-153
                    if (!hit_valid && scene.env_kind != kEnvNone) {// miss, mega_path.cpp:70-76 -> evaluate_miss, uniform.cpp:67-76
                        f3 L = mk3(scene.env_L[0], scene.env_L[1], scene.env_L[2]);
                        auto pdf = kInvPi * 0.25f;
                        if (ENV && scene.env_kind != kEnvConstant) { env_evaluate(scene, rd, L, pdf); }
                        Li += beta * L * balance(pdf_bsdf, pdf * scene.env_prob);
                    }
                    SurfacePoint it;
                    auto has_surface = false;
#ifdef LR_STALL_PROBE
                    const auto t_h0 = probe_clock();
#endif
                    if (hit_valid) {
                        reconstruct_baked(scene, hit_tri, hit_u, hit_v, it);
                        it.back_facing = dot(wo, it.ng) < 0.0f;
                        if (COUNT) { local.surface_hits++; }
                        if (scene.has_lights && (it.flags & LR_SHAPE_HAS_LIGHT)) {// hit light, mega_path.cpp:79-86
                            f3 L;
                            float pdf;
                            const auto prim = reinterpret_cast<const uint32_t *>(scene.shade_tris + hit_tri)[29];
                            light_evaluate(scene, it, prim, ro, L, pdf);
                            pdf *= (1.f - scene.env_prob) / static_cast<float>(scene.light_count);
                            Li += beta * L * balance(pdf_bsdf, pdf);
                        }
                        has_surface = (it.flags & LR_SHAPE_HAS_SURFACE) != 0u;
                    }
                    // wavefront mode: a Disney / Mix / Layered surface is not shaded here (megapath_kernel.h): the path goes into the
                    // queue of its closure kind; heavy_kernel.h shades the vertex and hands the path back as a continuation record
                    if (WF && has_surface) {
                        const auto heavy_kind = (it.flags >> 10u) & 3u;// (baked into the triangle's record: lrhip.hip, build_shade_tris)
                        if (heavy_kind != 0u) { park_kind = heavy_kind - 1u, has_surface = false; }
                    }
                    if (WF) {// ---- park (at once: the hit and the direction die here instead of living through the closure code below -- 25 -> 15
                        // spilled VGPRs in the camera pass): one atomic per closure kind and wave, field-major stores (coalesced over the parking lanes)
                        if (lr_any(park_kind != kInvalid)) {
                            // (round 6: the queues' slots behind ONE atomic instruction -- the first lane of every kind reserves its kind's -- and one
                            // round trip, where a batch with hits of all three kinds made three in a row)
                            unsigned long long kind_mask[kWfKinds];
#pragma unroll
                            for (auto k = 0u; k < kWfKinds; k++) { kind_mask[k] = lr_ballot(park_kind == k); }
                            // (the first lane of each kind is its leader: lanes 0-2 need not be among the ones that got here)
                            static_assert(kWfKinds == 3u, "three closure kinds");
                            const auto my_mask = park_kind == 0u ? kind_mask[0] : (park_kind == 1u ? kind_mask[1] : (park_kind == 2u ? kind_mask[2] : 0ull));
                            auto reserved = 0u;
                            if (my_mask != 0ull && lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(my_mask))) - 1u) {
                                reserved = atomicAdd(scene.wf.counts + kWfCountHeavy + park_kind, static_cast<uint32_t>(__popcll(my_mask)));
                            }
#pragma unroll
                            for (auto k = 0u; k < kWfKinds; k++) {
                                const auto mask = kind_mask[k];
                                if (mask == 0ull) { continue; }
                                const auto leader = __ffsll(static_cast<long long>(mask)) - 1;
                                const auto out = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(reserved), leader)) + lane_rank(mask);
                                if (park_kind == k && out < scene.wf.capacity) {// (capacity >= the slice's paths: never full; a bound, not a policy)
                                    if (LEAN_STATE) { load_ids(); }
                                    if (PCG) { sampler_take(0u); }
                                    const auto q = wf_heavy_queue<SAMPLER_WORDS>(scene, k);
                                    q.put3(out, 0u, rd);
                                    q.put(out, 3u, hit_tri), q.put(out, 4u, hit_u), q.put(out, 5u, hit_v);
                                    q.put3(out, 6u, beta), q.put3(out, 9u, Li);
                                    q.put(out, 12u, pixel_index), q.put(out, 13u, dp & 0xffffu);
                                    uint32_t words[kWfSamplerWordsMax];
                                    sampler.save(words);
#pragma unroll
                                    for (auto w = 0u; w < SAMPLER_WORDS; w++) { q.put(out, kWfHeavyWords + w, words[w]); }
                                }
                            }
                        }
                    }
#ifdef LR_STALL_PROBE
                    asm volatile("" : "+v"(it.p.x), "+v"(it.ng.x), "+v"(Li.x));
                    const auto t_h1 = probe_clock();
                    t_hit = t_h1 - t_h0;
#endif
                    if (has_surface) {
                        if (COUNT) { local.path_length_sum++, local.nee_samples++; }
                        // random numbers are drawn where they are used, in the reference's order (mega_path.cpp:90-97):
                        // light selection, light surface (2), lobe, bsdf (2), [rr]
                        if constexpr (PADDED && LR_PADDED_DRAWS_OUT_OF_LINE != 0) {
                            sampler_take(0u);
                            const auto r = padded_draws(&scene, sampler.lo, sampler.w3, sampler.w2, false);
                            u_light_selection = r.a, u_light_surface = f2{r.b, r.c};
                        } else {
                            if (PCG) { sampler_take(0u); }
                            u_light_selection = sampler.next_1d();
                            u_light_surface = sampler.next_2d();
                            if (PCG) { sampler_leave(), sampler_left = !PADDED; }
                        }
                        // ---- sample one light, uniform.cpp:78-137 + light_sampler.cpp:57-63 (dev_shade.h: sample_one_light)
#if LR_LIGHT_OUT_OF_LINE
                        const auto pick = sample_one_light_call<ENV>(&scene, it.p, it.ng, it.shading.n, it.offset_bits, u_light_selection, u_light_surface.x, u_light_surface.y);
#else
                        const auto pick = sample_one_light<ENV>(scene, it, u_light_selection, u_light_surface);
#endif
                        shadow = pick.shadow;
#ifdef LR_STALL_PROBE
                        asm volatile("" : "+v"(shadow.o.x), "+v"(shadow.d.x));
#endif
                        const auto t_closure = COUNT ? __builtin_readcyclecounter() : 0ull;
#ifdef LR_STALL_PROBE
                        t_light = static_cast<uint32_t>(t_closure) - t_h1;
#endif
                        // ---- material, mega_path.cpp:111-143: the five basic closures (and Disney in a <Disney> variant) inline
                        const LobeTables tables{scene.closures, scene.surfaces, scene.textures, scene.texels};
                        DClosure closure;
                        Frame sh;
                        load_lobe(tables, it.uv, it.ng, wo, (it.tags >> 12u) & 4095u, it.shading, closure, sh);
#ifdef LR_STALL_PROBE
                        asm volatile("" : "+v"(closure.c0[0]), "+v"(closure.s0), "+v"(sh.n.x));
                        const auto t_l1 = probe_clock();
                        t_lobe = t_l1 - static_cast<uint32_t>(t_closure);
#endif
                        if (pick.pdf > 0.0f) {
                            const auto eval = closure_evaluate<DISNEY>(closure, sh, it.ng, wo, shadow.d);
                            const auto w = balance(pick.pdf, eval.pdf) / pick.pdf;
                            nee = w * beta * eval.f * pick.L;
                            // the reference traces the shadow ray unconditionally; a zero contribution cannot change Li
                            want_shadow = nee.x != 0.f || nee.y != 0.f || nee.z != 0.f;
                        }
#ifdef LR_STALL_PROBE
                        asm volatile("" : "+v"(nee.x), "+v"(nee.y), "+v"(nee.z));
                        const auto t_e1 = probe_clock();
                        t_eval = t_e1 - t_l1;
#endif
                        if constexpr (PADDED && LR_PADDED_DRAWS_OUT_OF_LINE != 0) {
                            sampler_take(1u);
                            const auto r = padded_draws(&scene, sampler.lo, sampler.w3, sampler.w2, rr);
                            u_lobe = r.a, u_bsdf = f2{r.b, r.c}, u_rr = r.d;
                        } else {
                            if (PCG) { sampler_take(1u); }
                            u_lobe = sampler.next_1d();
                            u_bsdf = sampler.next_2d();
                            if (PCG) {
                                if (rr) { u_rr = sampler.next_1d(); }
                                sampler_leave();
                            }
                        }
                        const auto bs = closure_sample<DISNEY>(closure, sh, it.ng, wo, u_lobe, u_bsdf);
                        auto eta = 1.f;
                        const auto has_eta = closure_eta(closure, eta);
                        ray.o = robust_origin(it, bs.wi);// spawn_ray, interaction.cpp:21-23
                        ray.d = bs.wi;
                        ray.t_min = 0.f, ray.t_max = kFloatMax;
                        pdf_bsdf = bs.pdf;
                        beta *= (bs.pdf > 0.f ? 1.f / bs.pdf : 0.f) * bs.f;
                        auto eta_scale = 1.f;
                        if (has_eta) {
                            if (bs.event == kEventEnter) { eta_scale = sqr(eta); }
                            else if (bs.event == kEventExit) { eta_scale = sqr(1.f / eta); }
                        }
                        if (any_nan(beta)) { beta = mk3(0.f); }// zero_if_any_nan
                        auto alive = !(beta.x <= 0.f && beta.y <= 0.f && beta.z <= 0.f);
                        if (!PCG && rr) { u_rr = sampler.next_1d(); }// (drawn before the closure in the reference: same stream position)
                        if (alive) {
                            const auto q = fmaxf(max_component(beta) * eta_scale, .05f);
                            if (rr) {
                                if (q < scene.rr_threshold && u_rr >= q) { alive = false; }
                                else { beta *= q < scene.rr_threshold ? 1.0f / q : 1.f; }
                            }
                        }
                        dp++;// (depth < 65536: lrhip_render sends deeper paths to the round 1-3 kernels)
                        want_closest = alive && (dp & 0xffffu) < scene.max_depth;
                        if (COUNT) { t_closure_sum += __builtin_readcyclecounter() - t_closure; }
#ifdef LR_STALL_PROBE
                        asm volatile("" : "+v"(beta.x), "+v"(ray.d.x));
                        t_sample = probe_clock() - t_e1;
#endif
                    }
                }
            }
            if (WF) {
                if (park_kind != kInvalid) { path_open = false; }// (it goes on elsewhere: nothing to accumulate here)
            }
            if (path_open && !want_shadow && !want_closest) {// path complete: film.accumulate (integrator.cpp:74)
                if (LEAN_STATE) { load_ids(); }
                const auto rgb = Li * scene.shutter_weight;
                if (CONT || path_item != item) {// (its wave has left the path's work item: the frame's sums directly)
                    wf_film_accumulate(scene, args.film, pixel_index, rgb, scene.film_clamp);
                } else if (!(any_nan(rgb) || any_inf(rgb))) {// ColorFilmInstance::_accumulate (color.cpp:107-130, effective_spp = 1) into the tile
                    const auto threshold = scene.film_clamp * fmaxf(1.f, 1.f);
                    const auto strength = fmaxf(fmaxf(fmaxf(fabsf(rgb.x), fabsf(rgb.y)), fabsf(rgb.z)), 0.f);
                    const auto c = rgb * (threshold / fmaxf(strength, threshold));
                    const auto px_sums = film_tile + (dp >> 16u) * 3u;
                    if (c.x != 0.f) { atomicAdd(px_sums + 0, radiance_to_fixed(c.x, scene.wf.accum_scale)); }
                    if (c.y != 0.f) { atomicAdd(px_sums + 1, radiance_to_fixed(c.y, scene.wf.accum_scale)); }
                    if (c.z != 0.f) { atomicAdd(px_sums + 2, radiance_to_fixed(c.z, scene.wf.accum_scale)); }
                } else {// rejected: the flush counts every sample of the item
                    atomicAdd(&args.film[pixel_index].w, -1.f);
                }
                path_open = false;
            }
            // ==== (A') path regeneration: contexts without a path take the next samples of the item's queue, in lane order
            const auto t_regen = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (COUNT) {// (the closure section is timed by the lanes that ran it: lane 0 reports the wave's figure)
                for (auto off = 32; off > 0; off >>= 1) { t_closure_sum = max(t_closure_sum, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(t_closure_sum), off))); }
            }
            // The loop below only HANDS OUT sample numbers (it is a loop because the wave may walk into its next work item half way through
            // the batch, and because a pixel beyond the frame's edge has no samples: such a lane asks again); the work of starting a path
            // follows it once.  (With the camera code inside the loop the register allocator took the whole shading block for a cold one.)
            auto need = mine && !path_open;// (a context without a path: it takes the next sample, if the launch has one left)
            auto got = false;
            auto new_k = 0u, new_item = kInvalid, new_px = 0u, new_py = 0u, new_s = 0u;
            for (;;) {
                const auto mask = lr_ballot(need);
                if (mask == 0ull) { break; }
                if (q_next >= q_total) {// the item's queue is dry: on to the next item, the old one's paths finish beside the new one's
                    if (!items_left) { break; }
                    take_item();
                    if (!items_left) { break; }
                    continue;
                }
                const auto avail = q_total - q_next;
                const auto rank = lane_rank(mask);
                if (need && rank < avail) {
                    const auto k = q_next + rank;
                    const auto px = tx * 8u + (k & 7u), py = ty * 8u + ((k >> 3u) & 7u);
                    if (CONT || (px < scene.camera.width && py < scene.camera.height)) {
                        new_k = k, new_item = item, new_px = px, new_py = py, new_s = s_begin + (k >> 6u);
                        got = true, need = false;
                    }
                }
                q_next += min(static_cast<uint32_t>(__popcll(mask)), avail);
            }
            if (got) {
                sampler_left = false;// (the context takes another path: every quad of its record is written below)
                if (CONT) {// a path comes back from the heavy kernel: as if this context's vertex had just been shaded
                    const auto rec = new_item * item_records + new_k;
                    const auto &q = cont_queue;
                    ray.o = q.get3(rec, 0u), ray.d = q.get3(rec, 3u);
                    ray.t_min = 0.f, ray.t_max = kFloatMax;
                    shadow.o = q.get3(rec, 6u), shadow.d = q.get3(rec, 9u);
                    shadow.t_min = 0.f, shadow.t_max = q.getf(rec, 12u);
                    nee = q.get3(rec, 13u), beta = q.get3(rec, 16u), Li = q.get3(rec, 19u);
                    pdf_bsdf = q.getf(rec, 22u);
                    pixel_index = q.get(rec, 23u);
                    const auto packed = q.get(rec, 24u);
                    dp = packed & 0xffffu;
                    want_shadow = (packed & (1u << 16u)) != 0u, want_closest = (packed & (1u << 17u)) != 0u;
                    uint32_t words[kWfSamplerWordsMax];
#pragma unroll
                    for (auto w = 0u; w < SAMPLER_WORDS; w++) { words[w] = q.get(rec, kWfContWords + w); }
                    sampler.restore(scene, words);
                    path_open = true;
                } else {// MegakernelPathTracingInstance::Li prologue, mega_path.cpp:52-62
                    pixel_index = new_py * scene.camera.width + new_px;
                    path_item = new_item;
                    // (MEASURED, NOT KEPT, round 6: this start of a path -- seed, filter tables, camera ray -- as one real call like the light sample: camera class
                    // 1219 against 1219, C2 -1.6 %, C3 -2.5 %, C5 -0.8 %, profiles/r06ze_camera_start_out_of_line.txt)
                    sampler.start(scene, new_px, new_py, new_s);
                    const auto u_filter = sampler.next_pixel_2d();
                    const auto u_lens = scene.camera.kind == LR_CAMERA_THIN_LENS ? sampler.next_2d() : f2{.5f, .5f};
                    float weight;
                    camera_ray(scene, scene.filter, new_px, new_py, u_filter, u_lens, ray, weight);
                    beta = mk3(weight);
                    Li = mk3(0.f), nee = mk3(0.f);
                    pdf_bsdf = 1e16f;
                    dp = (new_k & 63u) << 16u;
                    path_open = true, want_shadow = false, want_closest = true;
                    if (COUNT) { local.paths++; }
                }
            }
            // ---- every context that goes on: its path state back into the record, its job's rays into the context
            const auto t_launch = COUNT ? __builtin_readcyclecounter() : 0ull;
            const auto go = path_open && (want_shadow || want_closest);
            if (go) {
                state_store(side, 0u, make_float4(nee.x, nee.y, nee.z, pdf_bsdf));
                state_store(side, 1u, make_float4(beta.x, beta.y, beta.z, __uint_as_float(dp)));
                if (!PCG || !sampler_left) {// (generic sampler: a vertex's draws have left quads 2-4 up to date; a path that starts writes them here)
                    uint32_t words[kWfSamplerWordsMax] = {word0, 0u, 0u, 0u};
                    if (!PCG || got) { sampler.save(words); }
                    state_store(side, 2u, make_float4(Li.x, Li.y, Li.z, __uint_as_float(words[0])));
                    if (got && PADDED) {// (sample index and pixel: all the record keeps of the stream)
                        state_store(side, 3u, make_float4(__uint_as_float(pixel_index), __uint_as_float(path_item), __uint_as_float(words[0]), __uint_as_float(words[3])));
                    } else if (got) {
                        state_store(side, 3u, make_float4(__uint_as_float(pixel_index), __uint_as_float(path_item), __uint_as_float(words[1]), __uint_as_float(words[2])));
                        if (PCG) { state_store(side, QUADS - 1u, make_float4(__uint_as_float(words[3]), 0.f, 0.f, 0.f)); }
                    } else if (!LEAN_STATE && !PCG) {
                        state_store(side, 3u, make_float4(__uint_as_float(pixel_index), __uint_as_float(path_item), 0.f, 0.f));
                    }
                }
                if (COUNT) { local.closest_rays += want_closest ? 1u : 0u, local.shadow_rays += want_shadow ? 1u : 0u; }
            }
            // ---- the contexts come back: the current one from the LDS, and with it origin and direction of the ray in flight; the other
            // one from shadow / ray (the shaded context's new job, or -- not `mine` -- the rays it waits with)
            {
                const auto stage = stack.stage;
                asm volatile("" ::: "memory");
                const auto p0 = stage[lane], p1 = stage[64u + lane], p2 = stage[128u + lane], p3 = stage[192u + lane];
                cur.so = mk3(p0.x, p0.y, p0.z), cur.s_tmax = p0.w;
                cur.sd = mk3(p1.x, p1.y, p1.z), cur.n_tmin = p1.w;
                cur.no = mk3(p2.x, p2.y, p2.z), cur.n_tmax = p2.w;
                cur.nd = mk3(p3.x, p3.y, p3.z);
                const auto keep_word = __float_as_uint(p3.w);
                tr.phase = keep_word & 7u, tr.occluded = (keep_word & 8u) != 0u;
                tr.sp = (keep_word >> 5u) & 127u;
#if LR_POOL_PARK_ON_STACK
                const auto keep_tri = stack.pop(tr.sp + 0u);
                const auto keep_u = __uint_as_float(stack.pop(tr.sp + 1u)), keep_v = __uint_as_float(stack.pop(tr.sp + 2u));
                tr.t_max = __uint_as_float(stack.pop(tr.sp + 3u)), tr.cur = stack.pop(tr.sp + 4u);
#else
                const auto park = s_park + tid;
                const auto keep_tri = park[0u * kBlockThreads];
                const auto keep_u = __uint_as_float(park[1u * kBlockThreads]), keep_v = __uint_as_float(park[2u * kBlockThreads]);
                tr.t_max = __uint_as_float(park[3u * kBlockThreads]), tr.cur = park[4u * kBlockThreads];
#endif
                cur.flags = (keep_word >> 12u) & 0xffu, cur.tri = keep_tri, cur.u = keep_u, cur.v = keep_v;
                tr.hit.tri = keep_tri, tr.hit.u = keep_u, tr.hit.v = keep_v;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                oth.so = shadow.o, oth.sd = shadow.d, oth.s_tmax = shadow.t_max;
                oth.no = ray.o, oth.nd = ray.d, oth.n_tmin = ray.t_min, oth.n_tmax = ray.t_max;
                oth.tri = kInvalid, oth.u = 0.f, oth.v = 0.f;// (a context that waits with its rays has no hit yet)
                // the shaded context: its new job, or empty (its path ended, or was parked, and no sample was left for it)
                const auto shaded_flags = (side != 0u ? kCtxSide : 0u) |
                                          (go ? kCtxOpen | (want_shadow ? kCtxShadow | kCtxHadShadow : 0u) | (want_closest ? kCtxClosest | kCtxHadClosest : 0u) : 0u);
                oth.flags = mine ? shaded_flags : (keep_word >> 20u) & 0xffu;
                if ((tr.phase & 3u) == kPhaseShadow) { tr.o = cur.so, tr.d = cur.sd, tr.t_min = 0.f; }
                else { tr.o = cur.no, tr.d = cur.nd, tr.t_min = cur.n_tmin; }
            }
            // ---- an idle lane starts on the job its context was just given
            if (tr.phase == kPhaseIdle && (oth.flags & kCtxRays) != 0u) {
                ctx_swap(cur, oth);
                ctx_start(cur, tr);
            }
#ifdef LR_STALL_PROBE
            if (COUNT) {// the block's sections as the slowest lane saw them (lanes that ran a section agree to within its divergence)
                const auto wave_max = [](uint32_t v) {
                    for (auto off = 32; off > 0; off >>= 1) { v = max(v, static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), off))); }
                    return v;
                };
                const auto m_hit = wave_max(t_hit), m_light = wave_max(t_light), m_lobe = wave_max(t_lobe), m_eval = wave_max(t_eval), m_sample = wave_max(t_sample);
                if (probe_wave && lane == 0u) {
                    atomicAdd(&args.counters->probe[14], 1ull);// (batches of the reporting waves)
                    atomicAdd(&args.counters->probe[15], static_cast<unsigned long long>(probe_clock() - t_block));// (... and their cycles in the block)
                    atomicAdd(&args.counters->probe[8], static_cast<unsigned long long>(m_hit + m_light));
                    atomicAdd(&args.counters->probe[9], static_cast<unsigned long long>(m_lobe));
                    atomicAdd(&args.counters->probe[10], static_cast<unsigned long long>(m_eval));
                    atomicAdd(&args.counters->probe[11], static_cast<unsigned long long>(m_sample));
                }
            }
#endif
            if (COUNT) {
                if (lane == 0u) {
                    const auto t_end = __builtin_readcyclecounter();
                    local.shade_cycles += t_end - t_shade;
                    local.shade_closure_cycles += t_closure_sum, local.shade_regen_cycles += t_launch - t_regen;
                    local.shade_light_cycles += (t_regen - t_shade) - t_closure_sum;// everything of (A) that is not the closure section
                }
                local.shade_calls++;
            }
        }
        if (!lr_any(tr.phase != kPhaseIdle)) { break; }// nothing in flight and nothing to shade: every context of the wave is out of samples
        // ==== (B) traverse: lanes switch to their other context's job inside the loop
        TraceStats ts{};
#ifdef LR_STALL_PROBE
        ts.probe_lds = probe_wave ? static_cast<uint32_t>(reinterpret_cast<uintptr_t>((TraversalStack::lds_u32 *)(s_probe + wave_in_block * kProbeWords))) : 0u;
#endif
        const auto t_trace = COUNT ? __builtin_readcyclecounter() : 0ull;
        for (;;) {
            const auto for_alpha = pool_trace<COUNT, ALPHA>(scene, stack, tr, cur, oth, items_left || q_next < q_total, ts);
            if (!ALPHA) { break; }
            // (candidates may wait when the wave leaves for the shading block, too: no lane takes one into it)
            if (lr_any((tr.phase & kPhasePendingAlpha) != 0u)) { resolve_pending_alpha(scene, stack, tr); }
            if (!for_alpha) { break; }
        }
#ifdef LR_STALL_PROBE
        if (COUNT && ts.probe_lds != 0u && lane == 0u) {// section cycles of this traversal call (the wave's LDS words) -> lrhip_counters::probe
            const auto words = s_probe + wave_in_block * kProbeWords;
#pragma unroll
            for (auto i = 0u; i < kProbeSlots; i++) { atomicAdd(&args.counters->probe[i], static_cast<unsigned long long>(words[i])), words[i] = 0u; }
            atomicAdd(&args.counters->probe[12], static_cast<unsigned long long>(ts.steps));// the sampled waves' iterations ...
            atomicAdd(&args.counters->probe[13], static_cast<unsigned long long>(__builtin_readcyclecounter() - t_trace));// ... and their cycles in the loop
        }
#endif
        if (COUNT) {
            if (lane == 0u) { local.trace_cycles += __builtin_readcyclecounter() - t_trace; }
            local.nodes_visited += ts.nodes, local.tris_tested += ts.tris, local.nodes_empty += ts.nodes_empty;
            local.trace_steps += ts.steps, local.trace_steps_busy += ts.steps_busy, local.trace_steps_starved += ts.steps_starved;
        }
    }
    flush_tile();// the last item's tile

    if (COUNT) {// one atomic per counter per wave
        if (lane == 0u) { local.wave_cycles = __builtin_readcyclecounter() - t_wave; }
        auto reduce = [&](unsigned long long v, unsigned long long *dst) {
            for (auto off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off); }
            if (lane == 0u) { atomicAdd(dst, v); }
        };
        reduce(local.paths, &args.counters->paths);
        reduce(local.closest_rays, &args.counters->closest_rays);
        reduce(local.shadow_rays, &args.counters->shadow_rays);
        reduce(local.nodes_visited, &args.counters->nodes_visited);
        reduce(local.tris_tested, &args.counters->tris_tested);
        reduce(local.surface_hits, &args.counters->surface_hits);
        reduce(local.nee_samples, &args.counters->nee_samples);
        reduce(local.path_length_sum, &args.counters->path_length_sum);
        reduce(local.trace_steps, &args.counters->trace_steps);
        reduce(local.trace_steps_busy, &args.counters->trace_steps_busy);
        reduce(local.shade_calls, &args.counters->shade_calls);
        reduce(local.shade_busy, &args.counters->shade_busy);
        reduce(local.trace_steps_starved, &args.counters->trace_steps_starved);
        reduce(local.shade_cycles, &args.counters->shade_cycles);
        reduce(local.trace_cycles, &args.counters->trace_cycles);
        reduce(local.wave_cycles, &args.counters->wave_cycles);
        reduce(local.nodes_empty, &args.counters->nodes_empty);
        reduce(local.shade_light_cycles, &args.counters->shade_light_cycles);
        reduce(local.shade_closure_cycles, &args.counters->shade_closure_cycles);
        reduce(local.shade_regen_cycles, &args.counters->shade_regen_cycles);
    }
}

}// namespace lrd
