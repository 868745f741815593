// megapool_kernel.h — round 4: the persistent-threads megakernel with a PATH POOL per wavefront.
//
// megapath_kernel.h (rounds 1-3) binds one path to one lane: a lane whose ray has finished waits, idle, until enough of its
// neighbours have finished too (the refill threshold), and the shading block then runs for the ~half of the wave that has
// something to shade.  Measured on the C2 stand-in (profiles/r03am_*): 56 % of the traversal loop's lane-steps and 47 % of the
// shading block's lanes did useful work while the VALU pipes were ~full -- the machine was busy computing masked-off lanes.
//
// Here a wave owns kPoolSlots = 128 PATH SLOTS -- twice its lanes -- whose state lives in a wave-private 16 KB record array in
// global memory (one 128-byte line per slot: rays to trace, throughput, radiance, sampler position, hit), i.e. in L2 / Infinity
// Cache, and lanes are workers:
//   * a slot with rays to trace (its shadow ray and / or its next path segment: one JOB) waits in the wave's RAY QUEUE, a slot whose
//     job is done waits in the SHADE QUEUE; both are FIFOs of slot numbers in LDS, filled and drained with ballot + prefix counts
//     (wave-private: no atomics);
//   * in the traversal loop a lane that finishes its job RETIRES it (hit -> slot, slot -> shade queue) and takes the next job off
//     the ray queue without leaving the loop, in batches of LR_POOL_REFILL lanes; the loop is left when the ray queue is dry and
//     lanes begin to idle -- by then at least 64 slots wait in the shade queue (128 slots - at most 64 in flight);
//   * the shading block takes 64 slots off the shade queue -- a FULL wave, whatever the lanes' own rays are doing (a lane keeps its
//     ray in flight in its registers while it shades another slot's vertex) -- and every slot leaves it with a new job: the path's
//     next rays, or the first ray of the next sample of the work item (path regeneration, as before).
// The scheduling model (tools/sched_model.py, calibrated on the round-3 counters): lane utilisation of the traversal loop 0.70 -> 0.95,
// of the shading block 0.63 -> 1.0, cost per job 0.89 -> 0.65.
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
// megapath_kernel.h -- the shading block below is that file's, reading a slot instead of the lane's registers -- and the wavefront-mode
// roles of that kernel (kFeatWf camera pass: heavy hits parked for heavy_kernel.h; kFeatCont: continuation records instead of
// camera samples) carry over unchanged.
#pragma once
#include "dev_wavefront.h"

namespace lrd {

#ifndef LR_MIN_WAVES
#define LR_MIN_WAVES 4
#endif
#ifndef LR_POOL_SLOTS
#define LR_POOL_SLOTS 128
#endif
#ifndef LR_POOL_REFILL
#define LR_POOL_REFILL 8   // idle lanes that trigger a retire + fetch round inside the traversal loop
#endif
#ifndef LR_POOL_MIN_READY
#define LR_POOL_MIN_READY 16// (tail of a launch, pool no longer full) slots that must wait for shading before the traversal loop is left for them
#endif
constexpr uint32_t kPoolSlots = LR_POOL_SLOTS;
static_assert((kPoolSlots & (kPoolSlots - 1u)) == 0u && kPoolSlots >= 64u && kPoolSlots <= 256u, "slots per wave: a power of two, one byte");

// ---- slot record: kPoolQuads x float4 (lean sampler: one 128-byte line)
//   0  shadow o.xyz | shadow t_max          1  shadow d.xyz | pixel index (frame)
//   2  next ray o.xyz | t_max               3  next ray d.xyz | t_min
//   4  nee.xyz | pdf_bsdf                   5  beta.xyz | depth (16) | pixel in tile (6) << 16 | job had shadow << 22 | closest << 23
//   6  Li.xyz | sampler word 0              7  hit: tri | occluded << 31 (miss: tri = 0x7fffffff), u, v | work item of the path
//   8  sampler words 1-3 (generic sampler only)
// Quads 0-3 are what a lane reads when it takes the job (both rays in ONE format), 7.xyz what it writes when it retires it.
constexpr uint32_t kPoolMiss = 0x7fffffffu;
template<bool GENERIC>
constexpr uint32_t pool_quads() { return GENERIC ? 9u : 8u; }
// job word (ray queue entry, lane register): slot | rays still to trace
enum : uint32_t { kJobSlotMask = 0xffu, kJobShadow = 1u << 8u, kJobClosest = 1u << 9u, kNoJob = 0xffffffffu };

struct PoolWave {// the wave's queues (wave-uniform: SGPRs); heads run free, entries live at (index & (kPoolSlots - 1))
    uint32_t rq_head, rq_count;// ray queue: slots with a job
    uint32_t sq_head, sq_count;// shade queue: slots whose job is done
};
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

LR_D uint32_t lane_rank(unsigned long long mask) {// lanes of `mask` below this one
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32u), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Retire + fetch round of the traversal loop.  Every idle lane whose job is complete writes the hit into its slot and queues the slot
// for shading; with FETCH every idle lane without a job then takes the next one off the ray queue and starts its first ray.
template<uint32_t QUADS, bool FETCH>
LR_D void pool_refill(TravState &tr, uint32_t &job, Ray &next, f3 &inv, float4 *slots, lds_u16 *rq, lds_u8 *sq, PoolWave &pw) {
    const auto idle = tr.phase == kPhaseIdle;
    const auto retire = idle && job != kNoJob;// (a lane goes idle only when no ray of its job is left, see pool_trace)
    const auto rmask = __ballot(retire);
    if (rmask != 0ull) {
        if (retire) {
            const auto slot = job & kJobSlotMask;
            auto out = reinterpret_cast<uint32_t *>(slots + slot * QUADS + 7u);
            out[0] = (tr.hit.tri & kPoolMiss) | (tr.occluded ? 0x80000000u : 0u);// (kInvalid & kPoolMiss = kPoolMiss)
            out[1] = __float_as_uint(tr.hit.u), out[2] = __float_as_uint(tr.hit.v);
            sq[(pw.sq_head + pw.sq_count + lane_rank(rmask)) & (kPoolSlots - 1u)] = static_cast<uint8_t>(slot);
            job = kNoJob;
        }
        pw.sq_count += static_cast<uint32_t>(__popcll(rmask));
    }
    if (FETCH && pw.rq_count != 0u) {
        const auto wmask = __ballot(idle);// (every idle lane is without a job now)
        const auto n = min(static_cast<uint32_t>(__popcll(wmask)), pw.rq_count);
        if (idle && lane_rank(wmask) < n) {
            job = rq[(pw.rq_head + lane_rank(wmask)) & (kPoolSlots - 1u)];
            const auto s = slots + (job & kJobSlotMask) * QUADS;
            const auto shadow_first = (job & kJobShadow) != 0u;
            const auto first = s + (shadow_first ? 0u : 2u);
            const auto qa = first[0], qb = first[1];
            tr.o = mk3(qa.x, qa.y, qa.z), tr.t_max = qa.w;
            tr.d = mk3(qb.x, qb.y, qb.z), tr.t_min = shadow_first ? 0.f : qb.w;
            if (shadow_first && (job & kJobClosest) != 0u) {// the path's next segment follows the shadow ray inside the loop
                const auto qc = s[2], qd = s[3];
                next.o = mk3(qc.x, qc.y, qc.z), next.t_max = qc.w;
                next.d = mk3(qd.x, qd.y, qd.z), next.t_min = qd.w;
            }
            tr.cur = 0u, tr.sp = 0u;// root
            tr.phase = shadow_first ? kPhaseShadow : kPhaseClosest;
            job &= shadow_first ? ~kJobShadow : ~kJobClosest;
            tr.hit.tri = kInvalid, tr.hit.u = 0.f, tr.hit.v = 0.f;
            tr.occluded = false;
            inv = safe_inverse(tr.d);
        }
        pw.rq_head += n, pw.rq_count -= n;
    }
}

// The traversal loop of the pool kernel: dev_trace.h's node and leaf steps, with job turnover inside the loop.  Returns when the ray
// queue is dry and LR_POOL_REFILL lanes have nothing to do while enough slots wait for shading, or when nothing is left to trace
// (ALPHA: also when a lane holds a candidate hit for the alpha test, dev_shade.h: resolve_pending_alpha).  Must be called by all 64 lanes.
template<bool COUNT, bool ALPHA, uint32_t QUADS>
LR_D void pool_trace(const DScene &scene, const TraversalStack &stack, TravState &tr, uint32_t &job, Ray &next, float4 *slots, lds_u16 *rq,
                     lds_u8 *sq, PoolWave &pw, TraceStats &stats) {
    const auto tl = TravLane::make(scene, stack);
    auto inv = safe_inverse(tr.d);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");// the slots and queue entries the shading block wrote
    for (;;) {
        {// ---- job turnover, LR_POOL_REFILL lanes at a time
            const auto idle = tr.phase == kPhaseIdle;
            if (static_cast<uint32_t>(__popcll(__ballot(idle))) >= static_cast<uint32_t>(LR_POOL_REFILL) && (pw.rq_count != 0u || __any(idle && job != kNoJob))) {
                pool_refill<QUADS, true>(tr, job, next, inv, slots, rq, sq, pw);
            }
        }
        if (COUNT) {
            stats.steps++, stats.steps_busy += tr.phase != kPhaseIdle ? 1u : 0u;
            stats.steps_starved += tr.phase == kPhaseIdle && pw.rq_count == 0u ? 1u : 0u;// idle with nothing to fetch
        }
        const auto live = ALPHA ? (tr.phase == kPhaseShadow || tr.phase == kPhaseClosest) : tr.phase != kPhaseIdle;// (not parked)
        const auto is_inner = live && tr.cur != kInvalid && !(tr.cur & kLeafFlag);
        const auto deep = __any(live && tr.sp + 3u > kStackLds);
        if (__any(is_inner)) { trav_node_step<COUNT>(stack, tl, tr, inv, is_inner, deep, stats); }
        if (live && tr.cur != kInvalid && (tr.cur & kLeafFlag) != 0u) { trav_leaf_step<COUNT, ALPHA>(stack, tl, tr, deep, stats); }
        // ---- ray finished: the job's next ray, or idle (retired at the next turnover)
        if (live && tr.cur == kInvalid) {
            if (tr.phase == kPhaseShadow && (job & kJobClosest) != 0u) {
                trav_begin(tr, next, kPhaseClosest);
                job &= ~kJobClosest;
                inv = safe_inverse(tr.d);
            } else {
                tr.phase = kPhaseIdle;
            }
        }
        if (ALPHA && __any((tr.phase & kPhasePendingAlpha) != 0u)) { break; }
        if (pw.rq_count == 0u) {
            const auto idle_mask = __ballot(tr.phase == kPhaseIdle);
            if (idle_mask == ~0ull) { break; }// nothing in flight, nothing to fetch
            const auto ready = pw.sq_count + static_cast<uint32_t>(__popcll(__ballot(tr.phase == kPhaseIdle && job != kNoJob)));
            if (static_cast<uint32_t>(__popcll(idle_mask)) >= static_cast<uint32_t>(LR_POOL_REFILL) && ready >= static_cast<uint32_t>(LR_POOL_MIN_READY)) { break; }
        }
    }
    pool_refill<QUADS, false>(tr, job, next, inv, slots, rq, sq, pw);// retire what has finished
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

template<uint32_t F>
__global__ __launch_bounds__(kBlockThreads, LR_MIN_WAVES) void megapool_kernel(DScenePtr scene_ptr, RenderArgs args) {
    const DScene &scene = *(const DScene *)scene_ptr;
    constexpr bool COUNT = (F & kFeatCount) != 0u, PCG = (F & kFeatGeneric) != 0u, ENV = (F & kFeatEnv) != 0u,
                   ALPHA = (F & kFeatAlpha) != 0u, DISNEY = (F & kFeatDisney) != 0u, WF = (F & kFeatWf) != 0u, CONT = (F & kFeatCont) != 0u;
    static_assert((F & kFeatPool) != 0u, "a pool variant");
    static_assert((F & (kFeatMix | kFeatLayered | kFeatAux | kFeatVpt | kFeatNest)) == 0u, "the pool scheduler exists for the lean kernels (closures inline)");
    static_assert(!WF || !DISNEY, "a wavefront variant is a lean kernel: the heavy closures live in heavy_kernel.h");
    static_assert(!CONT || WF, "the continuation pass exists in wavefront mode only");
    constexpr uint32_t SAMPLER_WORDS = PathSampler<PCG>::kSavedWords;
    constexpr uint32_t QUADS = pool_quads<PCG>();
    __shared__ uint32_t s_stack[kStackLds * kBlockThreads];
    __shared__ float4 s_stage[kWavesPerBlock * kStageWave];// 4 KiB of node packets per wave
    __shared__ unsigned long long s_film[CONT ? 1u : kWavesPerBlock * 192u];// per-wave tile accumulators, fixed point [pixel][rgb]
    __shared__ uint16_t s_rq[kWavesPerBlock * kPoolSlots];
    __shared__ uint8_t s_sq[kWavesPerBlock * kPoolSlots];
    const auto tid = threadIdx.x;
    const auto lane = tid & 63u;
    const auto gtid = blockIdx.x * kBlockThreads + tid;
    const auto wave_in_block = __builtin_amdgcn_readfirstlane(tid >> 6u);
    TraversalStack stack{s_stack + tid, args.spill + gtid, args.total_threads, s_stage + wave_in_block * kStageWave};
    const auto film_tile = s_film + (CONT ? 0u : wave_in_block * 192u);
    const auto rq = (lds_u16 *)(s_rq + wave_in_block * kPoolSlots);
    const auto sq = (lds_u8 *)(s_sq + wave_in_block * kPoolSlots);
    const auto slots = args.pool + static_cast<size_t>(blockIdx.x * kWavesPerBlock + wave_in_block) * (kPoolSlots * QUADS);
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
    PoolWave pw{0u, 0u, 0u, 0u};
    auto next_fresh = 0u;          // slots [next_fresh, kPoolSlots) have never held a path
    if (!CONT) { film_tile[lane * 3u] = 0ull, film_tile[lane * 3u + 1u] = 0ull, film_tile[lane * 3u + 2u] = 0ull; }
    // ---- the lane as a traversal worker: its ray in flight, the job (slot) it belongs to
    TravState tr{};
    tr.phase = kPhaseIdle;
    uint32_t job = kNoJob;
    Ray next{};// the job's path segment waiting behind its shadow ray

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

    auto after_trace = true;
    for (;;) {
        // ==== (A) shading batches: while a full wave of slots waits (or, after the traversal loop gave up, whatever waits)
        for (;;) {
            const auto fresh = (items_left || q_next < q_total) ? kPoolSlots - next_fresh : 0u;
            const auto shadeable = pw.sq_count + fresh;
            if (!(shadeable >= 64u || (shadeable != 0u && after_trace))) { break; }
            after_trace = false;
            const auto t_shade = COUNT ? __builtin_readcyclecounter() : 0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");// what the traversal loop retired
            // ---- up to 64 slots off the shade queue, one per lane
            const auto n_sq = min(pw.sq_count, 64u);
            auto slot = kInvalid;
            if (lane < n_sq) { slot = sq[(pw.sq_head + lane) & (kPoolSlots - 1u)]; }
            pw.sq_head += n_sq, pw.sq_count -= n_sq;
            // ---- the slot's path
            PathSampler<PCG> sampler{};
            Ray ray{}, shadow{};
            f3 beta = mk3(0.f), Li = mk3(0.f), nee = mk3(0.f);
            auto pdf_bsdf = 1e16f;
            auto depth = 0u, pixel_index = 0u, pix = 0u, path_item = kInvalid;
            auto path_open = false, traced_shadow = false, traced_closest = false, occluded = false;
            f3 ro = mk3(0.f), rd = mk3(0.f);// the path segment the job traced
            auto hit_tri = kPoolMiss;
            auto hit_u = 0.f, hit_v = 0.f;
            if (slot != kInvalid) {
                const auto s = slots + slot * QUADS;
                const auto q2 = s[2], q3 = s[3], q4 = s[4], q5 = s[5], q6 = s[6], q7 = s[7];
                pixel_index = reinterpret_cast<const uint32_t *>(s + 1u)[3];
                ro = mk3(q2.x, q2.y, q2.z), rd = mk3(q3.x, q3.y, q3.z);
                nee = mk3(q4.x, q4.y, q4.z), pdf_bsdf = q4.w;
                beta = mk3(q5.x, q5.y, q5.z);
                const auto packed = __float_as_uint(q5.w);
                depth = packed & 0xffffu, pix = (packed >> 16u) & 63u;
                traced_shadow = (packed & (1u << 22u)) != 0u, traced_closest = (packed & (1u << 23u)) != 0u;
                Li = mk3(q6.x, q6.y, q6.z);
                uint32_t words[kWfSamplerWordsMax];
                words[0] = __float_as_uint(q6.w);
                if (PCG) {
                    const auto q8 = s[QUADS - 1u];
                    words[1] = __float_as_uint(q8.x), words[2] = __float_as_uint(q8.y), words[3] = __float_as_uint(q8.z);
                }
                sampler.restore(scene, words);
                const auto h = __float_as_uint(q7.x);
                hit_tri = h & kPoolMiss, occluded = (h >> 31u) != 0u;
                hit_u = q7.y, hit_v = q7.z;
                path_item = __float_as_uint(q7.w);
                path_open = true;
            }
            auto want_shadow = false, want_closest = false;
            unsigned long long t_closure_sum = 0ull;// (COUNT: wave cycles inside the closure section of this batch; lanes agree)
            auto park_kind = kInvalid;// WF: closure kind (0 Disney, 1 Mix, 2 Layered) of the heavy surface this path just reached
            if (path_open) {
                if (traced_shadow) {// direct lighting of the bounce that spawned the shadow ray, mega_path.cpp:124-130
                    if (!occluded) { Li += nee; }
                }
                if (traced_closest) {// one iteration of the reference's depth loop, mega_path.cpp:63-154
                    if (COUNT) { local.shade_busy++; }
                    const auto wo = -rd;
                    const auto hit_valid = hit_tri != kPoolMiss;
                    if (!hit_valid && scene.env_kind != kEnvNone) {// miss, mega_path.cpp:70-76 -> evaluate_miss, uniform.cpp:67-76
                        f3 L = mk3(scene.env_L[0], scene.env_L[1], scene.env_L[2]);
                        auto pdf = kInvPi * 0.25f;
                        if (ENV && scene.env_kind != kEnvConstant) { env_evaluate(scene, rd, L, pdf); }
                        Li += beta * L * balance(pdf_bsdf, pdf * scene.env_prob);
                    }
                    SurfacePoint it;
                    auto has_surface = false;
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
                        const auto kind = scene.closures[(it.tags >> 12u) & 4095u].kind;
                        if (kind >= LR_SURFACE_DISNEY) { park_kind = kind - LR_SURFACE_DISNEY, has_surface = false; }
                    }
                    if (has_surface) {
                        if (COUNT) { local.path_length_sum++, local.nee_samples++; }
                        // random numbers are drawn where they are used, in the reference's order (mega_path.cpp:90-97):
                        // light selection, light surface (2), lobe, bsdf (2), [rr]
                        const auto u_light_selection = sampler.next_1d();
                        const auto u_light_surface = sampler.next_2d();
                        // ---- sample one light, uniform.cpp:78-137 + light_sampler.cpp:57-63 (dev_shade.h: sample_one_light)
                        const auto pick = sample_one_light<ENV>(scene, it, u_light_selection, u_light_surface);
                        shadow = pick.shadow;
                        const auto t_closure = COUNT ? __builtin_readcyclecounter() : 0ull;
                        // ---- material, mega_path.cpp:111-143: the five basic closures (and Disney in a <Disney> variant) inline
                        const LobeTables tables{scene.closures, scene.surfaces, scene.textures, scene.texels};
                        DClosure closure;
                        Frame sh;
                        load_lobe(tables, it.uv, it.ng, wo, (it.tags >> 12u) & 4095u, it.shading, closure, sh);
                        if (pick.pdf > 0.0f) {
                            const auto eval = closure_evaluate<DISNEY>(closure, sh, it.ng, wo, shadow.d);
                            const auto w = balance(pick.pdf, eval.pdf) / pick.pdf;
                            nee = w * beta * eval.f * pick.L;
                            // the reference traces the shadow ray unconditionally; a zero contribution cannot change Li
                            want_shadow = nee.x != 0.f || nee.y != 0.f || nee.z != 0.f;
                        }
                        const auto u_lobe = sampler.next_1d();
                        const auto u_bsdf = sampler.next_2d();
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
                        const auto rr = depth + 1u >= scene.rr_depth;// Russian roulette, mega_path.cpp:148-153
                        auto u_rr = 0.f;
                        if (rr) { u_rr = sampler.next_1d(); }// (drawn before the closure in the reference: same stream position)
                        if (alive) {
                            const auto q = fmaxf(max_component(beta) * eta_scale, .05f);
                            if (rr) {
                                if (q < scene.rr_threshold && u_rr >= q) { alive = false; }
                                else { beta *= q < scene.rr_threshold ? 1.0f / q : 1.f; }
                            }
                        }
                        depth++;
                        want_closest = alive && depth < scene.max_depth;
                        if (COUNT) { t_closure_sum += __builtin_readcyclecounter() - t_closure; }
                    }
                }
            }
            if (WF) {// ---- park: one atomic per closure kind and wave, field-major stores (coalesced over the parking lanes)
                if (__any(park_kind != kInvalid)) {
#pragma unroll
                    for (auto k = 0u; k < kWfKinds; k++) {
                        const auto mask = __ballot(park_kind == k);
                        if (mask == 0ull) { continue; }
                        const auto out = wf_reserve(scene.wf.counts + kWfCountHeavy + k, mask, lane);
                        if (park_kind == k && out < scene.wf.capacity) {// (capacity >= the slice's paths: never full; a bound, not a policy)
                            const auto q = wf_heavy_queue<SAMPLER_WORDS>(scene, k);
                            q.put3(out, 0u, rd);
                            q.put(out, 3u, hit_tri), q.put(out, 4u, hit_u), q.put(out, 5u, hit_v);
                            q.put3(out, 6u, beta), q.put3(out, 9u, Li);
                            q.put(out, 12u, pixel_index), q.put(out, 13u, depth);
                            uint32_t words[kWfSamplerWordsMax];
                            sampler.save(words);
#pragma unroll
                            for (auto w = 0u; w < SAMPLER_WORDS; w++) { q.put(out, kWfHeavyWords + w, words[w]); }
                        }
                    }
                }
                if (park_kind != kInvalid) { path_open = false; }// (it goes on elsewhere: nothing to accumulate here)
            }
            if (path_open && !want_shadow && !want_closest) {// path complete: film.accumulate (integrator.cpp:74)
                const auto rgb = Li * scene.shutter_weight;
                if (CONT || path_item != item) {// (its wave has left the path's work item: the frame's sums directly)
                    wf_film_accumulate(scene, args.film, pixel_index, rgb, scene.film_clamp);
                } else if (!(any_nan(rgb) || any_inf(rgb))) {// ColorFilmInstance::_accumulate (color.cpp:107-130, effective_spp = 1) into the tile
                    const auto threshold = scene.film_clamp * fmaxf(1.f, 1.f);
                    const auto strength = fmaxf(fmaxf(fmaxf(fabsf(rgb.x), fabsf(rgb.y)), fabsf(rgb.z)), 0.f);
                    const auto c = rgb * (threshold / fmaxf(strength, threshold));
                    const auto px_sums = film_tile + pix * 3u;
                    if (c.x != 0.f) { atomicAdd(px_sums + 0, radiance_to_fixed(c.x, scene.wf.accum_scale)); }
                    if (c.y != 0.f) { atomicAdd(px_sums + 1, radiance_to_fixed(c.y, scene.wf.accum_scale)); }
                    if (c.z != 0.f) { atomicAdd(px_sums + 2, radiance_to_fixed(c.z, scene.wf.accum_scale)); }
                } else {// rejected: the flush counts every sample of the item
                    atomicAdd(&args.film[pixel_index].w, -1.f);
                }
                path_open = false;
            }
            // ==== (A') path regeneration: slots without a path take the next samples of the item's queue, in lane order; lanes that
            // got no slot off the shade queue open the pool's unused slots (start of the launch)
            const auto t_regen = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (COUNT) {// (the closure section is timed by the lanes that ran it: lane 0 reports the wave's figure)
                for (auto off = 32; off > 0; off >>= 1) { t_closure_sum = max(t_closure_sum, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(t_closure_sum), off))); }
            }
            if (next_fresh < kPoolSlots && (items_left || q_next < q_total)) {
                const auto mask = __ballot(slot == kInvalid);
                if (mask != 0ull) {
                    const auto s = next_fresh + lane_rank(mask);
                    if (slot == kInvalid && s < kPoolSlots) { slot = s; }
                    next_fresh = min(kPoolSlots, next_fresh + static_cast<uint32_t>(__popcll(mask)));
                }
            }
            auto need = !path_open && slot != kInvalid;
            for (;;) {
                const auto mask = __ballot(need);
                if (mask == 0ull) { break; }
                if (q_next >= q_total) {// the item's queue is dry: on to the next item, the old one's paths finish beside the new one's
                    if (!items_left) { break; }
                    flush_tile();
                    item = next_item(CONT ? scene.wf.counts + kWfWorkCont : args.work_counter, item_count, lane);
                    q_next = 0u, q_total = 0u, s_count = 0u;
                    if (item == kInvalid) {
                        items_left = false;
                        break;
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
                    continue;
                }
                const auto avail = q_total - q_next;
                const auto rank = lane_rank(mask);
                if (need && rank < avail) {
                    const auto k = q_next + rank;
                    if (CONT) {// a path comes back from the heavy kernel: as if this slot's vertex had just been shaded
                        const auto rec = item * item_records + k;
                        const auto &q = cont_queue;
                        ray.o = q.get3(rec, 0u), ray.d = q.get3(rec, 3u);
                        ray.t_min = 0.f, ray.t_max = kFloatMax;
                        shadow.o = q.get3(rec, 6u), shadow.d = q.get3(rec, 9u);
                        shadow.t_min = 0.f, shadow.t_max = q.getf(rec, 12u);
                        nee = q.get3(rec, 13u), beta = q.get3(rec, 16u), Li = q.get3(rec, 19u);
                        pdf_bsdf = q.getf(rec, 22u);
                        pixel_index = q.get(rec, 23u);
                        const auto packed = q.get(rec, 24u);
                        depth = packed & 0xffffu;
                        want_shadow = (packed & (1u << 16u)) != 0u, want_closest = (packed & (1u << 17u)) != 0u;
                        uint32_t words[kWfSamplerWordsMax];
#pragma unroll
                        for (auto w = 0u; w < SAMPLER_WORDS; w++) { words[w] = q.get(rec, kWfContWords + w); }
                        sampler.restore(scene, words);
                        path_open = true, need = false;
                    } else {// MegakernelPathTracingInstance::Li prologue, mega_path.cpp:52-62
                        pix = k & 63u;
                        const auto px = tx * 8u + (pix & 7u), py = ty * 8u + (pix >> 3u);
                        if (px < scene.camera.width && py < scene.camera.height) {// (a pixel beyond the frame's edge has no samples: the lane asks again)
                            pixel_index = py * scene.camera.width + px;
                            path_item = item;
                            sampler.start(scene, px, py, s_begin + (k >> 6u));
                            const auto u_filter = sampler.next_pixel_2d();
                            const auto u_lens = scene.camera.kind == LR_CAMERA_THIN_LENS ? sampler.next_2d() : f2{.5f, .5f};
                            float weight;
                            camera_ray(scene, scene.filter, px, py, u_filter, u_lens, ray, weight);
                            beta = mk3(weight);
                            Li = mk3(0.f), nee = mk3(0.f);
                            pdf_bsdf = 1e16f;
                            depth = 0u;
                            path_open = true, want_shadow = false, want_closest = true, need = false;
                            if (COUNT) { local.paths++; }
                        }
                    }
                }
                q_next += min(static_cast<uint32_t>(__popcll(mask)), avail);
            }
            // ---- every slot that goes on: its state back into the record, the slot into the ray queue
            const auto t_launch = COUNT ? __builtin_readcyclecounter() : 0ull;
            const auto go = path_open && (want_shadow || want_closest);
            if (go) {
                const auto s = slots + slot * QUADS;
                uint32_t words[kWfSamplerWordsMax] = {0u, 0u, 0u, 0u};
                sampler.save(words);
                s[0] = make_float4(shadow.o.x, shadow.o.y, shadow.o.z, shadow.t_max);
                s[1] = make_float4(shadow.d.x, shadow.d.y, shadow.d.z, __uint_as_float(pixel_index));
                s[2] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.t_max);
                s[3] = make_float4(ray.d.x, ray.d.y, ray.d.z, ray.t_min);
                s[4] = make_float4(nee.x, nee.y, nee.z, pdf_bsdf);
                s[5] = make_float4(beta.x, beta.y, beta.z, __uint_as_float((depth & 0xffffu) | (pix << 16u) | (want_shadow ? 1u << 22u : 0u) | (want_closest ? 1u << 23u : 0u)));
                s[6] = make_float4(Li.x, Li.y, Li.z, __uint_as_float(words[0]));
                s[7] = make_float4(__uint_as_float(kPoolMiss), 0.f, 0.f, __uint_as_float(path_item));
                if (PCG) { s[QUADS - 1u] = make_float4(__uint_as_float(words[1]), __uint_as_float(words[2]), __uint_as_float(words[3]), 0.f); }
                if (COUNT) { local.closest_rays += want_closest ? 1u : 0u, local.shadow_rays += want_shadow ? 1u : 0u; }
            }
            {
                const auto mask = __ballot(go);
                if (go) {
                    rq[(pw.rq_head + pw.rq_count + lane_rank(mask)) & (kPoolSlots - 1u)] =
                        static_cast<uint16_t>(slot | (want_shadow ? kJobShadow : 0u) | (want_closest ? kJobClosest : 0u));
                }
                pw.rq_count += static_cast<uint32_t>(__popcll(mask));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
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
        if (pw.rq_count == 0u && pw.sq_count == 0u && !__any(tr.phase != kPhaseIdle || job != kNoJob)) { break; }// every slot of the pool is out of samples
        // ==== (B) traverse: jobs turn over inside the loop until the ray queue is dry
        TraceStats ts{0u, 0u, 0u, 0u, 0u, 0u};
        const auto t_trace = COUNT ? __builtin_readcyclecounter() : 0ull;
        for (;;) {
            pool_trace<COUNT, ALPHA, QUADS>(scene, stack, tr, job, next, slots, rq, sq, pw, ts);
            if (!ALPHA) { break; }
            if (!__any((tr.phase & kPhasePendingAlpha) != 0u)) { break; }
            resolve_pending_alpha(scene, stack, tr);
        }
        after_trace = true;
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
