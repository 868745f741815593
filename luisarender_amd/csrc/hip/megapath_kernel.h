// megapath_kernel.h — the persistent-threads megakernel path tracer for gfx950.
//
// One launch renders (tiles x sample-chunks) work items pulled from an atomic queue.  A
// wavefront owns one 8x8-pixel tile and a range of sample indices at a time: 64 x S samples, numbered
// k = 64 * (s - s_begin) + pixel.  Lanes are NOT bound to pixels: whenever a lane's path dies it takes
// the next unstarted k of the item ("path regeneration" from a wave-wide queue: ballot + prefix count,
// no atomics), so short paths (pixels that see a light or the sky) do not leave their lanes idle while
// the long ones finish.  Radiance is accumulated into a per-wave LDS copy of the tile (ds_add_f32) and
// written to the film once per item — no global film atomics (the reference does four float atomics
// per sample, src/films/color.cpp:116-121).  The order in which a wave picks and finishes samples is a
// function of the item alone (wave-internal control flow only depends on its own data), so the sums are
// bit-reproducible run to run and independent of how tiles are sharded over GPUs.
//
// Per loop iteration every lane without a ray in flight consumes its results, shades, and launches up
// to two rays which are traced back to back in ONE resumable traversal loop (dev_trace.h): the shadow
// ray of the bounce just shaded and the continuation ray.
//
// The estimator is the reference's MegakernelPathTracingInstance::Li
// (src/integrators/mega_path.cpp:49-156) and film accumulation ColorFilmInstance::_accumulate
// (src/films/color.cpp:107-130).
#pragma once
#include "dev_wavefront.h"

namespace lrd {

#ifndef LR_MIN_WAVES
#define LR_MIN_WAVES 4
#endif
#ifndef LR_REFILL
#define LR_REFILL 40
#endif

// The kernel is specialised by a FEATURE MASK (the reference JIT-compiles one kernel per scene, so a scene only pays
// for the closures / environment / alpha test it uses; here a curated set of masks is precompiled, one translation
// unit each, and lrhip_render picks the smallest superset of what the uploaded scene needs):
//   kFeatCount    diagnostics counters (tests, tools)
//   kFeatGeneric  PCG32 / Sobol / PaddedSobol sampler instead of the default xxhash32 + LCG Independent stream
//   kFeatEnv      image-based Spherical / Directional / Combined environment (constant Spherical is always in)
//   kFeatAlpha    alpha-tested traversal (Geometry::_alpha_skip) — candidate hits evaluated inside the leaf step
//   kFeatDisney   Disney closure (thick + thin)
//   kFeatMix      Mix closure (children: any non-Mix closure the mask holds)
//   kFeatLayered  Layered closure (random walk over two nested closures; implies the Disney interpreter)
//   kFeatNest     free composition of Mix and Layered (round 2): Mix trees with Layered leaves, Layered surfaces whose interfaces are
//                 Mix trees.  Its own variant: the larger call graph cost the everything-variant 9 % on a scene that does not nest
//   kFeatWf/Cont  wavefront mode (round 3): every scene with Mix or Layered surfaces.  See the enum below and dev_scene.h
//   kFeatAux      the sibling integrators that reuse this kernel's pieces: DirectLighting
//                 (src/integrators/direct.cpp:66-200) and NormalVisualizer (normal.cpp:36-70), selected at run time by
//                 scene.integrator_kind; debug / AOV views, so they only exist on top of the all-features variant
// the precompiled scene-feature sets, smallest first (each also exists x {Count} x {Generic}); csrc/hip/variants/*.hip
constexpr uint32_t kSceneVariants[] = {
    0u,
    kFeatEnv,
    kFeatAlpha,
    kFeatEnv | kFeatAlpha,
    kFeatDisney,
    kFeatEnv | kFeatDisney,
    // round 6: lean kernels that decode 8-bit texels (kFeatByteTex), both schedulers: the Disney sets, which serve every packed scene
    // without alpha tests or Mix / Layered surfaces (lrhip_upload_scene packs no other scene's images).  AHEAD of the call-making
    // variants, which decode too: the search takes the first superset
    kFeatByteTex | kFeatDisney,
    kFeatByteTex | kFeatEnv | kFeatDisney,
    kFeatEnv | kFeatAlpha | kFeatDisney | kFeatMix,
    kFeatSceneMask,
    kFeatSceneMask | kFeatNest,
    kFeatSceneMask | kFeatAux,
    kFeatVpt,
    // wavefront mode: the lean kernel that parks heavy hits (camera pass) and its continuation pass; picked by lrhip_render for scenes
    // with Mix / Layered surfaces, never by the superset search (they hold none of the closure bits)
    kFeatWf,
    kFeatEnv | kFeatWf,
    kFeatAlpha | kFeatWf,
    kFeatEnv | kFeatAlpha | kFeatWf,
    kFeatWf | kFeatCont,
    kFeatEnv | kFeatWf | kFeatCont,
    kFeatAlpha | kFeatWf | kFeatCont,
    kFeatEnv | kFeatAlpha | kFeatWf | kFeatCont,
    // round 4: the same lean kernels under the path-pool scheduler (megapool_kernel.h); lrhip_render asks for them by the kFeatPool bit
    kFeatPool,
    kFeatPool | kFeatEnv,
    kFeatPool | kFeatAlpha,
    kFeatPool | kFeatEnv | kFeatAlpha,
    kFeatPool | kFeatDisney,
    kFeatPool | kFeatEnv | kFeatDisney,
    kFeatByteTex | kFeatPool | kFeatDisney,
    kFeatByteTex | kFeatPool | kFeatEnv | kFeatDisney,
    kFeatPool | kFeatWf,
    kFeatPool | kFeatEnv | kFeatWf,
    kFeatPool | kFeatAlpha | kFeatWf,
    kFeatPool | kFeatEnv | kFeatAlpha | kFeatWf,
    kFeatPool | kFeatWf | kFeatCont,
    kFeatPool | kFeatEnv | kFeatWf | kFeatCont,
    kFeatPool | kFeatAlpha | kFeatWf | kFeatCont,
    kFeatPool | kFeatEnv | kFeatAlpha | kFeatWf | kFeatCont,
};
constexpr uint32_t kSceneVariantCount = sizeof(kSceneVariants) / sizeof(kSceneVariants[0]);

// waves per SIMD requested from the register allocator (512 VGPRs / waves).  Measured (Msamples/s at 2 / 3 / 4 waves):
// lean C2 -- / 487 / 536; environment + Disney (C4) 442 / 563 / 578; everything incl. Layered (C5) 174 / 192 / 149 in round 1
// (with the heavy closures out of line; 165 / 141 / 104 when they were inlined into the shading block).
// 3 waves (168 VGPRs) is a trap for the Layered variants: that build comes out MISCOMPILED or not depending on unrelated code
// and flags -- NaN samples all over tests/test_gpu_parity.py::test_layered_closure in round 1; fine after round 2's changes to
// the shading block (238 Msamples/s on the kitchen stand-in against 201 at 2 waves); NaN samples again once the SLP vectorizer
// was switched off (Makefile).  It is the compiler's SGPR-to-VGPR-lane spilling around the out-of-line calls: with
// -mllvm -amdgpu-spill-sgpr-to-vgpr=0 the 3-wave build is bit-identical to the 2- and 4-wave builds.  Without the SLP
// vectorizer the register pressure is low enough for 4 waves (128 VGPRs): kitchen stand-in 212 (2 waves) / 262 (3, SGPR spills to
// memory) / 274 (4), all three with identical images.  One more change to the shading block later the 4-wave <124> (not <125>)
// lost samples too, non-deterministically: every variant that makes real calls is now BUILT with that flag (Makefile:
// CALL_SAFE_FLAGS; 266 Msamples/s), and 4 waves stay.  tests/test_gpu_parity.py::test_shipped_kernels_equal_their_counting_twins
// holds every such binary to its counting twin and to itself run twice.
#ifndef LR_WAVES_LAYERED
#define LR_WAVES_LAYERED 4
#endif
// MEASURED AND NOT KEPT (round 2, profiles/archive/r02d_heavy_parking.txt): (1) the traversal as a real call in these variants, so that it
// gets a register allocation of its own (its loops then hold no spills): kitchen stand-in 202 -> 129 Msamples/s at 2 waves, 165 at
// 4 -- the state crosses the call through scratch and the loop loses its software pipelining across calls; (2) <60> at 3 waves per
// SIMD: +9 % without parking, nothing with it; (3) texture / environment code inlined in the heavy variants: basic hits 335 -> 391
// but Disney/Mix hits 248 -> 216.
#ifndef LR_HEAVY_BATCH
#define LR_HEAVY_BATCH 12// parked heavy hits that trigger the out-of-line closures (1 = never park).  Kitchen stand-in, 64 spp, Layered at 2 waves:
// 181 (never) / 193 (6) / 203 (12) / 196 (24) / 166 (40) Msamples/s; Layered at 3 waves: 228 (8) / 238 (12) / 244 (16) / 239 (24)
#endif
#ifndef LR_HEAVY_BATCH_LAYERED
#define LR_HEAVY_BATCH_LAYERED 16
#endif
#ifndef LR_WAVES_MIX
#define LR_WAVES_MIX LR_MIN_WAVES
#endif
constexpr uint32_t min_waves_of(uint32_t f) { return (f & kFeatLayered) ? LR_WAVES_LAYERED : (f & kFeatMix) ? LR_WAVES_MIX : LR_MIN_WAVES; }

// (work distribution: dev_wavefront.h next_item)
template<uint32_t F>
__global__ __launch_bounds__(kBlockThreads, min_waves_of(F)) void megapath_kernel(DScenePtr scene_ptr, RenderArgs args) {
    const DScene &scene = *(const DScene *)scene_ptr;
    constexpr bool COUNT = (F & kFeatCount) != 0u, PCG = (F & kFeatGeneric) != 0u, ENV = (F & kFeatEnv) != 0u,
                   ALPHA = (F & kFeatAlpha) != 0u, DISNEY = (F & kFeatDisney) != 0u, MIX = (F & kFeatMix) != 0u,
                   LAYERED = (F & kFeatLayered) != 0u, AUX = (F & kFeatAux) != 0u, WF = (F & kFeatWf) != 0u, CONT = (F & kFeatCont) != 0u;
    static_assert(!LAYERED || DISNEY, "the Layered interpreter instantiates the Disney closure");
    // (Disney inline in the wavefront kernels, only Mix / Layered parked, was measured: C5 at 512 spp 442 -> 386 Msamples/s -- the lean
    // kernel pays 109 spilled VGPRs for it, profiles/archive/r03i_wavefront_ab.txt)
    static_assert(!WF || !(DISNEY || MIX || LAYERED || AUX), "a wavefront variant is a lean kernel: the heavy closures live in heavy_kernel.h");
    static_assert(!CONT || WF, "the continuation pass exists in wavefront mode only");
    constexpr uint32_t SAMPLER_WORDS = PathSampler<PCG>::kSavedWords;
    constexpr int HEAVY_BATCH = LAYERED ? LR_HEAVY_BATCH_LAYERED : LR_HEAVY_BATCH;
    constexpr bool PARK_HEAVY = (MIX || LAYERED) && !AUX && HEAVY_BATCH > 1;
    __shared__ uint32_t s_stack[kStackLds * kBlockThreads];
    __shared__ float4 s_stage[kWavesPerBlock * kStageWave];// 4 KiB of node packets per wave
    const auto tid = threadIdx.x;
    const auto lane = tid & 63u;
    const auto gtid = blockIdx.x * kBlockThreads + tid;
    __shared__ float4 s_film[kWavesPerBlock * 64u];// per-wave tile accumulators (sum r, g, b, n)
    TraversalStack stack{s_stack + tid, args.spill + gtid, args.total_threads, s_stage + __builtin_amdgcn_readfirstlane(tid >> 6u) * kStageWave};
    const auto film_tile = s_film + (tid >> 6u) * 64u;
    DCounters local{};
    const auto t_wave = COUNT ? __builtin_readcyclecounter() : 0ull;

    // the continuation pass (CONT) works through the records the heavy kernel wrote this round, kWfItemRecords of them per item;
    // their number is only known on the device
    // -- and their number shrinks from round to round.  A wave works through an item 64 records at a time, each batch as long as
    // its longest path, so with few records left the items get smaller (down to one batch): a late round then takes one batch's
    // latency instead of eight (round 3: rounds with a few thousand records took 2.5 ms each with fixed 512-record items).
    const auto cont_total = CONT ? min(scene.wf.counts[kWfCountCont], scene.wf.capacity) : 0u;
    const auto cont_waves = gridDim.x * kWavesPerBlock;
    const auto item_records = CONT ? min(kWfItemRecords, max(64u, ((cont_total + cont_waves - 1u) / cont_waves + 63u) & ~63u)) : 1u;
    const auto item_count = CONT ? (cont_total + item_records - 1u) / item_records : args.item_count;
    const auto cont_queue = wf_cont_queue(scene);
    for (;;) {
        // ---- next work item of this wavefront
        uint32_t item = next_item(CONT ? scene.wf.counts + kWfWorkCont : args.work_counter, item_count, lane);
        if (item == kInvalid) { break; }
        const auto range = CONT ? ItemRange{0u, 0u, 0u, 0u} : item_range(args, item);
        const auto tile_index = range.tile_index, chunk = range.chunk;
        const auto tile = args.tile_begin + tile_index * args.tile_stride;
        const auto ty = tile / args.tiles_x, tx = (tile - ty * args.tiles_x + ty) % args.tiles_x;// row ty is rotated by ty (lrhip.h)
        const auto s_begin = range.s_begin, s_end = range.s_end;
        // the item's sample queue: k = 64 * (s - s_begin) + pixel_in_tile (CONT: record item * item_records + k)
        const auto q_total = CONT ? min(item_records, cont_total - item * item_records) : (s_end > s_begin ? (s_end - s_begin) * 64u : 0u);
        auto q_next = 0u;// wave-uniform
        film_tile[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto px = 0u, py = 0u;
        auto pixel = film_tile;// LDS accumulator of the pixel this lane's current sample belongs to
        auto pixel_index = 0u; // WF: the same pixel in the film (a parked path takes it along)
        // ---- per-lane path state
        PathSampler<PCG> sampler{};
        TravState tr{};
        tr.phase = kPhaseIdle;
        Ray ray{};// continuation ray waiting behind a shadow ray in flight
        f3 beta = mk3(0.f), Li = mk3(0.f), nee = mk3(0.f);
        auto pdf_bsdf = 1e16f;
        auto depth = 0u;
        auto path_open = false, traced_shadow = false, traced_closest = false;
        for (;;) {
            // ==== (A) lanes without a ray in flight: consume results and shade
            auto want_shadow = false, want_closest = false;
            Ray shadow{};
            const auto t_shade = COUNT ? __builtin_readcyclecounter() : 0ull;
            // ---- heavy hits wait for company.  In a variant with out-of-line closures (Mix / Layered present) a hit on a
            // Disney / Mix / Layered surface costs the WAVE the whole heavy call, however few lanes take it -- and with ~8 % heavy
            // hits nearly every shading round has one or two.  Such a lane stays parked (idle, hit and shadow result kept in its
            // state, exactly like a lane that has not been shaded yet) until LR_HEAVY_BATCH lanes are parked or the wave has
            // nothing else to do; then they are shaded together.  A sample's value does not depend on when it is shaded.
            auto parked = false;
            if (PARK_HEAVY) {
                const auto ready = tr.phase == kPhaseIdle && traced_closest;
                auto heavy_hit = false;
                if (ready && tr.hit.inst != kInvalid) {
#if LR_BAKED_SHADING
                    const auto rec = reinterpret_cast<const float4 *>(scene.shade_tris + tr.hit.tri);
                    const auto flags = __float_as_uint(rec[0].w), tags = __float_as_uint(rec[1].w);
#else
                    const auto h = reinterpret_cast<const uint4 *>(scene.instances + tr.hit.inst)[0];
                    const auto flags = h.x & 1023u, tags = h.y;
#endif
                    heavy_hit = (flags & LR_SHAPE_HAS_SURFACE) != 0u && scene.closures[(tags >> 12u) & 4095u].kind >= LR_SURFACE_DISNEY;
                }
                const auto heavy_lanes = __popcll(lr_ballot(heavy_hit));
                if (heavy_lanes > 0 && heavy_lanes < HEAVY_BATCH) {
                    const auto others = lr_any(tr.phase != kPhaseIdle || (ready && !heavy_hit) || (tr.phase == kPhaseIdle && !path_open && q_next < q_total));
                    parked = heavy_hit && others;
                }
            }
            unsigned long long t_closure_sum = 0ull;// (COUNT: wave cycles inside the closure section of this round; lanes agree)
            auto park_kind = kInvalid;// WF: closure kind (0 Disney, 1 Mix, 2 Layered) of the heavy surface this lane's path just reached
            if (tr.phase == kPhaseIdle && !parked) {
                if (traced_shadow) {// direct lighting of the bounce that spawned the shadow ray, mega_path.cpp:124-130
                    if (!tr.occluded) { Li += nee; }
                    traced_shadow = false;
                }
                if (traced_closest) {// one iteration of the reference's depth loop, mega_path.cpp:63-154
                    traced_closest = false;
                    if (COUNT) { local.shade_busy++; }
#ifdef LR_PROBE_SHADE
                    {// sensitivity probe: LR_PROBE_SHADE extra dependent VALU ops per shaded vertex
                        float dummy = pdf_bsdf;
#pragma unroll
                        for (auto i = 0; i < LR_PROBE_SHADE; i++) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(dummy)); }
                        asm volatile("" ::"v"(dummy));
                    }
#endif
                    auto wo = -tr.d;
                    auto hit = tr.hit;
                    auto hit_valid = hit.inst != kInvalid;
                    // sibling integrators (kFeatAux): DirectLighting = this loop cut after the BSDF-sampled vertex's emission,
                    // with the estimator chosen by importance_sampling; NormalVisualizer = the first hit's normal
                    const auto is_direct = AUX && scene.integrator_kind == LR_INTEGRATOR_DIRECT;
                    const auto is_normal = AUX && scene.integrator_kind == LR_INTEGRATOR_NORMAL;
                    const auto direct_lights = !is_direct || (scene.integrator_flags & LR_DIRECT_SAMPLE_LIGHTS) != 0u;
                    const auto direct_surfaces = !is_direct || (scene.integrator_flags & LR_DIRECT_SAMPLE_SURFACES) != 0u;
                    // MIS weight of an emission found by BSDF sampling: 1 for camera rays (pdf_bsdf = 1e16) and, in
                    // DirectLighting, when lights are not sampled (direct.cpp:187-189)
                    auto mis_bsdf = [&](float pdf_light) { return direct_lights ? balance(pdf_bsdf, pdf_light) : 1.f; };
                    if (!hit_valid && scene.env_kind != kEnvNone && !is_normal) {// miss, mega_path.cpp:70-76 -> evaluate_miss, uniform.cpp:67-76
                        f3 L = mk3(scene.env_L[0], scene.env_L[1], scene.env_L[2]);
                        auto pdf = kInvPi * 0.25f;
                        if (ENV && scene.env_kind != kEnvConstant) { env_evaluate(scene, tr.d, L, pdf); }
                        // (DirectLighting's camera-ray miss adds eval.L unweighted, :92-98: the same thing, pdf_bsdf = 1e16)
                        Li += beta * L * mis_bsdf(pdf * scene.env_prob);
                    }
                    SurfacePoint it;
                    auto has_surface = false;
                    if (hit_valid) {
#if LR_BAKED_SHADING
                        reconstruct_baked(scene, hit.tri, hit.u, hit.v, it);
#else
                        reconstruct<true>(scene, hit.inst, hit.prim, mk3(1.f - hit.u - hit.v, hit.u, hit.v), it);
#endif
                        it.back_facing = dot(wo, it.ng) < 0.0f;
                        if (COUNT) { local.surface_hits++; }
                        if (scene.has_lights && (it.flags & LR_SHAPE_HAS_LIGHT)) {// hit light, mega_path.cpp:79-86
                            f3 L;
                            float pdf;
                            light_evaluate(scene, it, hit.prim, tr.o, L, pdf);
                            pdf *= (1.f - scene.env_prob) / static_cast<float>(scene.light_count);
                            if (!is_direct || depth == 0u || pdf > 0.f) { Li += beta * L * mis_bsdf(pdf); }// (direct.cpp:185: light_eval.pdf > 0)
                        }
                        has_surface = (it.flags & LR_SHAPE_HAS_SURFACE) != 0u;
                        if (is_direct && depth >= 1u) { has_surface = false; }// direct.cpp:194: the loop ends after the sampled vertex
                        if (is_normal) {// normal.cpp:48-66
                            auto ns = it.ng;
                            if (scene.integrator_flags & LR_NORMAL_SHADING) {
                                ns = it.shading.n;
                                if (has_surface) {
                                    DClosure c;
                                    Frame fr;
                                    load_lobe(LobeTables{scene.closures, scene.surfaces, scene.textures, scene.texels}, it.uv, it.ng, wo,
                                              (it.tags >> 12u) & 4095u, it.shading, c, fr);
                                    ns = fr.n;
                                }
                            }
                            if (scene.integrator_flags & LR_NORMAL_REMAP) { ns = ns * .5f + mk3(.5f); }
                            Li = beta * ns;
                            has_surface = false;
                        }
                    }
                    // wavefront mode: a Disney / Mix / Layered surface is not shaded here.  The path -- direction, hit, throughput,
                    // radiance so far (the emission of this vertex included), sampler position -- goes into the queue of its
                    // closure kind; heavy_kernel.h shades the vertex and hands the path back as a continuation record.
                    if (WF && has_surface) {
#if LR_BAKED_SHADING
                        const auto heavy_kind = (it.flags >> 10u) & 3u;// (baked into the triangle's record: lrhip.hip, build_shade_tris)
                        if (heavy_kind != 0u) { park_kind = heavy_kind - 1u, has_surface = false; }
#else
                        const auto kind = scene.closures[(it.tags >> 12u) & 4095u].kind;
                        if (kind >= LR_SURFACE_DISNEY) { park_kind = kind - LR_SURFACE_DISNEY, has_surface = false; }
#endif
                    }
                    if (has_surface) {
                        if (COUNT) { local.path_length_sum++, local.nee_samples++; }
                        // random numbers are drawn where they are used, in the reference's order (mega_path.cpp:90-97):
                        // light selection, light surface (2), lobe, bsdf (2), [rr]
                        // (DirectLighting in surface-only mode draws no light sample, direct.cpp:114-118)
                        auto u_light_selection = direct_lights ? sampler.next_1d() : 0.f;
                        auto u_light_surface = direct_lights ? sampler.next_2d() : f2{0.f, 0.f};
                        // ---- sample one light, uniform.cpp:78-137 + light_sampler.cpp:57-63 (dev_shade.h: sample_one_light)
                        f3 light_L = mk3(0.f);
                        auto light_pdf = 0.f;
                        if (direct_lights) {
                            auto pick = sample_one_light<ENV>(scene, it, u_light_selection, u_light_surface);
                            shadow = pick.shadow, light_L = pick.L, light_pdf = pick.pdf;
                        }
                        const auto t_closure = COUNT ? __builtin_readcyclecounter() : 0ull;
                        // ---- material, mega_path.cpp:111-143.  The five basic closures are evaluated inline; Disney / Mix /
                        // Layered surfaces go through the out-of-line heavy path (dev_heavy.h) when this variant holds Mix or
                        // Layered (HEAVY_CALL), so that their registers are not the main loop's.  A <Disney only> variant keeps
                        // Disney inline (C4: 584 Msamples/s inline).
                        constexpr bool HEAVY_CALL = MIX || LAYERED;
                        const LobeTables tables{scene.closures, scene.surfaces, scene.textures, scene.texels};
                        DClosure closure;
                        Frame sh;
                        load_lobe(tables, it.uv, it.ng, wo, (it.tags >> 12u) & 4095u, it.shading, closure, sh);
                        const auto is_heavy = HEAVY_CALL && closure.kind >= LR_SURFACE_DISNEY;
                        HeavyCtx heavy;
                        if (is_heavy) {
                            heavy.tb = tables, heavy.uv = it.uv, heavy.ng = it.ng, heavy.p = it.p, heavy.wo = wo;
                            heavy.shading = sh, heavy.closure = closure;
                        }
                        if (light_pdf > 0.0f) {
                            BsdfEval eval;
                            if (is_heavy) { eval = heavy_evaluate<MIX, LAYERED>(&heavy, shadow.d); }
                            else { eval = closure_evaluate<DISNEY && !HEAVY_CALL>(closure, sh, it.ng, wo, shadow.d); }
                            auto w = (direct_surfaces ? balance(light_pdf, eval.pdf) : 1.f) / light_pdf;// (direct.cpp:151-153)
                            nee = w * beta * eval.f * light_L;
                            if (is_direct && !(eval.pdf > 0.f)) { nee = mk3(0.f); }// direct.cpp:150
                            // the reference traces the shadow ray unconditionally; a zero contribution cannot change Li
                            want_shadow = nee.x != 0.f || nee.y != 0.f || nee.z != 0.f;
                        }
                        auto u_lobe = sampler.next_1d();
                        auto u_bsdf = direct_surfaces ? sampler.next_2d() : f2{0.f, 0.f};
                        BsdfSample bs;
                        auto has_eta = false;
                        auto eta = 1.f;
                        if (is_heavy) {
                            auto hs = heavy_sample<MIX, LAYERED>(&heavy, u_lobe, u_bsdf);
                            bs = hs.bs, has_eta = hs.has_eta != 0u, eta = hs.eta;
                        } else {
                            bs = closure_sample<DISNEY && !HEAVY_CALL>(closure, sh, it.ng, wo, u_lobe, u_bsdf);
                            has_eta = closure_eta(closure, eta);
                        }
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
                        if (is_direct && !(bs.pdf > 0.f)) { alive = false; }// direct.cpp:185: surface_sample.eval.pdf > 0
                        if (!direct_surfaces) { alive = false; }            // light-only sampling: no continuation ray
                        auto rr = depth + 1u >= scene.rr_depth;// Russian roulette, mega_path.cpp:148-153
                        auto u_rr = 0.f;
                        if (rr) { u_rr = sampler.next_1d(); }// (drawn before the closure in the reference: same stream position)
                        if (alive) {
                            auto q = fmaxf(max_component(beta) * eta_scale, .05f);
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
                if (WF && park_kind != kInvalid) { path_open = false; }// (it goes on elsewhere: nothing to accumulate here)
                if (path_open && !want_shadow && !want_closest) {// path complete: film.accumulate (integrator.cpp:74)
                    if (CONT) { wf_film_accumulate(scene, args.film, pixel_index, Li * scene.shutter_weight, scene.film_clamp); }
                    else { film_accumulate(pixel, Li * scene.shutter_weight, scene.film_clamp); }
                    path_open = false;
                }
            }
            if (WF) {// ---- park: one atomic per closure kind and wave, field-major stores (coalesced over the parking lanes)
                if (lr_any(park_kind != kInvalid)) {
#pragma unroll
                    for (auto k = 0u; k < kWfKinds; k++) {
                        const auto mask = lr_ballot(park_kind == k);
                        if (mask == 0ull) { continue; }
                        const auto slot = wf_reserve(scene.wf.counts + kWfCountHeavy + k, mask, lane);
                        if (park_kind == k && slot < scene.wf.capacity) {// (capacity >= the slice's paths: never full; a bound, not a policy)
                            const auto q = wf_heavy_queue<SAMPLER_WORDS>(scene, k);
                            q.put3(slot, 0u, tr.d);// (the hit and the ray are still in the traversal state: the launch below resets them)
                            q.put(slot, 3u, tr.hit.tri), q.put(slot, 4u, tr.hit.u), q.put(slot, 5u, tr.hit.v);
                            q.put3(slot, 6u, beta), q.put3(slot, 9u, Li);
                            q.put(slot, 12u, pixel_index), q.put(slot, 13u, depth);
                            uint32_t words[kWfSamplerWordsMax];
                            sampler.save(words);
#pragma unroll
                            for (auto w = 0u; w < SAMPLER_WORDS; w++) { q.put(slot, kWfHeavyWords + w, words[w]); }
                        }
                    }
                }
            }
            // ==== (A') path regeneration: lanes with no path take the next samples of the item's queue, in lane order
            const auto t_regen = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (COUNT) {// (the closure section is timed by the lanes that ran it: lane 0 reports the wave's figure)
                for (auto off = 32; off > 0; off >>= 1) { t_closure_sum = max(t_closure_sum, static_cast<unsigned long long>(__shfl_xor(static_cast<long long>(t_closure_sum), off))); }
            }
            {
                const auto need = tr.phase == kPhaseIdle && !path_open;
                const auto mask = lr_ballot(need);
                const auto k = q_next + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32u), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
                q_next = min(q_next + static_cast<uint32_t>(__popcll(mask)), q_total);
                if (CONT) {
                    if (need && k < q_total) {// a path comes back from the heavy kernel: as if this lane had just shaded its vertex
                        const auto slot = item * item_records + k;
                        const auto &q = cont_queue;
                        ray.o = q.get3(slot, 0u), ray.d = q.get3(slot, 3u);
                        ray.t_min = 0.f, ray.t_max = kFloatMax;
                        shadow.o = q.get3(slot, 6u), shadow.d = q.get3(slot, 9u);
                        shadow.t_min = 0.f, shadow.t_max = q.getf(slot, 12u);
                        nee = q.get3(slot, 13u), beta = q.get3(slot, 16u), Li = q.get3(slot, 19u);
                        pdf_bsdf = q.getf(slot, 22u);
                        pixel_index = q.get(slot, 23u);
                        const auto packed = q.get(slot, 24u);
                        depth = packed & 0xffffu;
                        want_shadow = (packed & (1u << 16u)) != 0u, want_closest = (packed & (1u << 17u)) != 0u;
                        uint32_t words[kWfSamplerWordsMax];
#pragma unroll
                        for (auto w = 0u; w < SAMPLER_WORDS; w++) { words[w] = q.get(slot, kWfContWords + w); }
                        sampler.restore(scene, words);
                        path_open = true;
                    }
                } else if (need && k < q_total) {// MegakernelPathTracingInstance::Li prologue, mega_path.cpp:52-62
                    const auto pix = k & 63u;
                    px = tx * 8u + (pix & 7u), py = ty * 8u + (pix >> 3u);
                    pixel = film_tile + pix;
                    if (WF) { pixel_index = py * scene.camera.width + px; }
                    if (px < scene.camera.width && py < scene.camera.height) {
                        sampler.start(scene, px, py, s_begin + (k >> 6u));
                        auto u_filter = sampler.next_pixel_2d();
                        auto u_lens = scene.camera.kind == LR_CAMERA_THIN_LENS ? sampler.next_2d() : f2{.5f, .5f};
                        float weight;
                        camera_ray(scene, scene.filter, px, py, u_filter, u_lens, ray, weight);
                        beta = mk3(weight);
                        Li = mk3(0.f);
                        pdf_bsdf = 1e16f;
                        depth = 0u;
                        path_open = true, want_closest = true;
                        if (COUNT) { local.paths++; }
                    }
                }
            }
            const auto t_launch = COUNT ? __builtin_readcyclecounter() : 0ull;
            if (tr.phase == kPhaseIdle) {
                // ---- launch: shadow ray first, the continuation ray follows inside the traversal loop
                if (want_shadow || want_closest) {
                    tr.hit.inst = kInvalid, tr.hit.prim = kInvalid, tr.hit.u = 0.f, tr.hit.v = 0.f;
                    tr.occluded = false;
                    traced_shadow = want_shadow, traced_closest = want_closest;
                    if (want_shadow) { trav_begin(tr, shadow, kPhaseShadow); }
                    else { trav_begin(tr, ray, kPhaseClosest); }
                    if (COUNT) { local.closest_rays += want_closest ? 1u : 0u, local.shadow_rays += want_shadow ? 1u : 0u; }
                }
            }
            if (!lr_any(tr.phase != kPhaseIdle)) {
                if (PARK_HEAVY && lr_any(parked)) { continue; }// parked paths are left: shade them now
                break;// every lane of the tile is out of samples
            }
            // ==== (B) traverse until `refill` lanes have results to shade
            TraceStats ts{0u, 0u, 0u, 0u, 0u, 0u};
            const auto t_trace = COUNT ? __builtin_readcyclecounter() : 0ull;
            trace_until_refill<COUNT, ALPHA>(scene, stack, tr, traced_closest, ray, LR_REFILL, ts);
            if (COUNT) {
                if (lane == 0u) {
                    local.shade_cycles += t_trace - t_shade, local.trace_cycles += __builtin_readcyclecounter() - t_trace;
                    local.shade_closure_cycles += t_closure_sum, local.shade_regen_cycles += t_launch - t_regen;
                    local.shade_light_cycles += (t_regen - t_shade) - t_closure_sum;// everything of (A) that is not the closure section
                }
                local.nodes_visited += ts.nodes, local.tris_tested += ts.tris, local.nodes_empty += ts.nodes_empty;
                local.trace_steps += ts.steps, local.trace_steps_busy += ts.steps_busy, local.trace_steps_starved += ts.steps_starved;
                local.shade_calls++;
            }
        }
        // ---- item complete: lane l adds pixel l of the tile to the film (or stores this chunk's partial plane)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (!CONT) {
            const auto wx = tx * 8u + (lane & 7u), wy = ty * 8u + (lane >> 3u);
            if (wx < scene.camera.width && wy < scene.camera.height) {
                const auto acc = film_tile[lane];
                const auto index = wy * scene.camera.width + wx;
                if (args.chunk_count == 1u) {
                    auto f = args.film[index];
                    f.x += acc.x, f.y += acc.y, f.z += acc.z, f.w += acc.w;
                    args.film[index] = f;
                } else {
                    args.partial[static_cast<size_t>(chunk) * scene.camera.width * scene.camera.height + index] = acc;
                }
            }
        }
    }

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
