// heavy_kernel.h — the heavy-closure kernel of wavefront mode (dev_scene.h: WfArgs; round 3).
//
// The lean megakernel parks every path that reaches a Disney / Mix / Layered surface in the HBM queue of that closure kind
// (megapath_kernel.h, kFeatWf).  This kernel shades those vertices: one lane per parked path, 64 paths of ONE closure kind per
// wave, so the Layered random walk or the Mix interpreter runs for full waves instead of for the two or three lanes of a
// megakernel wave that happened to hit such a surface -- and with a register allocation of its own, which is what the heavy
// closures cost the megakernel variants that held them (round 2: <124> carried a 2.4 KB/lane spill frame and its TRAVERSAL loop ran
// 2.2x slower per step than the lean kernel's, profiles/archive/r03a stats).  Per vertex it does what the megakernel's shading block does
// for a basic closure -- one iteration of the reference's depth loop from the light sample on (src/integrators/mega_path.cpp:88-154):
// light selection + light sample (uniform.cpp:78-137), closure evaluate (NEE, MIS), closure sample, throughput, Russian roulette --
// with the random numbers drawn in the reference's order from the path's own stream, and writes a CONTINUATION record (shadow ray +
// next ray) that the megakernel's continuation pass traces (kFeatCont); a path that needs no ray ends here (film add).
// The reference's wavefront integrator queues paths per surface tag for the same reason: src/integrators/wave_path_v2.cpp:419-440.
//
// Work distribution: chunks of 64 records of the kernel's closure kind; waves draw chunks from an atomic counter.
#pragma once
#include "dev_wavefront.h"

namespace lrd {

#ifndef LR_HEAVY_WAVES
#define LR_HEAVY_WAVES 3// waves per SIMD the register allocator may assume (3: 168 VGPRs.  C5 at 1024 spp, round 5: 555.5 / 555.7 / 567.3 / 562.6 Msamples/s at 1 / 2 / 3 / 4, profiles/r05j_c5_heavy_waves.txt)
#endif

// One instantiation per closure KIND (0 Disney, 1 Mix, 2 Layered): the Disney kernel holds the Disney closure inline and nothing
// else (no calls, a fraction of the registers and of the 1.8 KB of scratch the interpreters' records take), the Mix kernel the Mix
// interpreter, the Layered kernel the random walk; a round launches the three back to back (an empty queue costs microseconds).
// F: kFeatCount (diagnostics counters), kFeatGeneric (PCG32 / Sobol / PaddedSobol sampler), kFeatNest (Mix / Layered nested in each
// other; the translation unit defines LR_NEST to match, dev_layered.h)
template<uint32_t F, uint32_t KIND>
__global__ __launch_bounds__(kBlockThreads, LR_HEAVY_WAVES) void heavy_kernel(DScenePtr scene_ptr, RenderArgs args) {
    const DScene &scene = *(const DScene *)scene_ptr;
    constexpr bool COUNT = (F & kFeatCount) != 0u, PCG = (F & kFeatGeneric) != 0u;
    constexpr uint32_t SAMPLER_WORDS = PathSampler<PCG>::kSavedWords;
    static_assert(KIND < kWfKinds, "closure kind: 0 Disney, 1 Mix, 2 Layered");
    const auto lane = threadIdx.x & 63u;
    const auto count = min(scene.wf.counts[kWfCountHeavy + KIND], scene.wf.capacity);
    const auto total_chunks = (count + 63u) / 64u;
    const auto cont = wf_cont_queue(scene);
    const auto q = wf_heavy_queue<SAMPLER_WORDS>(scene, KIND);
    unsigned long long n_vertices = 0ull;
    for (;;) {
        uint32_t chunk = 0u;
        if (lane == 0u) { chunk = atomicAdd(scene.wf.counts + kWfWorkHeavy + KIND, 1u); }
        chunk = static_cast<uint32_t>(__shfl(static_cast<int>(chunk), 0));
        if (chunk >= total_chunks) { break; }
        const auto slot = chunk * 64u + lane;
        const auto active = slot < count;
        auto want_shadow = false, want_closest = false;
        Ray ray{}, shadow{};
        f3 beta = mk3(0.f), Li = mk3(0.f), nee = mk3(0.f);
        auto pdf_bsdf = 0.f;
        auto pixel_index = 0u, depth = 0u;
        PathSampler<PCG> sampler{};
        if (active) {
            const auto d = q.get3(slot, 0u);
            const auto tri = q.get(slot, 3u);
            const auto hu = q.getf(slot, 4u), hv = q.getf(slot, 5u);
            beta = q.get3(slot, 6u), Li = q.get3(slot, 9u);
            pixel_index = q.get(slot, 12u), depth = q.get(slot, 13u);
            uint32_t words[kWfSamplerWordsMax];
#pragma unroll
            for (auto w = 0u; w < SAMPLER_WORDS; w++) { words[w] = q.get(slot, kWfHeavyWords + w); }
            sampler.restore(scene, words);
            if (COUNT) { n_vertices++; }
            // ---- the vertex, as the megakernel left it: hit reconstruction (the emission of the hit is already in Li)
            const auto wo = -d;
            SurfacePoint it;
            reconstruct_baked(scene, tri, hu, hv, it);
            it.back_facing = dot(wo, it.ng) < 0.0f;
            // ---- mega_path.cpp:90-97: light selection, light surface (2), lobe, bsdf (2), [rr] -- drawn in this order
            const auto u_light_selection = sampler.next_1d();
            const auto u_light_surface = sampler.next_2d();
            const auto pick = sample_one_light<true>(scene, it, u_light_selection, u_light_surface);// uniform.cpp:78-137
            shadow = pick.shadow;
            const LobeTables tables{scene.closures, scene.surfaces, scene.textures, scene.texels};
            HeavyCtx heavy;
            load_lobe(tables, it.uv, it.ng, wo, (it.tags >> 12u) & 4095u, it.shading, heavy.closure, heavy.shading);
            heavy.tb = tables, heavy.uv = it.uv, heavy.ng = it.ng, heavy.p = it.p, heavy.wo = wo;
            // the interpreters this kind needs: Disney none (inline closure); a Mix tree may hold Layered leaves and a Layered surface
            // Mix interfaces only in the nested variants
            constexpr bool MIX = KIND == 1u || (KIND == 2u && LR_NEST != 0), LAYERED = KIND == 2u || (KIND == 1u && LR_NEST != 0);
            if (pick.pdf > 0.0f) {// mega_path.cpp:111-130
                BsdfEval eval;
                if constexpr (KIND == 0u) { eval = closure_evaluate<true>(heavy.closure, heavy.shading, it.ng, wo, shadow.d); }
                else { eval = heavy_evaluate<MIX, LAYERED>(&heavy, shadow.d); }
                const auto w = balance(pick.pdf, eval.pdf) / pick.pdf;
                nee = w * beta * eval.f * pick.L;
                // the reference traces the shadow ray unconditionally; a zero contribution cannot change Li
                want_shadow = nee.x != 0.f || nee.y != 0.f || nee.z != 0.f;
            }
            const auto u_lobe = sampler.next_1d();
            const auto u_bsdf = sampler.next_2d();
            HeavySample hs;// mega_path.cpp:132-143
            if constexpr (KIND == 0u) {
                hs.bs = closure_sample<true>(heavy.closure, heavy.shading, it.ng, wo, u_lobe, u_bsdf);
                hs.eta = 1.f;
                hs.has_eta = closure_eta(heavy.closure, hs.eta) ? 1u : 0u;
            } else {
                hs = heavy_sample<MIX, LAYERED>(&heavy, u_lobe, u_bsdf);
            }
            const auto bs = hs.bs;
            ray.o = robust_origin(it, bs.wi);// spawn_ray, interaction.cpp:21-23
            ray.d = bs.wi;
            ray.t_min = 0.f, ray.t_max = kFloatMax;
            pdf_bsdf = bs.pdf;
            beta *= (bs.pdf > 0.f ? 1.f / bs.pdf : 0.f) * bs.f;
            auto eta_scale = 1.f;
            if (hs.has_eta != 0u) {
                if (bs.event == kEventEnter) { eta_scale = sqr(hs.eta); }
                else if (bs.event == kEventExit) { eta_scale = sqr(1.f / hs.eta); }
            }
            if (any_nan(beta)) { beta = mk3(0.f); }// zero_if_any_nan
            auto alive = !(beta.x <= 0.f && beta.y <= 0.f && beta.z <= 0.f);
            const auto rr = depth + 1u >= scene.rr_depth;// Russian roulette, mega_path.cpp:148-153
            auto u_rr = 0.f;
            if (rr) { u_rr = sampler.next_1d(); }// (drawn before the closure in the reference: same stream position)
            if (alive) {
                const auto qq = fmaxf(max_component(beta) * eta_scale, .05f);
                if (rr) {
                    if (qq < scene.rr_threshold && u_rr >= qq) { alive = false; }
                    else { beta *= qq < scene.rr_threshold ? 1.0f / qq : 1.f; }
                }
            }
            depth++;
            want_closest = alive && depth < scene.max_depth;
            if (!want_shadow && !want_closest) {// path complete: film.accumulate (integrator.cpp:74)
                wf_film_accumulate(scene, args.film, pixel_index, Li * scene.shutter_weight, scene.film_clamp);
            }
        }
        // ---- the paths that go on: continuation records, one atomic per wave
        const auto go_on = want_shadow || want_closest;
        const auto mask = lr_ballot(go_on);
        if (mask != 0ull) {
            const auto out = wf_reserve(scene.wf.counts + kWfCountCont, mask, lane);
            if (go_on && out < scene.wf.capacity) {
                cont.put3(out, 0u, ray.o), cont.put3(out, 3u, ray.d);
                cont.put3(out, 6u, shadow.o), cont.put3(out, 9u, shadow.d), cont.put(out, 12u, shadow.t_max);
                cont.put3(out, 13u, nee), cont.put3(out, 16u, beta), cont.put3(out, 19u, Li);
                cont.put(out, 22u, pdf_bsdf), cont.put(out, 23u, pixel_index);
                cont.put(out, 24u, depth | (want_shadow ? 1u << 16u : 0u) | (want_closest ? 1u << 17u : 0u));
                uint32_t words[kWfSamplerWordsMax];
                sampler.save(words);
#pragma unroll
                for (auto w = 0u; w < SAMPLER_WORDS; w++) { cont.put(out, kWfContWords + w, words[w]); }
            }
        }
    }
    if (COUNT) {// the vertices shaded here are path vertices with a light sample each, like the megakernel's (megapath_kernel.h)
        for (auto off = 32; off > 0; off >>= 1) { n_vertices += __shfl_down(n_vertices, off); }
        if (lane == 0u && n_vertices != 0ull) {
            atomicAdd(&args.counters->path_length_sum, n_vertices);
            atomicAdd(&args.counters->nee_samples, n_vertices);
        }
    }
}

}// namespace lrd
