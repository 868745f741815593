// megavpt_kernel.h — the volumetric megakernel for gfx950: MegakernelVolumePathTracingNaive
// (src/integrators/mega_vpt_naive.cpp:68-483) over Homogeneous / Vacuum media (src/media/homogeneous.cpp:48-137,
// vacuum.cpp), the Henyey-Greenstein phase function (src/phasefunctions/henyey_greenstein.cpp:22-45) and the priority
// MediumTracker (src/util/medium_tracker.cpp).  It reuses the hot path's pieces — camera, samplers, the resumable BVH4
// traversal, hit reconstruction, light sampling, every surface closure, the LDS film tile — but not its scheduling:
//
//   * the reference's `_transmittance` walks a shadow segment through EVERY surface on it with closest-hit traces, so one
//     bounce needs an unbounded number of dependent traces.  Each lane therefore runs a small state machine
//     (BEGIN -> MAIN trace -> [medium NEE walk] -> surface -> [surface NEE walk] -> shade -> END) that returns to the wave
//     loop whenever it needs a trace; the wave traces all pending rays together (trace_steps) and re-enters;
//   * lanes are bound to pixels and samples run one after the other (no sample queue): this integrator is a
//     feature row, not the headline path — parity first;
//   * the two MediumTrackers (32 x (priority, tag), the path's and the walk's copy) live in scratch.
//
// Conscious deviation shared with the oracle: the walk stops after kMaxCrossings surfaces (the reference loops while any
// transmittance channel is positive, which never ends for a segment lying in a surface).
#pragma once
#include "megapath_kernel.h"// film_accumulate, balance, the feature bits

namespace lrd {

struct DevPCG32 {// src/util/rng.cpp:142-176
    uint64_t state, inc;
    LR_D uint32_t uniform_uint() {
        auto oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dull + inc;
        auto xorshifted = static_cast<uint32_t>(((oldstate >> 18u) ^ oldstate) >> 27u);
        auto rot = static_cast<uint32_t>(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    LR_D void set_sequence(uint64_t init_seq) {
        state = 0u;
        inc = (init_seq << 1u) | 1u;
        (void)uniform_uint();
        state += 0x853c49e6748fea9bull;
        (void)uniform_uint();
    }
    LR_D float uniform_float() { return uint_to_unit_float(uniform_uint()); }
};

// MediumTracker, medium_tracker.cpp:12-85.  MediumInfo = (priority, tag) with the priority equal to the list's, so only
// tags are stored.  Entries at and beyond `size` are (VACUUM_PRIORITY, INVALID_TAG) and the list is sorted by ascending
// priority, so the reference's fixed 32-iteration loops are equivalent to loops over the first size + 1 entries.
struct DMediumTracker {
    static constexpr uint32_t capacity = 32u;
    uint32_t priority[capacity];
    uint32_t tag[capacity];
    uint32_t size;
    LR_D void init() {
        for (auto i = 0u; i < capacity; i++) { priority[i] = LR_MEDIUM_VACUUM_PRIORITY, tag[i] = LR_INVALID_ID; }
        size = 0u;
    }
    LR_D bool vacuum() const { return priority[0] == LR_MEDIUM_VACUUM_PRIORITY; }
    LR_D bool true_hit(uint32_t p) const { return p <= priority[0]; }
    LR_D uint32_t current_tag() const { return vacuum() ? LR_INVALID_ID : tag[0]; }
    LR_D void enter(uint32_t p, uint32_t t) {
        if (size == capacity) { return; }
        size++;
        auto x = p, v = t;
        auto n = min(size + 1u, capacity);
        for (auto i = 0u; i < n; i++) {
            auto pi = priority[i], ti = tag[i];
            auto should_swap = pi > x;
            priority[i] = should_swap ? x : pi;
            tag[i] = should_swap ? v : ti;
            x = should_swap ? pi : x;
            v = should_swap ? ti : v;
        }
    }
    LR_D void exit(uint32_t p, uint32_t t) {
        auto remove_num = 0u;
        auto n = min(size + 1u, capacity - 1u);
        for (auto i = 0u; i < n; i++) {
            auto should_remove = priority[i] == p && tag[i] == t && remove_num == 0u;
            remove_num += should_remove ? 1u : 0u;
            priority[i] = priority[i + remove_num];
            tag[i] = tag[i + remove_num];
        }
        if (remove_num != 0u) {
            size--;
            priority[size] = LR_MEDIUM_VACUUM_PRIORITY, tag[size] = LR_INVALID_ID;
        }
    }
};

enum : uint32_t { kMediumAbsorb = 0u, kMediumScatter = 1u, kMediumHitSurface = 3u, kMediumInvalid = ~0u };// medium.h:28-32

struct DMediumSample {// Medium::Sample::zero, medium.h:56-61
    f3 f;
    float pdf;
    Ray ray;
    uint32_t event;
};

LR_D f3 vpt_channel_pdf(DevPCG32 &rng) {// homogeneous.cpp:50-54
    auto a = rng.uniform_float(), b = rng.uniform_float(), c = rng.uniform_float();
    auto inv = a + b + c;
    return mk3(a / inv, b / inv, c / inv);
}
LR_D float comp(f3 v, uint32_t i) { return i == 0u ? v.x : (i == 1u ? v.y : v.z); }
LR_D uint32_t vpt_sample_discrete3(f3 w, float u) {// sampling.cpp:182-194
    auto u_rescaled = u * (w.x + w.y + w.z);
    auto accum = w.x;
    if (u_rescaled <= accum) { return 0u; }
    accum += w.y;
    if (u_rescaled <= accum) { return 1u; }
    return 2u;// (third channel; also the reference's out-of-range "-1" by rounding)
}

// HomogeneousMediumClosure::sample, homogeneous.cpp:48-122
LR_D DMediumSample vpt_medium_sample(const lr_medium &m, const Ray &ray, float t_max, DevPCG32 &rng) {
    DMediumSample out;
    auto sigma_a = mk3(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]), sigma_s = mk3(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]);
    auto sigma_t = sigma_a + sigma_s;
    auto pdf_channels = vpt_channel_pdf(rng);
    auto channel = vpt_sample_discrete3(pdf_channels, rng.uniform_float());
    auto u = rng.uniform_float();
    auto t = -logf(fmaxf(1.f - u, 1.17549435e-38f)) / comp(sigma_t, channel);
    if (t > t_max) {// hit surface
        out.event = kMediumHitSurface;
        auto Tr = exp3(sigma_t * (-t_max));
        out.ray.o = ray.o + ray.d * t_max, out.ray.d = ray.d, out.ray.t_min = 0.f, out.ray.t_max = kFloatMax;
        auto pdf = pdf_channels * Tr;
        out.f = Tr, out.pdf = pdf.x + pdf.y + pdf.z;
    } else {
        auto p_absorb = comp(sigma_a, channel) / comp(sigma_t, channel), p_scatter = comp(sigma_s, channel) / comp(sigma_t, channel);
        auto absorb = rng.uniform_float() * (p_absorb + p_scatter) <= p_absorb;// sample_discrete(float2), sampling.cpp:162-166
        if (absorb) {
            out.event = kMediumAbsorb;
            out.ray = ray;
            out.f = mk3(0.f);
            auto pdf = pdf_channels * sigma_t;
            out.pdf = pdf.x + pdf.y + pdf.z;
        } else {// scatter: the direction is built around the WORLD y axis, not around wo (henyey_greenstein.cpp:28-45; kept)
            out.event = kMediumScatter;
            auto Tr = exp3(sigma_t * (-t));
            auto ux = rng.uniform_float(), uy = rng.uniform_float();
            auto g = m.g;
            auto cos_theta = fabsf(g) < 1e-3f ? 1.f - 2.f * ux : -1.f / (2.f * g) * (1.f + g * g - sqr((1.f - g * g) / (1.f + g - 2.f * g * ux)));
            auto sin_theta = sqrtf(fmaxf(0.f, 1.f - cos_theta * cos_theta));
            auto phi = 2.f * kPi * uy;
            out.ray.o = ray.o + ray.d * t, out.ray.d = mk3(sin_theta * cosf(phi), cos_theta, sin_theta * sinf(phi));
            out.ray.t_min = 0.f, out.ray.t_max = kFloatMax;
            auto pdf = pdf_channels * (sigma_t * Tr);
            out.f = Tr * sigma_s, out.pdf = pdf.x + pdf.y + pdf.z;
        }
    }
    return out;
}

// closure of the surface at `it` seen from `wo` (Surface::Instance::closure incl. NormalMap wrapper) — evaluate / sample
// for any closure kind: the five basic ones inline, Disney / Mix / Layered through the out-of-line heavy path
struct VptClosure {
    DClosure closure;
    Frame sh;
    HeavyCtx heavy;
    bool is_heavy;
};
LR_D void vpt_closure(const DScene &scene, const SurfacePoint &it, f3 wo, float eta_i, VptClosure &c) {
    const LobeTables tables{scene.closures, scene.surfaces, scene.textures, scene.texels};
    load_lobe(tables, it.uv, it.ng, wo, (it.tags >> 12u) & 4095u, it.shading, c.closure, c.sh, eta_i);
    c.is_heavy = c.closure.kind >= LR_SURFACE_DISNEY;
    if (c.is_heavy) {
        c.heavy.tb = tables, c.heavy.uv = it.uv, c.heavy.ng = it.ng, c.heavy.p = it.p, c.heavy.wo = wo;
        c.heavy.shading = c.sh, c.heavy.closure = c.closure;
    }
}
LR_D BsdfEval vpt_evaluate(const VptClosure &c, f3 ng, f3 wo, f3 wi) {
    return c.is_heavy ? heavy_evaluate<true, true>(&c.heavy, wi) : closure_evaluate<false>(c.closure, c.sh, ng, wo, wi);
}

enum : uint32_t { kVptBegin = 0u, kVptMain, kVptWalkMedium, kVptWalkSurface, kVptSurface, kVptShade, kVptEnd, kVptDone };
constexpr uint32_t kVptMaxCrossings = 64u;

template<uint32_t F>
__global__ __launch_bounds__(kBlockThreads, 2) void megavpt_kernel(DScenePtr scene_ptr, RenderArgs args) {
    const DScene &scene = *(const DScene *)scene_ptr;
    constexpr bool COUNT = (F & 1u) != 0u, PCG = (F & 2u) != 0u;
    __shared__ uint32_t s_stack[kStackLds * kBlockThreads];
    __shared__ float4 s_stage[kWavesPerBlock * kStageWave];
    __shared__ float4 s_film[kWavesPerBlock * 64u];
    const auto tid = threadIdx.x;
    const auto lane = tid & 63u;
    const auto gtid = blockIdx.x * kBlockThreads + tid;
    TraversalStack stack{s_stack + tid, args.spill + gtid, args.total_threads, s_stage + __builtin_amdgcn_readfirstlane(tid >> 6u) * kStageWave};
    const auto film_tile = s_film + (tid >> 6u) * 64u;
    DCounters local{};

    for (;;) {
        uint32_t item = 0u;
        if (lane == 0u) { item = atomicAdd(args.work_counter, 1u); }
        item = __shfl(item, 0);
        if (item >= args.item_count) { break; }
        const auto range = item_range(args, item);
        const auto tile_index = range.tile_index, chunk = range.chunk;
        const auto tile = args.tile_begin + tile_index * args.tile_stride;
        const auto ty = tile / args.tiles_x, tx = (tile - ty * args.tiles_x + ty) % args.tiles_x;// row ty is rotated by ty (lrhip.h)
        const auto s_begin = range.s_begin, s_end = range.s_end;
        film_tile[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        const auto px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3u);
        const auto inside = px < scene.camera.width && py < scene.camera.height;

        for (auto s = s_begin; s < s_end; s++) {
            // ---- per-lane path state (mega_vpt_naive.cpp:170-245)
            PathSampler<PCG> sampler{};
            DevPCG32 rng{};
            DMediumTracker tracker, walk_tracker;
            TravState tr{};
            tr.phase = kPhaseIdle;
            Ray ray{};
            f3 beta = mk3(0.f), Li = mk3(0.f);
            auto pdf_bsdf = 1e16f, eta_scale = 1.f, eta = 1.f, u_rr = 0.f;
            auto depth = 0u;
            auto state = static_cast<uint32_t>(kVptDone);
            // main hit of this depth
            SurfacePoint it{};
            auto it_valid = false, has_medium = false;
            auto it_prim = 0u;
            auto t_max = kFloatMax;
            auto medium_event = static_cast<uint32_t>(kMediumInvalid);
            // transmittance walk
            f3 walk_f = mk3(1.f), light_p = mk3(0.f), walk_dir = mk3(0.f);
            auto walk_pdf = 0.f;
            auto crossings = 0u;
            Ray walk_ray{};
            LightPick pick{};
            auto u_lobe = 0.f;
            f2 u_bsdf{0.f, 0.f};
            tracker.init();
            if (inside) {
                sampler.start(scene, px, py, s);
                auto u_filter = sampler.next_pixel_2d();
                auto u_lens = scene.camera.kind == LR_CAMERA_THIN_LENS ? sampler.next_2d() : f2{.5f, .5f};
                float weight;
                camera_ray(scene, scene.filter, px, py, u_filter, u_lens, ray, weight);
                beta = mk3(weight);
                auto u_rng = sampler.next_2d();// PCG32 rng(U64(as<UInt2>(generate_2d()))): x -> high word (u64.h:48-51)
                rng.set_sequence((static_cast<uint64_t>(__float_as_uint(u_rng.x)) << 32u) | __float_as_uint(u_rng.y));
                if (scene.env_medium_tag != LR_INVALID_ID) { tracker.enter(scene.media[scene.env_medium_tag].priority, scene.env_medium_tag); }
                state = kVptBegin;
                if (COUNT) { local.paths++; }
            }

            auto request = [&](const Ray &r) {
                tr.hit.inst = kInvalid, tr.hit.prim = kInvalid, tr.hit.u = 0.f, tr.hit.v = 0.f;
                tr.occluded = false;
                trav_begin(tr, r, kPhaseClosest);
                if (COUNT) { local.closest_rays++; }
            };
            auto start_walk = [&](uint32_t walk_state) {// _transmittance prologue, :95-106
                walk_ray = pick.shadow;
                walk_dir = pick.shadow.d;
                light_p = pick.shadow.o + pick.shadow.d * pick.shadow.t_max;
                walk_f = mk3(1.f), walk_pdf = 0.f, crossings = 0u;
                walk_tracker = tracker;
                state = walk_state;
                request(walk_ray);
            };

            for (;;) {
                // ==== advance every lane that has no ray in flight until it needs a trace or its path ends
                if (tr.phase == kPhaseIdle) {
                    while (state != kVptDone) {
                        if (state == kVptBegin) {// top of the depth loop, :247-254
                            if (depth >= scene.max_depth) { state = kVptDone; break; }
                            eta = 1.f;
                            u_rr = depth + 1u >= scene.rr_depth ? sampler.next_1d() : 0.f;
                            state = kVptMain;
                            request(ray);
                            break;
                        }
                        if (state == kVptMain) {// main hit known, :254-316
                            it_valid = tr.hit.inst != kInvalid;
                            has_medium = false;
                            t_max = kFloatMax;
                            if (it_valid) {
                                it_prim = tr.hit.prim;
                                reconstruct<true>(scene, tr.hit.inst, tr.hit.prim, mk3(1.f - tr.hit.u - tr.hit.v, tr.hit.u, tr.hit.v), it);
                                it.back_facing = dot(-ray.d, it.ng) < 0.0f;
                                has_medium = (it.flags & LR_SHAPE_HAS_MEDIUM) != 0u;
                                t_max = length(it.p - ray.o);
                                if (COUNT) { local.surface_hits++; }
                            }
                            medium_event = kMediumInvalid;
                            if (!tracker.vacuum()) {// direct lighting of the medium point at the ray origin, :283-299
                                auto u_light_selection = sampler.next_1d();
                                auto u_light_surface = sampler.next_2d();
                                SurfacePoint mp{};// Interaction{pg}: ng = pg (interaction.h:77-78), identity frame, null shape
                                mp.p = ray.o, mp.ng = ray.o;
                                mp.shading.s = mk3(1.f, 0.f, 0.f), mp.shading.t = mk3(0.f, 1.f, 0.f), mp.shading.n = mk3(0.f, 0.f, 1.f);
                                mp.offset_bits = 0u;
                                if (COUNT) { local.nee_samples++; }
                                pick = sample_one_light<true, true>(scene, mp, u_light_selection, u_light_surface);
                                start_walk(kVptWalkMedium);
                                break;
                            }
                            state = kVptSurface;
                            continue;
                        }
                        if (state == kVptWalkMedium || state == kVptWalkSurface) {// one crossing of _transmittance, :108-165
                            auto walking = false;
                            if (tr.hit.inst != kInvalid) {
                                SurfacePoint wit;
                                reconstruct<true>(scene, tr.hit.inst, tr.hit.prim, mk3(1.f - tr.hit.u - tr.hit.v, tr.hit.u, tr.hit.v), wit);
                                wit.back_facing = dot(-walk_ray.d, wit.ng) < 0.0f;
                                if (COUNT) { local.surface_hits++; }
                                auto wo = -walk_dir, wi = walk_dir;
                                auto t2surface = length(wit.p - walk_ray.o);
                                auto w_has_medium = (wit.flags & LR_SHAPE_HAS_MEDIUM) != 0u;
                                auto w_has_surface = (wit.flags & LR_SHAPE_HAS_SURFACE) != 0u;
                                auto medium_tag = wit.tags >> 24u;
                                VptClosure wc;
                                auto frame = wit.shading;
                                if (w_has_surface) {
                                    vpt_closure(scene, wit, wo, 1.f, wc);
                                    frame = wc.sh;
                                }
                                auto wo_l = to_local(frame, wo), wi_l = to_local(frame, wi);// _event, :68-93
                                auto event = wo_l.z * wi_l.z > 0.f ? kEventReflect : (wi_l.z > 0.f ? kEventExit : kEventEnter);
                                if (!walk_tracker.vacuum()) {// HomogeneousMediumClosure::transmittance, homogeneous.cpp:124-137
                                    auto &m = scene.media[walk_tracker.current_tag()];
                                    auto pc = vpt_channel_pdf(rng);
                                    auto Tr = exp3(mk3(m.sigma_a[0] + m.sigma_s[0], m.sigma_a[1] + m.sigma_s[1], m.sigma_a[2] + m.sigma_s[2]) * (-t2surface));
                                    auto pdf = pc * Tr;
                                    walk_f = walk_f * Tr, walk_pdf += pdf.x + pdf.y + pdf.z;
                                }
                                if (w_has_medium) {
                                    auto priority = scene.media[medium_tag].priority;
                                    if (event == kEventExit) { walk_tracker.exit(priority, medium_tag); }
                                    else { walk_tracker.enter(priority, medium_tag); }
                                }
                                if (w_has_surface) {
                                    auto e = vpt_evaluate(wc, wit.ng, wo, wi);
                                    walk_f = walk_f * e.f, walk_pdf += e.pdf;
                                }
                                auto p_from = robust_origin(wit, light_p - wit.p);// spawn_ray_to, interaction.cpp:25-30
                                auto Lv = light_p - p_from;
                                auto dist = length(Lv);
                                walk_ray.o = p_from, walk_ray.d = Lv * (1.f / dist), walk_ray.t_min = 0.f, walk_ray.t_max = dist * .9999f;
                                crossings++;
                                walking = (walk_f.x > 0.f || walk_f.y > 0.f || walk_f.z > 0.f) && crossings < kVptMaxCrossings;
                            }
                            if (walking) {
                                request(walk_ray);
                                break;
                            }
                            if (state == kVptWalkMedium) {// :293-313
                                if (walk_pdf > 0.f) {
                                    auto w = 1.f / (pdf_bsdf + walk_pdf + pick.pdf);
                                    Li += w * beta * walk_f * pick.L;
                                }
                                auto &medium = scene.media[tracker.current_tag()];
                                eta = medium.eta;
                                if (medium.kind != LR_MEDIUM_VACUUM) {
                                    auto ms = vpt_medium_sample(medium, ray, t_max, rng);
                                    ray = ms.ray;
                                    medium_event = ms.event;
                                    beta *= ms.f * (ms.pdf > 0.f ? 1.f / ms.pdf : 0.f);
                                    pdf_bsdf = ms.pdf;
                                }
                                state = kVptSurface;
                            } else {
                                state = kVptShade;
                            }
                            continue;
                        }
                        if (state == kVptSurface) {// :318-361
                            if (medium_event != kMediumInvalid && medium_event != kMediumHitSurface) { state = kVptEnd; continue; }
                            if (!it_valid) {
                                if (scene.env_kind != kEnvNone) {
                                    f3 L = mk3(scene.env_L[0], scene.env_L[1], scene.env_L[2]);
                                    auto pdf = kInvPi * 0.25f;
                                    if (scene.env_kind != kEnvConstant) { env_evaluate(scene, ray.d, L, pdf); }
                                    Li += beta * L * balance(pdf_bsdf, pdf * scene.env_prob);
                                }
                                state = kVptDone;
                                break;
                            }
                            if (scene.has_lights && (it.flags & LR_SHAPE_HAS_LIGHT)) {
                                f3 L;
                                float pdf;
                                light_evaluate(scene, it, it_prim, ray.o, L, pdf);
                                pdf *= (1.f - scene.env_prob) / static_cast<float>(scene.light_count);
                                Li += beta * L * balance(pdf_bsdf, pdf);
                            }
                            if (!(it.flags & LR_SHAPE_HAS_SURFACE)) { state = kVptDone; break; }
                            if (COUNT) { local.path_length_sum++, local.nee_samples++; }
                            auto u_light_selection = sampler.next_1d();
                            auto u_light_surface = sampler.next_2d();
                            u_lobe = sampler.next_1d();
                            u_bsdf = sampler.next_2d();
                            pick = sample_one_light<true>(scene, it, u_light_selection, u_light_surface);
                            start_walk(kVptWalkSurface);
                            break;
                        }
                        if (state == kVptShade) {// :363-447
                            auto medium_tag = it.tags >> 24u;
                            auto medium_priority = LR_MEDIUM_VACUUM_PRIORITY;
                            auto eta_next = 1.f;
                            if (has_medium) { medium_priority = scene.media[medium_tag].priority, eta_next = scene.media[medium_tag].eta; }
                            auto wo = -ray.d;
                            VptClosure c;
                            vpt_closure(scene, it, wo, 1.f, c);// _event's closure (eta_i = 1), :68-80
                            auto wo_l = to_local(c.sh, wo), wi_l = to_local(c.sh, ray.d);
                            auto event_skip = wo_l.z * wi_l.z > 0.f ? kEventReflect : (wi_l.z > 0.f ? kEventExit : kEventEnter);
                            if (eta != 1.f) { vpt_closure(scene, it, wo, eta, c); }// surface->closure(call, *it, swl, wo, eta, time), :381
                            uint32_t event;
                            if (!tracker.true_hit(medium_tag)) {// (the TAG is passed where a priority is expected, :384; kept)
                                event = event_skip;
                                auto d = ray.d;
                                ray.o = robust_origin(it, d), ray.d = d, ray.t_min = 0.f, ray.t_max = kFloatMax;
                                pdf_bsdf = 1e16f;
                            } else {
                                if (pick.pdf > 0.0f) {
                                    auto eval = vpt_evaluate(c, it.ng, wo, pick.shadow.d);
                                    auto w = 1.f / (pick.pdf + eval.pdf + walk_pdf);
                                    Li += w * beta * eval.f * pick.L * walk_f;
                                }
                                BsdfSample bs;
                                if (c.is_heavy) { bs = heavy_sample<true, true>(&c.heavy, u_lobe, u_bsdf).bs; }
                                else { bs = closure_sample<false>(c.closure, c.sh, it.ng, wo, u_lobe, u_bsdf); }
                                event = bs.event;
                                pdf_bsdf = bs.pdf;
                                ray.o = robust_origin(it, bs.wi), ray.d = bs.wi, ray.t_min = 0.f, ray.t_max = kFloatMax;
                                beta *= (bs.pdf > 0.f ? 1.f / bs.pdf : 0.f) * bs.f;
                                if (has_medium) {
                                    if (event == kEventEnter) { eta_scale = sqr(eta_next / eta); }
                                    else if (event == kEventExit) { eta_scale = sqr(eta / eta_next); }
                                }
                            }
                            if (has_medium) {
                                if (event == kEventEnter) { tracker.enter(medium_priority, medium_tag); }
                                else if (event == kEventExit) { tracker.exit(medium_priority, medium_tag); }
                            }
                            state = kVptEnd;
                            continue;
                        }
                        if (state == kVptEnd) {// :469-477
                            if (any_nan(beta)) { beta = mk3(0.f); }
                            if (beta.x <= 0.f && beta.y <= 0.f && beta.z <= 0.f) { state = kVptDone; break; }
                            auto q = fmaxf(max_component(beta) * eta_scale, .05f);
                            if (depth + 1u >= scene.rr_depth) {
                                if (q < scene.rr_threshold && u_rr >= q) { state = kVptDone; break; }
                                beta *= q < scene.rr_threshold ? 1.0f / q : 1.f;
                            }
                            depth++;
                            state = kVptBegin;
                            continue;
                        }
                    }
                }
                if (!lr_any(tr.phase != kPhaseIdle)) { break; }
                // ==== trace every pending ray of the wave to completion
                TraceStats ts{0u, 0u, 0u, 0u, 0u, 0u};
                trace_until_refill<COUNT, true>(scene, stack, tr, false, ray, 65, ts);
                if (COUNT) {
                    local.nodes_visited += ts.nodes, local.tris_tested += ts.tris;
                    local.trace_steps += ts.steps, local.trace_steps_busy += ts.steps_busy;
                }
            }
            if (inside) { film_accumulate(film_tile + lane, Li * scene.shutter_weight, scene.film_clamp); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (inside) {
            const auto acc = film_tile[lane];
            const auto index = py * scene.camera.width + px;
            if (args.chunk_count == 1u) {
                auto f = args.film[index];
                f.x += acc.x, f.y += acc.y, f.z += acc.z, f.w += acc.w;
                args.film[index] = f;
            } else {
                args.partial[static_cast<size_t>(chunk) * scene.camera.width * scene.camera.height + index] = acc;
            }
        }
    }
    if (COUNT) {
        auto reduce = [&](unsigned long long v, unsigned long long *dst) {
            for (auto off = 32; off > 0; off >>= 1) { v += __shfl_down(v, off); }
            if (lane == 0u) { atomicAdd(dst, v); }
        };
        reduce(local.paths, &args.counters->paths);
        reduce(local.closest_rays, &args.counters->closest_rays);
        reduce(local.nodes_visited, &args.counters->nodes_visited);
        reduce(local.tris_tested, &args.counters->tris_tested);
        reduce(local.surface_hits, &args.counters->surface_hits);
        reduce(local.nee_samples, &args.counters->nee_samples);
        reduce(local.path_length_sum, &args.counters->path_length_sum);
        reduce(local.trace_steps, &args.counters->trace_steps);
        reduce(local.trace_steps_busy, &args.counters->trace_steps_busy);
    }
}

}// namespace lrd
