// dev_layered.h — LayeredSurfaceClosure (src/surfaces/layered.cpp:195-470, itself a port of PBRT-v4's
// LayeredBxDF onto nested closures) for the FULL kernel variants.
//
// The two interfaces are ordinary closure records (`DClosure` + shading frame, resolved by the kernel's
// load_lobe like Mix children); the medium between them is (thickness, g, albedo).  evaluate() is a
// stochastic estimator: a random walk between the interfaces driven by an LCG that is seeded with a
// hash of the hit position and direction BITS (layered.cpp:271,416) — so, like the alpha test, parity
// with the CPU oracle is statistical (fp contraction changes those bits), never bit-level.
// Where the reference evaluates several lcg(seed) calls in one argument list (C++ leaves their order
// unspecified) they are taken left to right, as in the oracle.
#pragma once
#include "dev_shade.h"

namespace lrd {

LR_HD uint32_t xxhash32_3(uint32_t x, uint32_t y, uint32_t z) {// rng.cpp:38-51
    constexpr uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    auto rot = [](uint32_t h) { return (h << 17u) | (h >> 15u); };
    auto h = z + P5 + x * P3;
    h = P4 * rot(h);
    h += y * P3;
    h = P4 * rot(h);
    h = P2 * (h ^ (h >> 15u));
    h = P3 * (h ^ (h >> 13u));
    return h ^ (h >> 16u);
}

#ifndef LR_NEST// free Mix / Layered composition: only in the variants that carry kFeatNest (512), one translation unit per variant
#if defined(LR_VARIANT) && ((LR_VARIANT) & 512) && !((LR_VARIANT) & 256)
#define LR_NEST 1
#else
#define LR_NEST 0
#endif
#endif

// Layered surfaces on a path through the interfaces (lr_scene.h: LR_LAYERED_MAX_LEVELS; a Layered surface as an interface of a Layered
// surface needs the free-composition variants).  The functions below are templated on LV, the Layered levels still allowed INSIDE the
// interfaces of the stack they work on: device code has no recursion, so each level is its own instantiation.
#ifndef LR_LAYER_LEVELS
#define LR_LAYER_LEVELS (LR_NEST ? LR_LAYERED_MAX_LEVELS : 1)
#endif
constexpr int kLayerLevels = LR_LAYER_LEVELS;
static_assert(LR_LAYER_LEVELS >= 1 && LR_LAYER_LEVELS <= 2, "dev_heavy.h: layer_inner_stack builds innermost stacks only");

struct LobeTables {// the scene tables closure loading reads (dev_heavy.h: load_lobe)
    const DClosure *closures;
    const DSurface *surfaces;
    const lr_texture *textures;
    const float *texels;
};

struct LayerStack {
    DClosure top, bottom;
    Frame f_top, f_bottom;// the children's own (possibly normal-mapped) shading frames
    Frame own;            // the Layered surface's frame
    f3 ng, p;
    float thickness, g;
    f3 albedo;
    uint32_t max_depth, samples;
#if LR_NEST
    // an interface may be a Mix tree (round 2): its children are loaded per call, which needs what populate_closure had
    LobeTables tb;
    f2 uv;
    f3 wo_pop;              // the hit's wo the closures were populated with (clamp_shading_normal of normal-mapped children)
    float eta_i, eta_bottom;// eta_i the top / the bottom (and their children) were populated with, layered.cpp:497-499
#endif
};
template<int LV>
LR_HEAVY BsdfEval layered_evaluate(const LayerStack &L, f3 wo, f3 wi, bool importance);
template<int LV>
LR_HEAVY BsdfSample layered_sample(const LayerStack &L, f3 wo, float u_lobe, f2 u, bool importance);
#if LR_NEST
// Mix interfaces (dev_heavy.h: the Mix interpreter; its leaves may be Layered surfaces again while LV > 0) and Layered interfaces
// (layer_inner_stack: populate_closure of the interface's own record, layered.cpp:478-500)
template<int LV>
__device__ BsdfEval layer_mix_evaluate(const LayerStack &L, bool is_top, f3 wo, f3 wi, bool importance);
template<int LV>
__device__ BsdfSample layer_mix_sample(const LayerStack &L, bool is_top, f3 wo, float uc, f2 u, bool importance);
__device__ void layer_inner_stack(const LayerStack &L, bool is_top, LayerStack &inner);
#endif

struct LayerRng {
    uint32_t state;
    LR_D float next() {// lcg, rng.cpp:132-140
        state = 1664525u * state + 1013904223u;
        return uint_to_unit_float(state);
    }
};

LR_D float layer_tr(float dz, f3 w) {// :214-217
    return fabsf(dz) <= 1.17549435e-38f ? 1.f : expf(-fabsf(dz / w.z));
}
LR_D float hg_phase(float cos_t, float g) {// HGPhaseFunction::HenyeyGreenstein, :21-24
    auto denom = 1.f + sqr(g) + 2.f * g * cos_t;
    return kInvPi / 4.0f * (1.f - sqr(g)) / (denom * sqrtf(denom));
}
LR_D f3 hg_sample(f3 wo, float g, f2 u, float &pdf) {// :25-38
    auto cos_t = fabsf(g) < 1e-3f ? 1.f - 2.f * u.x : -1.f / (2.f * g) * (1.f + sqr(g) - sqr((1.f - sqr(g)) / (1.f + g - 2.f * g * u.x)));
    auto sin_t = sqrtf(1.f - sqr(cos_t));
    float sn, cs;
    sincos_2pi(u.y, sn, cs);// phi = 2 pi u.y
    auto wi = to_world(frame_from_normal(wo), mk3(sin_t * cs, sin_t * sn, cos_t));
    pdf = hg_phase(cos_t, g);
    return wi;
}
LR_D float power_heuristic(float f, float g) {// sampling.cpp:142-159
    auto ff = f * f, gg = g * g;
    auto sum = ff + gg;
    return isinf(ff) ? 1.f : (sum == 0.f ? 0.f : ff / sum);
}
LR_D bool is_black(f3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }

// TopOrBottom (layered.cpp:54-102): `is_top` picks the interface
// (not inlined: the random walk has ~20 call sites and each inlined copy would carry the whole closure interpreter)
template<int LV>
__device__ __noinline__ BsdfEval layer_eval(const LayerStack &L, bool is_top, f3 wo, f3 wi, bool importance) {
#if LR_NEST
    const auto kind = is_top ? L.top.kind : L.bottom.kind;
    if (kind == LR_SURFACE_MIX) { return layer_mix_evaluate<LV>(L, is_top, wo, wi, importance); }
    if constexpr (LV > 0) {
        if (kind == LR_SURFACE_LAYERED) {// a Layered interface: the walk of the inner stack, in the outer walk's transport mode
            LayerStack inner;
            layer_inner_stack(L, is_top, inner);
            return layered_evaluate<LV - 1>(inner, wo, wi, importance);
        }
    }
#endif
    return is_top ? closure_evaluate<true>(L.top, L.f_top, L.ng, wo, wi, importance) :
                    closure_evaluate<true>(L.bottom, L.f_bottom, L.ng, wo, wi, importance);
}
template<int LV>
__device__ __noinline__ BsdfSample layer_sample(const LayerStack &L, bool is_top, f3 wo, float uc, f2 u, bool importance) {
#if LR_NEST
    const auto kind = is_top ? L.top.kind : L.bottom.kind;
    if (kind == LR_SURFACE_MIX) { return layer_mix_sample<LV>(L, is_top, wo, uc, u, importance); }
    if constexpr (LV > 0) {
        if (kind == LR_SURFACE_LAYERED) {
            LayerStack inner;
            layer_inner_stack(L, is_top, inner);
            return layered_sample<LV - 1>(inner, wo, uc, u, importance);
        }
    }
#endif
    return is_top ? closure_sample<true>(L.top, L.f_top, L.ng, wo, uc, u, importance) :
                    closure_sample<true>(L.bottom, L.f_bottom, L.ng, wo, uc, u, importance);
}
LR_D f3 layer_to_local(const LayerStack &L, bool is_top, f3 w) { return to_local(is_top ? L.f_top : L.f_bottom, w); }
LR_D f3 layer_to_world(const LayerStack &L, bool is_top, f3 w) { return to_world(is_top ? L.f_top : L.f_bottom, w); }

// LayeredSurfaceClosure::_evaluate, :256-398 (+ the public wrapper's side validation, surface.cpp:45-56).  importance: the transport
// mode of this evaluation -- RADIANCE for a surface hit by a path, either for a Layered interface of an outer walk
template<int LV>
LR_HEAVY BsdfEval layered_evaluate(const LayerStack &L, f3 wo, f3 wi, bool importance) {
    const auto mode = importance, reverse_mode = !importance;// :286
    auto samples = static_cast<float>(L.samples);
    auto wi_local = to_local(L.own, wi), wo_local = to_local(L.own, wo);
    auto entered_top = wo_local.z > 0.f;
    auto sh = same_hemisphere(wo_local, wi_local);
    auto enter_top = entered_top;         // enter interface
    auto exit_top = !(sh != entered_top); // TopOrBottom(bottom, top, sh ^ entered_top): flag -> bottom
    auto nonexit_top = sh != entered_top;
    auto exit_z = (sh != entered_top) ? 0.f : L.thickness;
    auto first = layer_eval<LV>(L, enter_top, wo, wi, mode);
    auto f = sh ? samples * first.f : mk3(0.f);
    auto pdf_sum = sh ? samples * first.pdf : 0.f;
    LayerRng rng{xxhash32_4(__float_as_uint(L.p.x), __float_as_uint(L.p.y), __float_as_uint(L.p.z),
                            xxhash32_3(__float_as_uint(wi.x), __float_as_uint(wi.y), __float_as_uint(wi.z)))};
    auto draw3 = [&](float &uc, f2 &u) { uc = rng.next(), u.x = rng.next(), u.y = rng.next(); };
    for (auto i = 0u; i < L.samples; i++) {
        float uc;
        f2 u;
        draw3(uc, u);
        auto wos = layer_sample<LV>(L, enter_top, wo, uc, u, mode);
        if (is_black(wos.f) || wos.pdf <= 0.f) { continue; }
        draw3(uc, u);
        auto wis = layer_sample<LV>(L, exit_top, wi, uc, u, reverse_mode);
        auto wis_wi_local = layer_to_local(L, exit_top, wis.wi);
        if (is_black(wis.f) || wis.pdf <= 0.f) { continue; }
        auto beta = wos.f * (1.f / wos.pdf);
        auto z = entered_top ? L.thickness : 0.f;
        auto w = wos.wi;
        auto w_local = layer_to_local(L, enter_top, w);
        for (auto depth = 0u; depth < L.max_depth; depth++) {
            if (depth > 3u && max_component(beta) < 0.25f) {
                auto q = fmaxf(0.f, 1.f - max_component(beta));
                if (rng.next() < q) { break; }
                beta = beta * (1.f / (1.f - q));
            }
            if (is_black(L.albedo)) {
                z = z == L.thickness ? 0.f : L.thickness;
                beta = beta * layer_tr(L.thickness, w_local);
            } else {
                auto dz = -logf(1.f - rng.next()) / (1.f / fabsf(w_local.z));
                auto zp = w_local.z > 0.f ? z + dz : z - dz;
                if (z == zp) { continue; }
                if (zp > 0.f && zp < L.thickness) {
                    auto wt = power_heuristic(wis.pdf, layer_eval<LV>(L, nonexit_top, -w, -wis.wi, mode).pdf);
                    f += beta * L.albedo * (hg_phase(dot(-w_local, -wis_wi_local), L.g) * wt * layer_tr(zp - exit_z, wis_wi_local)) * wis.f * (1.f / wis.pdf);
                    f2 up;
                    up.x = rng.next(), up.y = rng.next();
                    float ps_pdf;
                    auto ps_wi = hg_sample(-w_local, L.g, up, ps_pdf);
                    if (ps_pdf <= 0.f || ps_wi.z == 0.f) { continue; }
                    beta = beta * L.albedo;// * ps.p / ps.pdf = 1
                    w_local = ps_wi;
                    w = layer_to_world(L, exit_top, w_local);
                    z = zp;
                    if ((z < exit_z && w_local.z > 0.f) || (z > exit_z && w_local.z < 0.f)) {
                        auto e = layer_eval<LV>(L, exit_top, -w, wi, mode);
                        if (!is_black(e.f)) { f += beta * e.f * (layer_tr(zp - exit_z, w_local) * power_heuristic(ps_pdf, e.pdf)); }
                    }
                    continue;
                }
                z = clampf(zp, 0.f, L.thickness);
            }
            if (z == exit_z) {
                draw3(uc, u);
                auto bs = layer_sample<LV>(L, exit_top, -w, uc, u, mode);
                if (is_black(bs.f) || bs.pdf <= 0.f) { break; }
                beta = beta * bs.f * (1.f / bs.pdf);
                w = bs.wi;
                w_local = layer_to_local(L, exit_top, w);
            } else {
                auto wns = layer_eval<LV>(L, nonexit_top, -w, -wis.wi, mode);
                auto wt = power_heuristic(wis.pdf, wns.pdf);
                f += beta * wns.f * (wt * layer_tr(L.thickness, wis_wi_local)) * wis.f * (1.f / wis.pdf);
                draw3(uc, u);
                auto bs = layer_sample<LV>(L, nonexit_top, -w, uc, u, mode);
                if (is_black(bs.f) || bs.pdf <= 0.f) { break; }
                beta = beta * bs.f * (1.f / bs.pdf);
                w = bs.wi;
                w_local = layer_to_local(L, nonexit_top, w);
                auto wes = layer_eval<LV>(L, exit_top, -w, wi, mode);
                if (!is_black(wes.f)) { f += beta * wes.f * (layer_tr(L.thickness, w_local) * power_heuristic(bs.pdf, wes.pdf)); }
            }
        }
    }
    for (auto i = 0u; i < L.samples; i++) {// pdf estimate, :360-395
        float uc;
        f2 u;
        if (sh) {
            auto r_top = !entered_top, t_top = entered_top;
            draw3(uc, u);
            auto wos = layer_sample<LV>(L, t_top, wo, uc, u, mode);
            draw3(uc, u);
            auto wis = layer_sample<LV>(L, t_top, wi, uc, u, reverse_mode);
            if (!is_black(wos.f) && wos.pdf > 0.f && !is_black(wis.f) && wis.pdf > 0.f) {
                draw3(uc, u);
                auto rs = layer_sample<LV>(L, r_top, -wos.wi, uc, u, mode);
                if (!is_black(rs.f) && rs.pdf > 0.f) {
                    auto r_pdf = layer_eval<LV>(L, r_top, -wos.wi, -wis.wi, mode).pdf;
                    pdf_sum += power_heuristic(wis.pdf, r_pdf) * r_pdf;
                    auto t_pdf = layer_eval<LV>(L, t_top, -rs.wi, wi, mode).pdf;
                    pdf_sum += power_heuristic(rs.pdf, t_pdf) * t_pdf;
                }
            }
        } else {
            auto ti_top = !entered_top, to_top = entered_top;
            draw3(uc, u);
            auto wos = layer_sample<LV>(L, to_top, wo, uc, u, mode);
            draw3(uc, u);
            auto wis = layer_sample<LV>(L, ti_top, wi, uc, u, reverse_mode);
            if (is_black(wos.f) || wos.pdf <= 0.f || is_black(wis.f) || wis.pdf <= 0.f) { continue; }
            pdf_sum += .5f * (layer_eval<LV>(L, to_top, wo, -wis.wi, mode).pdf + layer_eval<LV>(L, ti_top, -wos.wi, wi, mode).pdf);
        }
    }
    BsdfEval out{f * (1.f / samples), lerp(1.f / (4.f * kPi), pdf_sum / samples, 0.9f)};
    if (!valid_sides(L.ng, L.own.n, wo, wi)) { out.f = mk3(0.f), out.pdf = 0.f; }
    return out;
}

// LayeredSurfaceClosure::_sample, :399-470
template<int LV>
LR_HEAVY BsdfSample layered_sample(const LayerStack &L, f3 wo, float u_lobe, f2 u, bool importance) {
    const auto mode = importance;
    auto wo_local = to_local(L.own, wo);
    auto entered_top = wo_local.z > 0.f;
    auto bs = layer_sample<LV>(L, entered_top, wo, u_lobe, u, mode);
    BsdfSample s{mk3(0.f), 0.f, mk3(0.f, 0.f, 1.f), kEventReflect};
    if (!is_black(bs.f) && bs.pdf != 0.f) {
        auto wi_local = to_local(L.own, bs.wi);
        if (same_hemisphere(wi_local, wo_local)) {
            s = bs;
        } else {
            auto w = bs.wi;
            auto w_local = wi_local;
            LayerRng rng{xxhash32_4(__float_as_uint(u.x), __float_as_uint(u.y), __float_as_uint(u_lobe),
                                    xxhash32_3(__float_as_uint(wo.x), __float_as_uint(wo.y), __float_as_uint(wo.z)))};
            auto f = bs.f;
            auto pdf = bs.pdf;
            auto z = entered_top ? L.thickness : 0.f;
            for (auto depth = 0u; depth < L.max_depth; depth++) {
                auto rr_beta = max_component(f) / pdf;
                if (depth > 3u && rr_beta < 0.25f) {
                    auto q = fmaxf(0.f, 1.f - rr_beta);
                    if (rng.next() < q) { break; }
                    pdf *= 1.f - q;
                }
                if (w_local.z == 0.f) { break; }
                if (!is_black(L.albedo)) {
                    auto dz = -logf(1.f - rng.next()) / (1.f / fabsf(w_local.z));
                    auto zp = w_local.z > 0.f ? z + dz : z - dz;
                    if (z == zp) { break; }
                    if (0.f < zp && zp < L.thickness) {
                        f2 up;
                        up.x = rng.next(), up.y = rng.next();
                        float ps_pdf;
                        auto ps_wi = hg_sample(-w_local, L.g, up, ps_pdf);
                        if (ps_pdf <= 0.f) { break; }
                        f = f * L.albedo * ps_pdf;
                        pdf *= ps_pdf;
                        w = ps_wi;// (the reference assigns the LOCAL phase sample to the world-space w: kept)
                        w_local = to_local(L.own, w);
                        z = zp;
                        continue;
                    }
                    z = clampf(zp, 0.f, L.thickness);
                } else {
                    z = z == L.thickness ? 0.f : L.thickness;
                    f = f * layer_tr(L.thickness, w_local);
                }
                auto interface_top = !(z == 0.f);// TopOrBottom(bottom, top, z == 0)
                auto uc = rng.next();
                f2 ub;
                ub.x = rng.next(), ub.y = rng.next();
                auto is = layer_sample<LV>(L, interface_top, -w, uc, ub, mode);
                if (is_black(is.f) || is.pdf <= 0.f) { break; }
                f = f * is.f;
                pdf *= is.pdf;
                w = is.wi;
                w_local = to_local(L.own, w);
                if (is.event == kEventEnter || is.event == kEventExit) {// event_transmit
                    s.f = f, s.pdf = pdf, s.wi = w;
                    s.event = same_hemisphere(w_local, wo_local) ? kEventReflect : (w_local.z > 0.f ? kEventExit : kEventEnter);
                    break;
                }
            }
        }
    }
    if (!valid_sides(L.ng, L.own.n, wo, s.wi)) { s.f = mk3(0.f), s.pdf = 0.f; }
    return s;
}

}// namespace lrd
