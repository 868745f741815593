// dev_bsdf.h — surface closures of the megakernel: microfacet/diffuse BxDF math in the local
// shading frame plus the per-material evaluate/sample switch.
//
// Implements, for CDNA4, the arithmetic of the reference's src/util/scattering.cpp:14-449 and the
// closures src/surfaces/{matte,mirror,glass,plastic,metal}.cpp behind Surface::Closure::evaluate /
// sample (src/base/surface.cpp:45-68).  The reference JIT-specialises one closure class per
// material; here one interpreter switches on DClosure::kind, and the three Fresnel flavours are
// selected by a small enum so the GGX lobe code exists once in the instruction stream.
#pragma once
#include "dev_scene.h"

namespace lrd {

enum : uint32_t { kEventReflect = 0u, kEventEnter = 1u, kEventExit = 2u };// surface.h:46-50
enum : uint32_t { kFresnelSchlick = 0u, kFresnelDielectric = 1u, kFresnelConductor = 2u };

LR_HD float roughness_to_alpha(float r) { return fmaxf(r * r, 1e-4f); }// scattering.cpp:129-135

LR_HD float fresnel_dielectric(float cos_i_in, float eta_i_in, float eta_t_in) {// scattering.cpp:30-52
    auto cos_i = clampf(cos_i_in, -1.f, 1.f);
    auto entering = cos_i > 0.f;
    auto eta_i = entering ? eta_i_in : eta_t_in;
    auto eta_t = entering ? eta_t_in : eta_i_in;
    cos_i = fabsf(cos_i);
    auto sin_i = sqrtf(fmaxf(0.f, 1.f - cos_i * cos_i));
    auto sin_t = eta_i / eta_t * sin_i;
    auto cos_t = sqrtf(fmaxf(0.f, 1.f - sin_t * sin_t));
    auto r_parl = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t);
    auto r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t);
    auto fr = (r_parl * r_parl + r_perp * r_perp) * .5f;
    return sin_t < 1.f ? fr : 1.f;
}

LR_HD f3 fresnel_conductor(float cos_i, float eta_i, f3 eta_t, f3 k) {// scattering.cpp:54-74
    cos_i = clampf(cos_i, -1.f, 1.f);
    auto eta = eta_t / eta_i;
    auto etak = k / eta_i;
    auto cos2 = cos_i * cos_i;
    auto sin2 = 1.f - cos2;
    auto eta2 = eta * eta;
    auto etak2 = etak * etak;
    auto t0 = eta2 - etak2 - mk3(sin2);
    auto a2plusb2 = sqrt3(t0 * t0 + 4.f * eta2 * etak2);
    auto t1 = a2plusb2 + mk3(cos2);
    auto a = sqrt3(.5f * (a2plusb2 + t0));
    auto t2 = 2.f * cos_i * a;
    auto rs = (t1 - t2) / (t1 + t2);
    auto t3 = cos2 * a2plusb2 + mk3(sin2 * sin2);
    auto t4 = t2 * sin2;
    auto rp = rs * (t3 - t4) / (t3 + t4);
    return .5f * (rp + rs);
}

LR_HD float fresnel_dielectric_integral(float eta) {// scattering.cpp:97-107
    auto x = 1.f / eta;
    auto lt = 0.75985009f + eta * (-2.09069066f + eta * (2.23559031f + eta * -0.90663979f));
    auto gt = 0.97945724f + x * (0.21762732f + x * -1.18995376f);
    return saturate(eta == 1.f ? 0.f : (eta < 1.f ? lt : gt));
}

// Fresnel term of a microfacet reflection lobe.  p0/p1/e0 meaning per mode:
//   Schlick    p0 = R0                          (mirror.cpp:67-79)
//   Dielectric e0 = eta_i, e1 = eta_t           (FresnelDielectric::evaluate, not abs)
//   Conductor  e0 = eta_i, p0 = n, p1 = k       (FresnelConductor::evaluate, abs cos)
struct FresnelArgs {
    uint32_t mode;
    f3 p0, p1;
    float e0, e1;
};
LR_HD f3 fresnel_eval(const FresnelArgs &fa, float cos_i) {
    if (fa.mode == kFresnelSchlick) {
        auto m = saturate(1.f - cos_i);
        auto w = sqr(sqr(m)) * m;
        return (1.f - w) * fa.p0 + mk3(w);
    }
    if (fa.mode == kFresnelDielectric) { return mk3(fresnel_dielectric(cos_i, fa.e0, fa.e1)); }
    return fresnel_conductor(fabsf(cos_i), fa.e0, fa.p0, fa.p1);
}

// ---- Trowbridge-Reitz (GGX) distribution, scattering.cpp:117-237; alpha clamped >= 1e-4 (:123-124)
struct GGX {
    float ax, ay;
};
LR_HD GGX make_ggx(float ax, float ay) { return {fmaxf(ax, 1e-4f), fmaxf(ay, 1e-4f)}; }
LR_HD float ggx_D(GGX g, f3 wh) {
    auto tan2 = tan2_theta(wh);
    auto cos4 = sqr(cos2_theta(wh));
    auto e = tan2 * (sqr(cos_phi(wh) / g.ax) + sqr(sin_phi(wh) / g.ay));
    auto d = 1.0f / (kPi * g.ax * g.ay * cos4 * sqr(1.f + e));
    return isinf(tan2) ? 0.f : d;
}
LR_HD float ggx_lambda(GGX g, f3 w) {
    auto tan_t = fabsf(tan_theta(w));
    auto alpha2 = sqr(cos_phi(w)) * sqr(g.ax) + sqr(sin_phi(w)) * sqr(g.ay);
    auto l = (-1.f + sqrtf(1.f + alpha2 * sqr(tan_t))) * .5f;
    return isinf(tan_t) ? 0.f : l;
}
LR_HD float ggx_G1(GGX g, f3 w) { return 1.0f / (1.0f + ggx_lambda(g, w)); }
LR_HD float ggx_G(GGX g, f3 wo, f3 wi) { return 1.0f / (1.0f + ggx_lambda(g, wo) + ggx_lambda(g, wi)); }
LR_HD float ggx_pdf(GGX g, f3 wo, f3 wh) { return ggx_D(g, wh) * ggx_G1(g, wo) * abs_dot(wo, wh) / abs_cos_theta(wo); }

LR_HD f2 ggx_sample11(float cos_t, f2 U) {// visible-normal slope sampling, scattering.cpp:172-208
    if (cos_t <= .9999f) {
        auto sin_t = sqrtf(fmaxf(0.f, 1.f - sqr(cos_t)));
        auto tan_t = sin_t / cos_t;
        auto a = 1.f / tan_t;
        auto G1 = 2.f / (1.f + sqrtf(1.f + 1.f / sqr(a)));
        auto A = 2.f * U.x / G1 - 1.f;
        auto tmp = fminf(1.f / (sqr(A) - 1.f), 1e10f);
        auto B = tan_t;
        auto D = sqrtf(fmaxf(sqr(B * tmp) - (sqr(A) - sqr(B)) * tmp, 0.f));
        auto sx1 = B * tmp - D;
        auto sx2 = B * tmp + D;
        auto sx = ((A < 0.f) || (sx2 * tan_t > 1.f)) ? sx1 : sx2;
        auto S = U.y > .5f ? 1.f : -1.f;
        auto U2 = U.y > .5f ? 2.f * (U.y - .5f) : 2.f * (.5f - U.y);
        auto z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                 (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        return {sx, S * z * sqrtf(1.f + sqr(sx))};
    }
    auto r = sqrtf(U.x / (1.f - U.x));
    auto phi = (2.f * kPi) * U.y;
    return {r * cosf(phi), r * sinf(phi)};
}
LR_HD f3 ggx_sample_wh(GGX g, f3 wo, f2 u) {// scattering.cpp:210-237
    auto s = sign(cos_theta(wo));
    auto wi = s * wo;
    auto stretched = normalize(mk3(g.ax * wi.x, g.ay * wi.y, wi.z));
    auto slope = ggx_sample11(cos_theta(stretched), u);
    auto cp = cos_phi(stretched), sp = sin_phi(stretched);
    f2 rot{cp * slope.x - sp * slope.y, sp * slope.x + cp * slope.y};
    auto wh = normalize(mk3(-(g.ax * rot.x), -(g.ay * rot.y), 1.f));
    return s * wh;
}

// ---- sampling helpers, src/util/sampling.cpp:13-31
LR_HD f2 sample_disk_concentric(f2 u_in) {
    f2 u{u_in.x * 2.0f - 1.0f, u_in.y * 2.0f - 1.0f};
    auto p = fabsf(u.x) > fabsf(u.y);
    auto r = p ? u.x : u.y;
    auto theta = p ? kPiOverFour * (u.y / u.x) : kPiOverTwo - kPiOverFour * (u.x / u.y);
    return {r * cosf(theta), r * sinf(theta)};
}
LR_HD f3 sample_cosine_hemisphere(f2 u) {
    auto d = sample_disk_concentric(u);
    return {d.x, d.y, sqrtf(fmaxf(1.0f - d.x * d.x - d.y * d.y, 0.0f))};
}
LR_HD float cosine_pdf(f3 wo, f3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.f; }
LR_HD f3 cosine_sample_wi(f3 wo, f2 u) {// BxDF::sample_wi, scattering.cpp:260-264
    auto wi = sample_cosine_hemisphere(u);
    wi.z *= sign(cos_theta(wo));
    return wi;
}

// ---- lobes
LR_HD f3 oren_nayar_eval(f3 r, float sigma_deg, f3 wo, f3 wi) {// scattering.cpp:370-400
    auto sigma = sigma_deg * (kPi / 180.f);
    auto sigma2 = sigma * sigma;
    auto a = 1.f - (sigma2 / (2.f * sigma2 + 0.66f));
    auto b = 0.45f * sigma2 / (sigma2 + 0.09f);
    auto s = same_hemisphere(wo, wi) ? kInvPi : 0.f;
    auto sin_i = sin_theta(wi), sin_o = sin_theta(wo);
    auto d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
    auto max_cos = (sin_i > 1e-4f && sin_o > 1e-4f) ? fmaxf(0.f, d_cos) : 0.f;
    auto aci = abs_cos_theta(wi), aco = abs_cos_theta(wo);
    auto sin_alpha = aci > aco ? sin_o : sin_i;
    auto tan_beta = aci > aco ? sin_i / aci : sin_o / aco;
    return s * (a + b * max_cos * sin_alpha * tan_beta) * r;
}

LR_HD f3 mf_reflection_eval(f3 R, GGX g, const FresnelArgs &fa, f3 wo, f3 wi) {// scattering.cpp:286-303
    auto wh = wi + wo;
    auto f = mk3(0.f);
    if (same_hemisphere(wo, wi) && (wh.x != 0.f || wh.y != 0.f || wh.z != 0.f)) {
        wh = normalize(wh);
        auto F = fresnel_eval(fa, dot(wi, face_forward(wh, mk3(0.f, 0.f, 1.f))));
        f = R * F * fabsf(0.25f * ggx_D(g, wh) * ggx_G(g, wo, wi) / (cos_theta(wi) * cos_theta(wo)));
    }
    return f;
}
LR_HD float mf_reflection_pdf(GGX g, f3 wo, f3 wi) {// scattering.cpp:311-320
    auto wh = wi + wo;
    auto p = 0.f;
    if (same_hemisphere(wo, wi) && (wh.x != 0.f || wh.y != 0.f || wh.z != 0.f)) {
        wh = normalize(wh);
        p = ggx_pdf(g, wo, wh) / (4.f * dot(wo, wh));
    }
    return p;
}

LR_HD bool refract_dir(f3 wi, f3 n, float eta, f3 &wt) {// scattering.cpp:14-28
    auto cos_i = dot(n, wi);
    auto sin2_i = fmaxf(0.0f, 1.f - cos_i * cos_i);
    auto sin2_t = eta * eta * sin2_i;
    auto cos_t = sqrtf(1.f - sin2_t);
    wt = (eta * cos_i - cos_t) * n - eta * wi;
    return sin2_t < 1.0f;
}
LR_HD f3 mf_transmission_eval(f3 T, GGX g, float eta_a, float eta_b, f3 wo, f3 wi) {// scattering.cpp:322-346
    auto cos_o = cos_theta(wo), cos_i = cos_theta(wi);
    auto eta = cos_o > 0.f ? eta_b / eta_a : eta_a / eta_b;
    auto wh = normalize(wo + wi * eta);
    wh = sign(cos_theta(wh)) * wh;
    auto f = mk3(0.f);
    if (!same_hemisphere(wo, wi) && cos_o != 0.f && cos_i != 0.f && dot(wo, wh) * dot(wi, wh) < 0.f) {
        auto G = ggx_G(g, wo, wi);
        auto sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        auto F = fresnel_dielectric(dot(wo, wh), eta_a, eta_b);
        auto D = ggx_D(g, wh);
        f = (1.f - F) * T * D * G * dot(wi, wh) * dot(wo, wh) / (cos_i * cos_o * sqr(sqrt_denom));
    }
    return f;
}
LR_HD float mf_transmission_pdf(GGX g, float eta_a, float eta_b, f3 wo, f3 wi) {// scattering.cpp:355-368
    auto pdf = 0.f;
    auto eta = cos_theta(wo) > 0.f ? eta_b / eta_a : eta_a / eta_b;
    auto wh = normalize(wo + wi * eta);
    if (!same_hemisphere(wo, wi) && dot(wo, wh) * dot(wi, wh) < 0.f) {
        auto sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        auto dwh_dwi = sqr(eta / sqrt_denom) * abs_dot(wi, wh);
        pdf = ggx_pdf(g, wo, wh) * dwh_dwi;
    }
    return pdf;
}

// ---- closures
struct BsdfEval {
    f3 f;      // includes |cos theta_i|
    float pdf;
};
struct BsdfSample {
    f3 f;
    float pdf;
    f3 wi;     // world space
    uint32_t event;
};

LR_HD bool valid_sides(f3 ng, f3 ns, f3 wo, f3 wi) {// validate_surface_sides, surface.cpp:35-43
    auto flip = sign(dot(ng, ns));
    return sign(flip * dot(wo, ns)) == sign(dot(wo, ng)) && sign(flip * dot(wi, ns)) == sign(dot(wi, ng));
}

LR_HD FresnelArgs closure_fresnel(const DClosure &c) {
    FresnelArgs fa;
    fa.p0 = mk3(c.c0[0], c.c0[1], c.c0[2]);
    fa.p1 = mk3(c.c1[0], c.c1[1], c.c1[2]);
    fa.e0 = c.s0, fa.e1 = c.s1;
    fa.mode = c.kind == LR_SURFACE_MIRROR ? kFresnelSchlick : (c.kind == LR_SURFACE_METAL ? kFresnelConductor : kFresnelDielectric);
    if (c.kind == LR_SURFACE_PLASTIC) { fa.e0 = 1.f, fa.e1 = c.s1; }
    return fa;
}

LR_HD float glass_refl_prob(const DClosure &c, f3 wo_l) {// glass.cpp:160-166
    auto F = fresnel_dielectric(cos_theta(wo_l), c.s0, c.s1);
    auto r = c.s2 * F;
    auto t = (1.f - c.s2) * (1.f - F);
    return r == 0.f ? 0.f : r / (r + t);
}
LR_HD float plastic_substrate_weight(float Fo, float kd_weight) {// plastic.cpp:126-129
    auto w = kd_weight * (1.0f - Fo);
    return w == 0.f ? 0.f : w / (w + Fo);
}
// coat + absorbing diffuse substrate in the flipped-to-+z local frame, plastic.cpp:147-163
LR_HD BsdfEval plastic_eval_local(const DClosure &c, GGX g, f3 wo_l, f3 wi_l) {
    FresnelArgs fa;
    fa.mode = kFresnelDielectric, fa.e0 = 1.f, fa.e1 = c.s1, fa.p0 = mk3(0.f), fa.p1 = mk3(0.f);
    auto eta = c.s1;
    auto f_coat = mf_reflection_eval(mk3(1.f), g, fa, wo_l, wi_l);
    auto pdf_coat = mf_reflection_pdf(g, wo_l, wi_l);
    auto Fi = fresnel_dielectric(abs_cos_theta(wi_l), 1.f, eta);
    auto Fo = fresnel_dielectric(abs_cos_theta(wo_l), 1.f, eta);
    auto sigma_a = mk3(c.c1[0], c.c1[1], c.c1[2]);
    auto a = exp3(-(1.f / abs_cos_theta(wi_l) + 1.f / abs_cos_theta(wo_l)) * sigma_a);
    auto kd = mk3(c.c0[0], c.c0[1], c.c0[2]);
    auto lambert = kd * (same_hemisphere(wo_l, wi_l) ? kInvPi : 0.f);
    auto f_diffuse = (1.f - Fi) * (1.f - Fo) * sqr(1.f / eta) * a * lambert;
    auto pdf_diffuse = cosine_pdf(wo_l, wi_l);
    auto w = plastic_substrate_weight(Fo, c.s0);
    return {(f_coat + f_diffuse) * abs_cos_theta(wi_l), lerp(pdf_coat, pdf_diffuse, w)};
}

// Surface::Closure::evaluate
LR_HD BsdfEval closure_evaluate(const DClosure &c, const Frame &sh, f3 ng, f3 wo, f3 wi) {
    auto wo_l = to_local(sh, wo);
    auto wi_l = to_local(sh, wi);
    BsdfEval e{mk3(0.f), 0.f};
    auto g = make_ggx(c.alpha_x, c.alpha_y);
    if (c.kind == LR_SURFACE_MATTE) {// matte.cpp:86-96
        e.f = oren_nayar_eval(mk3(c.c0[0], c.c0[1], c.c0[2]), c.s0, wo_l, wi_l) * abs_cos_theta(wi_l);
        e.pdf = cosine_pdf(wo_l, wi_l);
    } else if (c.kind == LR_SURFACE_PLASTIC) {// plastic.cpp:139-166
        auto flip = cos_theta(wo_l) < 0.f ? -1.f : 1.f;
        wo_l.z *= flip, wi_l.z *= flip;
        e = plastic_eval_local(c, g, wo_l, wi_l);
    } else if (c.kind == LR_SURFACE_GLASS && !same_hemisphere(wo_l, wi_l)) {// glass.cpp:186-190
        auto ratio = glass_refl_prob(c, wo_l);
        e.f = mf_transmission_eval(mk3(c.c1[0], c.c1[1], c.c1[2]), g, c.s0, c.s1, wo_l, wi_l) * abs_cos_theta(wi_l);
        e.pdf = mf_transmission_pdf(g, c.s0, c.s1, wo_l, wi_l) * (1.f - ratio);
    } else if (c.kind != LR_SURFACE_NULL) {// mirror.cpp:101-115, metal.cpp:228-241, glass.cpp:182-185
        auto fa = closure_fresnel(c);
        auto R = c.kind == LR_SURFACE_METAL ? mk3(1.f) : mk3(c.c0[0], c.c0[1], c.c0[2]);
        auto f = mf_reflection_eval(R, g, fa, wo_l, wi_l);
        auto pdf = mf_reflection_pdf(g, wo_l, wi_l);
        if (c.kind == LR_SURFACE_METAL) { f = f * mk3(c.c2[0], c.c2[1], c.c2[2]); }
        if (c.kind == LR_SURFACE_GLASS) { pdf *= glass_refl_prob(c, wo_l); }
        e.f = f * abs_cos_theta(wi_l);
        e.pdf = pdf;
    }
    if (!valid_sides(ng, sh.n, wo, wi)) { e.f = mk3(0.f), e.pdf = 0.f; }
    return e;
}

// Surface::Closure::sample
LR_HD BsdfSample closure_sample(const DClosure &c, const Frame &sh, f3 ng, f3 wo, float u_lobe, f2 u) {
    auto wo_l = to_local(sh, wo);
    BsdfSample s{mk3(0.f), 0.f, mk3(0.f, 0.f, 1.f), kEventReflect};
    auto g = make_ggx(c.alpha_x, c.alpha_y);
    if (c.kind == LR_SURFACE_MATTE) {// matte.cpp:98-112
        auto wi_l = cosine_sample_wi(wo_l, u);
        s.pdf = cosine_pdf(wo_l, wi_l);
        s.f = oren_nayar_eval(mk3(c.c0[0], c.c0[1], c.c0[2]), c.s0, wo_l, wi_l) * abs_cos_theta(wi_l);
        s.wi = to_world(sh, wi_l);
    } else if (c.kind == LR_SURFACE_PLASTIC) {// plastic.cpp:168-213
        auto flip = cos_theta(wo_l) < 0.f ? -1.f : 1.f;
        wo_l.z *= flip;
        auto Fo = fresnel_dielectric(abs_cos_theta(wo_l), 1.f, c.s1);
        auto w = plastic_substrate_weight(Fo, c.s0);
        f3 wi_l;
        auto valid = true;
        if (u_lobe < w) {
            wi_l = cosine_sample_wi(wo_l, u);
        } else {
            wi_l = reflect(-wo_l, ggx_sample_wh(g, wo_l, u));
            valid = same_hemisphere(wo_l, wi_l);
        }
        if (valid) {
            auto e = plastic_eval_local(c, g, wo_l, wi_l);
            s.f = e.f, s.pdf = e.pdf;
            wi_l.z *= flip;
            s.wi = to_world(sh, wi_l);
        }
    } else if (c.kind != LR_SURFACE_NULL) {
        auto ratio = c.kind == LR_SURFACE_GLASS ? glass_refl_prob(c, wo_l) : 1.f;
        auto wh = ggx_sample_wh(g, wo_l, u);
        f3 wi_l;
        if (c.kind != LR_SURFACE_GLASS || u_lobe < ratio) {// reflection: BxDF::sample, scattering.cpp:247-254
            wi_l = reflect(-wo_l, wh);
            if (same_hemisphere(wo_l, wi_l)) {
                auto fa = closure_fresnel(c);
                auto R = c.kind == LR_SURFACE_METAL ? mk3(1.f) : mk3(c.c0[0], c.c0[1], c.c0[2]);
                s.f = mf_reflection_eval(R, g, fa, wo_l, wi_l);
                s.pdf = mf_reflection_pdf(g, wo_l, wi_l);
                if (c.kind == LR_SURFACE_METAL) { s.f = s.f * mk3(c.c2[0], c.c2[1], c.c2[2]); }
            }
            s.pdf *= ratio;
        } else {// glass transmission, glass.cpp:216-223, scattering.cpp:348-353
            auto eta = cos_theta(wo_l) > 0.f ? c.s0 / c.s1 : c.s1 / c.s0;
            wi_l = mk3(0.f);
            auto refr = refract_dir(wo_l, wh, eta, wi_l);
            if (refr && !same_hemisphere(wo_l, wi_l)) {
                s.f = mf_transmission_eval(mk3(c.c1[0], c.c1[1], c.c1[2]), g, c.s0, c.s1, wo_l, wi_l);
                s.pdf = mf_transmission_pdf(g, c.s0, c.s1, wo_l, wi_l);
            }
            s.pdf *= (1.f - ratio);
            s.event = cos_theta(wo_l) > 0.f ? kEventEnter : kEventExit;
        }
        s.f = s.f * abs_cos_theta(wi_l);
        s.wi = to_world(sh, wi_l);
    }
    if (!valid_sides(ng, sh.n, wo, s.wi)) { s.f = mk3(0.f), s.pdf = 0.f; }
    return s;
}

}// namespace lrd
