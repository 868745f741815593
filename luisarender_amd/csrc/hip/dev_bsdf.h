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
    float sn, cs;
    sincos_2pi(U.y, sn, cs);// phi = 2 pi U.y
    return {r * cs, r * sn};
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
    // theta = p ? pi/4 (u.y / u.x) : pi/2 - pi/4 (u.x / u.y): both quotients lie in [-1, 1], so sin / cos need no range reduction
    // (cos(pi/2 - a) = sin a, sin(pi/2 - a) = cos a); dev_math.h: sincos_small, 1 ulp, ~14 instructions instead of 232
    float sa, ca;
    sincos_small(kPiOverFour * (p ? u.y / u.x : u.x / u.y), sa, ca);
    return {r * (p ? ca : sa), r * (p ? sa : ca)};
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
LR_HD f3 mf_transmission_eval(f3 T, GGX g, float eta_a, float eta_b, f3 wo, f3 wi, bool importance = false) {// scattering.cpp:322-346
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
        if (importance) { f = f * sqr(eta); }// TransportMode::IMPORTANCE (Layered's reverse walks only)
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
// ---- Disney (src/surfaces/disney.cpp): thick (:351-588) and thin (:590-841) closures on one structure
LR_HD float schlick_weight(float c) {
    auto m = saturate(1.f - c);
    return sqr(sqr(m)) * m;
}
LR_HD float fr_schlick(float R0, float c) { return lerp(R0, 1.f, schlick_weight(c)); }
LR_HD float gtr1(float c, float alpha) {// :210-214
    auto a2 = alpha * alpha;
    return (a2 - 1.f) / (kPi * logf(a2) * (1.f + (a2 - 1.f) * c * c));
}
LR_HD float smith_g_ggx(float c, float alpha) {// :217-221
    auto a2 = alpha * alpha, c2 = c * c;
    return 1.f / (c + sqrtf(a2 + c2 - a2 * c2));
}

struct DisneyLobes {
    f3 Cdiff, Css, Csheen, Cspec0, Cst, Cdt;
    float roughness, metallic, eta, eta_i, eta_t, clearcoat, gloss;
    GGX dist, tdist;
    float w[5];
    uint32_t mask;  // bit i: technique i enabled; bits 8.. : has_diffuse(8) fake_ss(9) sheen(10) clearcoat(11) spec_trans(12) diff_trans(13)
    uint32_t count;
    bool thin, two_sided;
};

LR_HD DisneyLobes disney_setup(const DClosure &c) {
    DisneyLobes L;
    auto color = mk3(c.c0[0], c.c0[1], c.c0[2]);
    auto color_lum = c.s0;
    auto metallic = c.e[kDisneyMetallic], specular_trans = c.e[kDisneySpecularTrans], diffuse_trans = c.s1;
    auto flatness = c.e[kDisneyFlatness], sheen = c.e[kDisneySheen], sheen_tint = c.e[kDisneySheenTint];
    auto lobes = c.x[0];
    L.thin = c.x[2] != 0u;
    auto transmissive = c.x[1] != 0u;
    auto diffuse_weight = (1.f - metallic) * (1.f - specular_trans);
    auto diff_refl_weight = L.thin ? diffuse_weight * (1.f - diffuse_trans) : diffuse_weight;
    auto diff_trans_weight = diffuse_weight * diffuse_trans;
    auto tint_weight = color_lum > 0.f ? 1.f / color_lum : 1.f;
    auto tint = saturate(color * tint_weight);
    auto tint_lum = color_lum * tint_weight;
    L.roughness = c.e[kDisneyRoughness], L.metallic = metallic;
    L.mask = 0u;
    L.Cdiff = L.Css = L.Csheen = L.Cst = L.Cdt = mk3(0.f);
    L.w[0] = L.w[1] = L.w[2] = L.w[3] = L.w[4] = 0.f;
    auto diffuse_like = diff_refl_weight * color_lum;
    if (lobes & 3u) {
        L.Cdiff = color * (diff_refl_weight * (1.f - flatness));
        L.mask |= 1u | (1u << 8u);
    }
    if (lobes & 4u) {
        auto Css_weight = L.thin ? diff_refl_weight * flatness * (1.f - diffuse_trans) : diffuse_weight * flatness;
        L.Css = Css_weight * color;
        L.mask |= 1u | (1u << 9u);
    }
    if (lobes & 8u) {
        auto Csheen_weight = L.thin ? diff_refl_weight * sheen * (1.f - diffuse_trans) : diffuse_weight * sheen;
        L.Csheen = Csheen_weight * (mk3(1.f) + sheen_tint * (tint - mk3(1.f)));
        L.mask |= (1u << 10u) | (L.thin ? 0u : 1u);// the thin closure does not enable the technique for sheen alone
        diffuse_like += Csheen_weight * lerp(1.f, tint_lum, sheen_tint) * .1f;
    }
    L.w[0] = saturate(diffuse_like);
    L.eta_i = c.e[kDisneyEtaI], L.eta_t = c.e[kDisneyEtaT];
    L.eta = L.eta_t / L.eta_i;
    auto R0 = sqr((L.eta - 1.f) / (L.eta + 1.f));
    auto spec_tint = c.e[kDisneySpecularTint];
    auto base = (mk3(1.f) + spec_tint * (tint - mk3(1.f))) * R0;
    L.Cspec0 = base + metallic * (color - base);
    L.two_sided = L.thin ? false : !transmissive;
    auto aspect = sqrtf(1.f - c.e[kDisneyAnisotropic] * .9f);
    L.dist = make_ggx(fmaxf(0.001f, L.roughness / aspect), fmaxf(0.001f, L.roughness * aspect));
    L.tdist = L.dist;
    L.w[1] = saturate(lerp(lerp(1.f, tint_lum, spec_tint) * R0, color_lum, metallic));
    L.mask |= 2u;
    L.clearcoat = 0.f, L.gloss = 0.f;
    if (lobes & 16u) {
        L.gloss = lerp(.1f, .001f, c.e[kDisneyClearcoatGloss]);
        L.clearcoat = c.e[kDisneyClearcoat];
        L.w[2] = saturate(L.clearcoat * fr_schlick(.04f, 1.f));
        L.mask |= 4u | (1u << 11u);
    }
    if (L.thin) {
        L.count = 5u;
        if (lobes & 128u) {
            auto rscaled = (.65f * L.eta - .35f) * L.roughness;
            L.tdist = make_ggx(fmaxf(.001f, rscaled / aspect), fmaxf(.001f, rscaled * aspect));
            auto Cst_weight = (1.f - metallic) * specular_trans;
            L.Cst = Cst_weight * color;
            L.w[3] = saturate(Cst_weight * color_lum);
            L.mask |= 8u | (1u << 12u);
        }
        if (lobes & 64u) {
            L.Cdt = diff_trans_weight * color;
            L.w[4] = saturate(diff_trans_weight * color_lum);
            L.mask |= 16u | (1u << 13u);
        }
    } else {
        L.count = transmissive ? 4u : 3u;
        if (transmissive && (lobes & 128u)) {
            auto Cst_weight = (1.f - metallic) * specular_trans;
            L.Cst = Cst_weight * sqrt3(color);
            L.w[3] = saturate(Cst_weight * sqrtf(color_lum));
            L.mask |= 8u | (1u << 12u);
        }
    }
    auto sum = 0.f;
    for (auto i = 0u; i < 5u; i++) {
        if (i < L.count && (L.mask & (1u << i))) { sum += L.w[i]; }
    }
    auto inv = sum == 0.f ? 0.f : 1.f / sum;
    for (auto i = 0u; i < 5u; i++) {
        if (i < L.count && (L.mask & (1u << i))) { L.w[i] *= inv; }
    }
    return L;
}

LR_HD f3 disney_fresnel(const DisneyLobes &L, float cos_in) {// DisneyFresnel::evaluate, :277-296
    auto cosI = L.two_sided ? fabsf(cos_in) : cos_in;
    auto fr = fresnel_dielectric(cosI, 1.f, L.eta);
    auto sw = schlick_weight(cosI);
    auto f0 = L.Cspec0 + sw * (mk3(1.f) - L.Cspec0);
    return mk3(fr) + L.metallic * (f0 - mk3(fr));
}

LR_HD BsdfEval disney_eval_local(const DisneyLobes &L, f3 wo, f3 wi, bool importance = false) {// _evaluate_local, :476-521 / :728-781
    auto f = mk3(0.f);
    auto pdf = 0.f;
    if (same_hemisphere(wo, wi)) {
        auto wh = wi + wo;
        auto valid = wh.x != 0.f || wh.y != 0.f || wh.z != 0.f;
        wh = normalize(wh);
        if ((L.mask & (1u << 8u)) && L.w[0] > 0.f) {
            auto Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
            f += L.Cdiff * (kInvPi * (1.f - Fo * .5f) * (1.f - Fi * .5f));
            auto cos_d = dot(wi, wh);
            auto Rr = 2.f * L.roughness * cos_d * cos_d;
            f += L.Cdiff * (valid ? kInvPi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1.f)) : 0.f);
            if (L.mask & (1u << 9u)) {
                auto Fss90 = cos_d * cos_d * L.roughness;
                auto Fss = lerp(1.0f, Fss90, Fo) * lerp(1.0f, Fss90, Fi);
                auto ss = 1.25f * (Fss * (1.f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - .5f) + .5f);
                f += L.Css * (valid ? kInvPi * ss : 0.f);
            }
            if (L.mask & (1u << 10u)) { f += L.Csheen * (valid ? schlick_weight(cos_d) : 0.f); }
            pdf += L.w[0] * cosine_pdf(wo, wi);
        }
        if (L.w[1] > 0.f) {// MicrofacetReflection with the Disney Fresnel
            if (valid) {
                auto F = disney_fresnel(L, dot(wi, face_forward(wh, mk3(0.f, 0.f, 1.f))));
                f += F * fabsf(0.25f * ggx_D(L.dist, wh) * ggx_G(L.dist, wo, wi) / (cos_theta(wi) * cos_theta(wo)));
                pdf += L.w[1] * (ggx_pdf(L.dist, wo, wh) / (4.f * dot(wo, wh)));
            }
        }
        if ((L.mask & (1u << 11u)) && L.w[2] > 0.f) {// DisneyClearcoat, :232-279
            auto Dr = gtr1(abs_cos_theta(wh), L.gloss);
            auto Fr = fr_schlick(.04f, dot(wo, wh));
            auto Gr = smith_g_ggx(abs_cos_theta(wo), .25f) * smith_g_ggx(abs_cos_theta(wi), .25f);
            f += mk3(valid ? L.clearcoat * Gr * Fr * Dr * .25f : 0.f);
            pdf += L.w[2] * (valid ? Dr * abs_cos_theta(wh) / (4.f * dot(wo, wh)) : 0.f);
        }
    } else {
        if ((L.mask & (1u << 12u)) && L.w[3] > 0.f) {
            f += mf_transmission_eval(L.Cst, L.tdist, L.eta_i, L.eta_t, wo, wi, importance);
            pdf += L.w[3] * mf_transmission_pdf(L.tdist, L.eta_i, L.eta_t, wo, wi);
        }
        if ((L.mask & (1u << 13u)) && L.w[4] > 0.f) {// LambertianTransmission
            f += L.Cdt * kInvPi;
            pdf += L.w[4] * (abs_cos_theta(wi) * kInvPi);
        }
    }
    return {f * abs_cos_theta(wi), pdf};
}

LR_HD void disney_sample_local(const DisneyLobes &L, f3 wo, float u_lobe, f2 u, f3 &wi, bool &valid, uint32_t &event) {// :538-587
    auto tech = 0u;
    auto sum = 0.f;
    for (auto i = 0u; i < 5u; i++) {
        if (i < L.count && (L.mask & (1u << i))) {
            tech = u_lobe > sum ? i : tech;
            sum += L.w[i];
        }
    }
    event = kEventReflect;
    valid = false;
    wi = mk3(0.f);
    if (tech == 0u && (L.mask & (1u << 8u))) {
        wi = cosine_sample_wi(wo, u), valid = true;
    } else if (tech == 1u) {
        wi = reflect(-wo, ggx_sample_wh(L.dist, wo, u));
        valid = same_hemisphere(wo, wi);
    } else if (tech == 2u && (L.mask & (1u << 11u))) {// DisneyClearcoat::sample_wi, :250-265
        auto a2 = L.gloss * L.gloss;
        auto ct = sqrtf(fmaxf(0.f, (1.f - powf(a2, 1.f - u.x)) / (1.f - a2)));
        auto st = sqrtf(fmaxf(0.f, 1.f - ct * ct));
        float sn, cs;
        sincos_2pi(u.y, sn, cs);// phi = 2 pi u.y
        auto wh = mk3(st * cs, st * sn, ct);
        wh = same_hemisphere(wo, wh) ? wh : -wh;
        wi = reflect(-wo, wh);
        valid = same_hemisphere(wo, wi);
    } else if (tech == 3u && (L.mask & (1u << 12u))) {
        auto e = cos_theta(wo) > 0.f ? L.eta_i / L.eta_t : L.eta_t / L.eta_i;
        auto refr = refract_dir(wo, ggx_sample_wh(L.tdist, wo, u), e, wi);
        valid = refr && !same_hemisphere(wo, wi);
        event = L.thin ? 4u : (cos_theta(wo) > 0.f ? kEventEnter : kEventExit);// 4 = Surface::event_through
    } else if (tech == 4u && (L.mask & (1u << 13u))) {
        wi = sample_cosine_hemisphere(u);
        wi.z *= -sign(cos_theta(wo));
        valid = true;
        event = 4u;
    }
}

// ---- the five basic closures in the local shading frame.
// Divergence is what the shading block pays for (every closure kind present in a wave runs one after the other), so the
// kinds share code wherever the reference's arithmetic is the same: ONE microfacet-reflection block serves Mirror, Metal,
// the reflection lobe of Glass and the coat of Plastic; ONE visible-normal sample serves all of them; and
// Surface::Closure::sample is "pick wi per kind, then evaluate" on the SAME evaluation code as Surface::Closure::evaluate
// (the reference evaluates the sampled direction with the same functions: BxDF::sample = sample_wi + evaluate + pdf,
// scattering.cpp:247-254).  Static VALU of evaluate + sample: 3456 -> see DESIGN.md §4.6.
// Inputs of basic_eval_local are in the frame Plastic works in (flipped so that wo is in +z, plastic.cpp:141-145).
LR_HD BsdfEval basic_eval_local(const DClosure &c, GGX g, f3 wo_l, f3 wi_l, bool importance) {
    BsdfEval e{mk3(0.f), 0.f};
    const auto kind = c.kind;
    const auto same = same_hemisphere(wo_l, wi_l);
    const auto reflective = kind == LR_SURFACE_PLASTIC || kind == LR_SURFACE_MIRROR || kind == LR_SURFACE_METAL || (kind == LR_SURFACE_GLASS && same);
    auto f_r = mk3(0.f);
    auto pdf_r = 0.f;
    if (reflective) {// mirror.cpp:101-115, metal.cpp:228-241, glass.cpp:182-185, plastic.cpp:147-150
        auto fa = closure_fresnel(c);
        auto R = (kind == LR_SURFACE_METAL || kind == LR_SURFACE_PLASTIC) ? mk3(1.f) : mk3(c.c0[0], c.c0[1], c.c0[2]);
        f_r = mf_reflection_eval(R, g, fa, wo_l, wi_l);
        pdf_r = mf_reflection_pdf(g, wo_l, wi_l);
    }
    if (kind == LR_SURFACE_MATTE) {// matte.cpp:86-96
        e.f = oren_nayar_eval(mk3(c.c0[0], c.c0[1], c.c0[2]), c.s0, wo_l, wi_l) * abs_cos_theta(wi_l);
        e.pdf = cosine_pdf(wo_l, wi_l);
    } else if (kind == LR_SURFACE_PLASTIC) {// coat + absorbing diffuse substrate, plastic.cpp:147-163
        auto eta = c.s1;
        auto Fi = fresnel_dielectric(abs_cos_theta(wi_l), 1.f, eta);
        auto Fo = fresnel_dielectric(abs_cos_theta(wo_l), 1.f, eta);
        auto sigma_a = mk3(c.c1[0], c.c1[1], c.c1[2]);
        auto a = exp3(-(1.f / abs_cos_theta(wi_l) + 1.f / abs_cos_theta(wo_l)) * sigma_a);
        auto kd = mk3(c.c0[0], c.c0[1], c.c0[2]);
        auto lambert = kd * (same ? kInvPi : 0.f);
        auto f_diffuse = (1.f - Fi) * (1.f - Fo) * sqr(1.f / eta) * a * lambert;
        auto pdf_diffuse = cosine_pdf(wo_l, wi_l);
        auto w = plastic_substrate_weight(Fo, c.s0);
        e.f = (f_r + f_diffuse) * abs_cos_theta(wi_l);
        e.pdf = lerp(pdf_r, pdf_diffuse, w);
    } else if (kind == LR_SURFACE_GLASS && !same) {// glass.cpp:186-190
        auto ratio = glass_refl_prob(c, wo_l);
        e.f = mf_transmission_eval(mk3(c.c1[0], c.c1[1], c.c1[2]), g, c.s0, c.s1, wo_l, wi_l, importance) * abs_cos_theta(wi_l);
        e.pdf = mf_transmission_pdf(g, c.s0, c.s1, wo_l, wi_l) * (1.f - ratio);
    } else if (kind != LR_SURFACE_NULL) {
        if (kind == LR_SURFACE_METAL) { f_r = f_r * mk3(c.c2[0], c.c2[1], c.c2[2]); }
        if (kind == LR_SURFACE_GLASS) { pdf_r *= glass_refl_prob(c, wo_l); }
        e.f = f_r * abs_cos_theta(wi_l);
        e.pdf = pdf_r;
    }
    return e;
}

// Surface::Closure::evaluate.  FULL = false compiles the Disney interpreter out (lean kernel variant).
template<bool FULL>
LR_HD BsdfEval closure_evaluate(const DClosure &c, const Frame &sh, f3 ng, f3 wo, f3 wi, bool importance = false) {
    auto wo_l = to_local(sh, wo);
    auto wi_l = to_local(sh, wi);
    BsdfEval e{mk3(0.f), 0.f};
    if (FULL && c.kind == LR_SURFACE_DISNEY) {
        e = disney_eval_local(disney_setup(c), wo_l, wi_l, importance);
    } else {
        auto flip = (c.kind == LR_SURFACE_PLASTIC && cos_theta(wo_l) < 0.f) ? -1.f : 1.f;// plastic.cpp:141-145
        wo_l.z *= flip, wi_l.z *= flip;
        e = basic_eval_local(c, make_ggx(c.alpha_x, c.alpha_y), wo_l, wi_l, importance);
    }
    if (!valid_sides(ng, sh.n, wo, wi)) { e.f = mk3(0.f), e.pdf = 0.f; }
    return e;
}

// Surface::Closure::sample
template<bool FULL>
LR_HD BsdfSample closure_sample(const DClosure &c, const Frame &sh, f3 ng, f3 wo, float u_lobe, f2 u, bool importance = false) {
    auto wo_l = to_local(sh, wo);
    BsdfSample s{mk3(0.f), 0.f, mk3(0.f, 0.f, 1.f), kEventReflect};
    if (FULL && c.kind == LR_SURFACE_DISNEY) {
        auto L = disney_setup(c);
        f3 wi_l;
        bool valid;
        disney_sample_local(L, wo_l, u_lobe, u, wi_l, valid, s.event);
        s.wi = to_world(sh, wi_l);
        if (valid) {
            auto e = disney_eval_local(L, wo_l, wi_l, importance);
            s.f = e.f, s.pdf = e.pdf;
        }
    } else if (c.kind != LR_SURFACE_NULL) {
        const auto kind = c.kind;
        auto g = make_ggx(c.alpha_x, c.alpha_y);
        auto flip = (kind == LR_SURFACE_PLASTIC && cos_theta(wo_l) < 0.f) ? -1.f : 1.f;
        wo_l.z *= flip;
        // ---- which lobe (plastic.cpp:168-181, glass.cpp:203-207)
        auto diffuse = kind == LR_SURFACE_MATTE;
        auto transmit = false;
        if (kind == LR_SURFACE_PLASTIC) {
            auto Fo = fresnel_dielectric(abs_cos_theta(wo_l), 1.f, c.s1);
            diffuse = u_lobe < plastic_substrate_weight(Fo, c.s0);
        } else if (kind == LR_SURFACE_GLASS) {
            transmit = !(u_lobe < glass_refl_prob(c, wo_l));
        }
        // ---- the direction
        f3 wi_l;
        auto valid = true;
        if (diffuse) {// matte.cpp:98-112, plastic.cpp:183-186
            wi_l = cosine_sample_wi(wo_l, u);
        } else {
            auto wh = ggx_sample_wh(g, wo_l, u);
            if (!transmit) {// reflection: BxDF::sample, scattering.cpp:247-254
                wi_l = reflect(-wo_l, wh);
                valid = same_hemisphere(wo_l, wi_l);
            } else {// glass transmission, glass.cpp:216-223, scattering.cpp:348-353
                auto eta = cos_theta(wo_l) > 0.f ? c.s0 / c.s1 : c.s1 / c.s0;
                wi_l = mk3(0.f);
                auto refr = refract_dir(wo_l, wh, eta, wi_l);
                valid = refr && !same_hemisphere(wo_l, wi_l);
                s.event = cos_theta(wo_l) > 0.f ? kEventEnter : kEventExit;
            }
        }
        // ---- its value: the same code Surface::Closure::evaluate runs
        if (valid) {
            auto e = basic_eval_local(c, g, wo_l, wi_l, importance);
            s.f = e.f, s.pdf = e.pdf;
        }
        if (valid || kind != LR_SURFACE_PLASTIC) {// (an invalid Plastic sample leaves wi at its default, plastic.cpp:196-206)
            wi_l.z *= flip;
            s.wi = to_world(sh, wi_l);
        }
    }
    if (!valid_sides(ng, sh.n, wo, s.wi)) { s.f = mk3(0.f), s.pdf = 0.f; }
    return s;
}

// Surface::Closure::eta (glass.cpp:151-153, disney.cpp:531-533): the relative index a transmissive sample crosses
LR_HD bool closure_eta(const DClosure &c, float &eta) {
    if (c.kind == LR_SURFACE_GLASS) { eta = c.s1; return true; }
    if (c.kind == LR_SURFACE_DISNEY && c.x[2] == 0u && c.x[1] != 0u && (c.x[0] & 128u) != 0u) { eta = c.e[kDisneyEtaT]; return true; }
    return false;
}

}// namespace lrd
