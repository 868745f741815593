// oracle_bsdf.h — BxDF math and surface closures of the reference, restated on the CPU.
// TEST INFRASTRUCTURE ONLY (see oracle_math.h).
//
// Follows src/util/scattering.cpp:14-449 (PBRT-v3 style BxDFs in the local shading frame),
// src/base/surface.cpp:35-68 (side validation) and the closures
// src/surfaces/{matte,mirror,glass,plastic,metal}.cpp.  Spectrum = 3 fixed channels
// (src/spectra/srgb.cpp), TransportMode::RADIANCE only (megapath never uses IMPORTANCE).
#pragma once
#include "../include/lr_scene.h"
#include "oracle_rng.h"

namespace oracle {

using Spectrum3 = float3;

// ---------------------------------------------------------------- scattering.cpp

inline bool refract(float3 wi, float3 n, float eta, float3 &wt) {// :14-28
    auto cosThetaI = dot(n, wi);
    auto sin2ThetaI = std::max(0.0f, 1.f - sqr(cosThetaI));
    auto sin2ThetaT = sqr(eta) * sin2ThetaI;
    auto cosThetaT = std::sqrt(1.f - sin2ThetaT);
    wt = (eta * cosThetaI - cosThetaT) * n - eta * wi;
    return sin2ThetaT < 1.0f;
}

inline float fresnel_dielectric(float cosThetaI_in, float etaI_in, float etaT_in) {// :30-52
    auto cosThetaI = clampf(cosThetaI_in, -1.f, 1.f);
    auto entering = cosThetaI > 0.f;
    auto etaI = entering ? etaI_in : etaT_in;
    auto etaT = entering ? etaT_in : etaI_in;
    cosThetaI = std::abs(cosThetaI);
    auto sinThetaI = std::sqrt(std::max(0.f, 1.f - sqr(cosThetaI)));
    auto sinThetaT = etaI / etaT * sinThetaI;
    auto cosThetaT = std::sqrt(std::max(0.f, 1.f - sqr(sinThetaT)));
    auto Rparl = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    auto Rperp = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    auto fr = (Rparl * Rparl + Rperp * Rperp) * .5f;
    return sinThetaT < 1.f ? fr : 1.f;
}

inline Spectrum3 fresnel_conductor(float cosThetaI, float etai, Spectrum3 etat, Spectrum3 k) {// :54-74
    cosThetaI = clampf(cosThetaI, -1.f, 1.f);
    auto eta = etat / etai;
    auto etak = k / etai;
    auto cosThetaI2 = cosThetaI * cosThetaI;
    auto sinThetaI2 = 1.f - cosThetaI2;
    auto eta2 = eta * eta;
    auto etak2 = etak * etak;
    auto t0 = eta2 - etak2 - f3(sinThetaI2);
    auto a2plusb2 = sqrt3(t0 * t0 + 4.f * eta2 * etak2);
    auto t1 = a2plusb2 + f3(cosThetaI2);
    auto a = sqrt3(.5f * (a2plusb2 + t0));
    auto t2 = 2.f * cosThetaI * a;
    auto Rs = (t1 - t2) / (t1 + t2);
    auto t3 = cosThetaI2 * a2plusb2 + f3(sinThetaI2 * sinThetaI2);
    auto t4 = t2 * sinThetaI2;
    auto Rp = Rs * (t3 - t4) / (t3 + t4);
    return .5f * (Rp + Rs);
}

inline float fresnel_dielectric_integral(float eta) {// :97-107
    auto fit_less_one = [](float e) {// polynomial(x, c0, c1, c2, c3) = c0 + c1 x + c2 x^2 + c3 x^3 (Horner)
        return 0.75985009f + e * (-2.09069066f + e * (2.23559031f + e * -0.90663979f));
    };
    auto fit_greater_one = [](float e) {
        auto x = 1.f / e;
        return 0.97945724f + x * (0.21762732f + x * -1.18995376f);
    };
    return saturate(eta == 1.f ? 0.f : (eta < 1.f ? fit_less_one(eta) : fit_greater_one(eta)));
}

struct TrowbridgeReitz {// :117-237
    float2 alpha;
    explicit TrowbridgeReitz(float2 a) : alpha{std::max(a.x, 1e-4f), std::max(a.y, 1e-4f)} {}// :123-124
    static float roughness_to_alpha(float r) { return std::max(sqr(r), 1e-4f); }               // :129-135

    float D(float3 wh) const {// :145-156
        auto tan2Theta = tan2_theta(wh);
        auto cos4Theta = sqr(cos2_theta(wh));
        auto e = tan2Theta * (sqr(cos_phi(wh) / alpha.x) + sqr(sin_phi(wh) / alpha.y));
        auto d = 1.0f / (pi * alpha.x * alpha.y * cos4Theta * sqr(1.f + e));
        return std::isinf(tan2Theta) ? 0.f : d;
    }
    float Lambda(float3 w) const {// :158-170
        auto tanTheta = std::abs(tan_theta(w));
        auto alpha2 = cos2_phi(w) * sqr(alpha.x) + sin2_phi(w) * sqr(alpha.y);
        auto alpha2Tan2Theta = alpha2 * sqr(tanTheta);
        auto L = (-1.f + std::sqrt(1.f + alpha2Tan2Theta)) * .5f;
        return std::isinf(tanTheta) ? 0.f : L;
    }
    float G1(float3 w) const { return 1.0f / (1.0f + Lambda(w)); }                    // :109-111
    float G(float3 wo, float3 wi) const { return 1.0f / (1.0f + Lambda(wo) + Lambda(wi)); }// :113-115
    float pdf(float3 wo, float3 wh) const {                                            // :117-121
        return D(wh) * G1(wo) * abs_dot(wo, wh) / abs_cos_theta(wo);
    }
    static float2 sample11(float cosTheta, float2 U) {// :172-208
        if (cosTheta <= .9999f) {
            auto sinTheta = std::sqrt(std::max(0.f, 1.f - sqr(cosTheta)));
            auto tanTheta = sinTheta / cosTheta;
            auto a = 1.f / tanTheta;
            auto G1 = 2.f / (1.f + std::sqrt(1.f + 1.f / sqr(a)));
            auto A = 2.f * U.x / G1 - 1.f;
            auto tmp = std::min(1.f / (sqr(A) - 1.f), 1e10f);
            auto B = tanTheta;
            auto D = std::sqrt(std::max(sqr(B * tmp) - (sqr(A) - sqr(B)) * tmp, 0.f));
            auto slope_x_1 = B * tmp - D;
            auto slope_x_2 = B * tmp + D;
            auto slope_x = ((A < 0.f) || (slope_x_2 * tanTheta > 1.f)) ? slope_x_1 : slope_x_2;
            auto S = U.y > .5f ? 1.f : -1.f;
            auto U2 = U.y > .5f ? 2.f * (U.y - .5f) : 2.f * (.5f - U.y);
            auto z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                     (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
            auto slope_y = S * z * std::sqrt(1.f + sqr(slope_x));
            return {slope_x, slope_y};
        }
        auto r = std::sqrt(U.x / (1.f - U.x));
        auto phi = (2.f * pi) * U.y;
        return {r * std::cos(phi), r * std::sin(phi)};
    }
    static float3 sample(float3 wi, float2 alpha, float2 U) {// :210-231
        auto wiStretched = normalize(f3(alpha.x * wi.x, alpha.y * wi.y, wi.z));
        auto slope = sample11(cos_theta(wiStretched), U);
        slope = {cos_phi(wiStretched) * slope.x - sin_phi(wiStretched) * slope.y,
                 sin_phi(wiStretched) * slope.x + cos_phi(wiStretched) * slope.y};
        slope = {alpha.x * slope.x, alpha.y * slope.y};
        return normalize(f3(-slope.x, -slope.y, 1.f));
    }
    float3 sample_wh(float3 wo, float2 u) const {// :233-237
        auto s = sign(cos_theta(wo));
        auto wh = sample(s * wo, alpha, u);
        return s * wh;
    }
};

struct BxdfSample {
    Spectrum3 f{0.f, 0.f, 0.f};
    float3 wi{0.f, 0.f, 1.f};
    float pdf{0.f};
};

// default cosine-hemisphere lobe helpers, :256-264
inline float cosine_pdf(float3 wo, float3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * inv_pi : 0.f; }
inline float3 cosine_sample_wi(float3 wo, float2 u) {
    auto wi = sample_cosine_hemisphere(u);
    wi.z *= sign(cos_theta(wo));
    return wi;
}

inline Spectrum3 lambert_eval(Spectrum3 r, float3 wo, float3 wi) {// :266-269
    return r * (same_hemisphere(wo, wi) ? inv_pi : 0.f);
}

struct OrenNayar {// :370-400
    Spectrum3 r;
    float a, b;
    OrenNayar(Spectrum3 R, float sigma_deg) : r{R} {
        auto sigma = sigma_deg * (pi / 180.f);// radians(sigma)
        auto sigma2 = sqr(sigma);
        a = 1.f - (sigma2 / (2.f * sigma2 + 0.66f));
        b = 0.45f * sigma2 / (sigma2 + 0.09f);
    }
    Spectrum3 evaluate(float3 wo, float3 wi) const {
        auto s = same_hemisphere(wo, wi) ? inv_pi : 0.f;
        auto sinThetaI = sin_theta(wi);
        auto sinThetaO = sin_theta(wo);
        auto dCos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
        auto maxCos = (sinThetaI > 1e-4f && sinThetaO > 1e-4f) ? std::max(0.f, dCos) : 0.f;
        auto absCosThetaI = abs_cos_theta(wi);
        auto absCosThetaO = abs_cos_theta(wo);
        auto sinAlpha = absCosThetaI > absCosThetaO ? sinThetaO : sinThetaI;
        auto tanBeta = absCosThetaI > absCosThetaO ? sinThetaI / absCosThetaI : sinThetaO / absCosThetaO;
        return s * (a + b * maxCos * sinAlpha * tanBeta) * r;
    }
};

// MicrofacetReflection, :286-320; `fresnel(cos)` returns a Spectrum3
template<typename F>
inline Spectrum3 microfacet_reflection_eval(Spectrum3 R, const TrowbridgeReitz &dist, F &&fresnel, float3 wo, float3 wi) {
    auto wh = wi + wo;
    Spectrum3 f{0.f, 0.f, 0.f};
    if (same_hemisphere(wo, wi) && (wh.x != 0.f || wh.y != 0.f || wh.z != 0.f)) {
        wh = normalize(wh);
        auto Fr = fresnel(dot(wi, face_forward(wh, f3(0.f, 0.f, 1.f))));
        auto D = dist.D(wh);
        auto G = dist.G(wo, wi);
        f = R * Fr * std::abs(0.25f * D * G / (cos_theta(wi) * cos_theta(wo)));
    }
    return f;
}
inline float microfacet_reflection_pdf(const TrowbridgeReitz &dist, float3 wo, float3 wi) {
    auto p = 0.f;
    auto wh = wi + wo;
    if (same_hemisphere(wo, wi) && (wh.x != 0.f || wh.y != 0.f || wh.z != 0.f)) {
        wh = normalize(wh);
        p = dist.pdf(wo, wh) / (4.f * dot(wo, wh));
    }
    return p;
}
template<typename F>
inline BxdfSample microfacet_reflection_sample(Spectrum3 R, const TrowbridgeReitz &dist, F &&fresnel, float3 wo, float2 u) {
    BxdfSample s;// BxDF::sample, :247-254
    auto wh = dist.sample_wh(wo, u);
    s.wi = reflect(-wo, wh);
    auto valid = same_hemisphere(wo, s.wi);
    s.pdf = valid ? microfacet_reflection_pdf(dist, wo, s.wi) : 0.f;
    s.f = valid ? microfacet_reflection_eval(R, dist, fresnel, wo, s.wi) : f3(0.f);
    return s;
}

// MicrofacetTransmission (RADIANCE), :322-368
inline Spectrum3 microfacet_transmission_eval(Spectrum3 T, const TrowbridgeReitz &dist, float eta_a, float eta_b, float3 wo, float3 wi,
                                              bool importance = false) {
    auto cosThetaO = cos_theta(wo);
    auto cosThetaI = cos_theta(wi);
    auto eta = cosThetaO > 0.f ? eta_b / eta_a : eta_a / eta_b;
    auto wh = normalize(wo + wi * eta);
    wh = sign(cos_theta(wh)) * wh;
    Spectrum3 f{0.f, 0.f, 0.f};
    if (!same_hemisphere(wo, wi) && cosThetaO != 0.f && cosThetaI != 0.f && dot(wo, wh) * dot(wi, wh) < 0.f) {
        auto G = dist.G(wo, wi);
        auto sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        auto F = fresnel_dielectric(dot(wo, wh), eta_a, eta_b);
        auto D = dist.D(wh);
        f = (1.f - F) * T * D * G * dot(wi, wh) * dot(wo, wh) / (cosThetaI * cosThetaO * sqr(sqrtDenom));
        if (importance) { f = f * sqr(eta); }// TransportMode::IMPORTANCE, scattering.cpp:340-342 (Layered only)
    }
    return f;
}
inline float microfacet_transmission_pdf(const TrowbridgeReitz &dist, float eta_a, float eta_b, float3 wo, float3 wi) {
    auto pdf = 0.f;
    auto entering = cos_theta(wo) > 0.f;
    auto eta = entering ? eta_b / eta_a : eta_a / eta_b;
    auto wh = normalize(wo + wi * eta);
    if (!same_hemisphere(wo, wi) && dot(wo, wh) * dot(wi, wh) < 0.f) {
        auto sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        auto dwh_dwi = sqr(eta / sqrtDenom) * abs_dot(wi, wh);
        pdf = dist.pdf(wo, wh) * dwh_dwi;
    }
    return pdf;
}
inline BxdfSample microfacet_transmission_sample(Spectrum3 T, const TrowbridgeReitz &dist, float eta_a, float eta_b, float3 wo, float2 u,
                                                 bool importance = false) {
    BxdfSample s;
    auto eta = cos_theta(wo) > 0.f ? eta_a / eta_b : eta_b / eta_a;
    auto wh = dist.sample_wh(wo, u);
    float3 wi{0.f, 0.f, 0.f};
    auto refr = refract(wo, wh, eta, wi);
    auto valid = refr && !same_hemisphere(wo, wi);
    s.wi = wi;
    s.pdf = valid ? microfacet_transmission_pdf(dist, eta_a, eta_b, wo, wi) : 0.f;
    s.f = valid ? microfacet_transmission_eval(T, dist, eta_a, eta_b, wo, wi, importance) : f3(0.f);
    return s;
}

// ---------------------------------------------------------------- interaction + textures

struct Interaction {// src/base/interaction.h
    lr_uint4 handle{};
    uint32_t inst{LR_INVALID_ID}, prim{LR_INVALID_ID};
    float3 pg, ng, ps;
    float2 uv;
    Frame shading;
    float area{0.f};
    bool back_facing{false};
    bool valid() const { return inst != LR_INVALID_ID; }
    uint32_t flags() const { return handle.x & 1023u; }
    bool has_light() const { return flags() & LR_SHAPE_HAS_LIGHT; }
    bool has_surface() const { return flags() & LR_SHAPE_HAS_SURFACE; }
    uint32_t light_tag() const { return handle.y & 4095u; }
    uint32_t surface_tag() const { return (handle.y >> 12u) & 4095u; }
    uint32_t mesh_index() const { return handle.x >> 10u; }
    float intersection_offset_factor() const {// Shape::Handle::decode, shape.cpp:88-93
        // An interaction without a shape (Interaction{p} of a medium point, interaction.h:77-78) keeps the DEFAULT-constructed
        // Shape::Handle, whose DSL members are zero: factor 0, p_robust() returns p itself (pinned by oracle/_ref: the reference's
        // own code gives exactly that; round 1 had decoded a zero handle word to factor 1 here).
        if (!valid()) { return 0.f; }
        auto x = static_cast<float>(handle.w & 0xffffu) * (1.0f / 65536.f);
        return clampf(x * 255.f + 1.f, 1.f, 256.f);
    }
};

inline float4 texel_fetch(const lr_scene &scene, const lr_texture &t, int x, int y) {
    auto wrap = [&](int v, int n, bool &zero) {
        switch (t.address) {
            case LR_TEX_ADDR_EDGE: return std::min(std::max(v, 0), n - 1);
            case LR_TEX_ADDR_MIRROR: {
                auto period = 2 * n;
                auto m = ((v % period) + period) % period;
                return m < n ? m : period - 1 - m;
            }
            case LR_TEX_ADDR_ZERO:
                if (v < 0 || v >= n) { zero = true; return 0; }
                return v;
            default: return ((v % n) + n) % n;
        }
    };
    auto zero = false;
    auto xx = wrap(x, static_cast<int>(t.width), zero), yy = wrap(y, static_cast<int>(t.height), zero);
    if (zero) { return {0.f, 0.f, 0.f, 0.f}; }
    auto p = scene.texels + (t.texel_offset + static_cast<uint64_t>(yy) * t.width + static_cast<uint64_t>(xx)) * 4u;
    return {p[0], p[1], p[2], p[3]};
}

// Texture::Instance::evaluate: constant.cpp:73-79, image.cpp:132-168 (bilinear, LOD 0)
inline float4 texture_evaluate(const lr_scene &scene, int32_t id, float2 uv_it) {
    auto &t = scene.textures[id];
    if (t.kind == LR_TEX_CONSTANT) { return {t.v[0], t.v[1], t.v[2], t.v[3]}; }
    if (t.kind == LR_TEX_IMAGE) {
        float2 uv{uv_it.x * t.uv_scale[0] + t.uv_offset[0], uv_it.y * t.uv_scale[1] + t.uv_offset[1]};
        float4 v;
        if (t.filter == LR_TEX_FILTER_POINT) {
            v = texel_fetch(scene, t, static_cast<int>(std::floor(uv.x * static_cast<float>(t.width))),
                            static_cast<int>(std::floor(uv.y * static_cast<float>(t.height))));
        } else {
            auto fx = uv.x * static_cast<float>(t.width) - 0.5f, fy = uv.y * static_cast<float>(t.height) - 0.5f;
            auto x0 = std::floor(fx), y0 = std::floor(fy);
            auto tx = fx - x0, ty = fy - y0;
            auto ix = static_cast<int>(x0), iy = static_cast<int>(y0);
            auto c00 = texel_fetch(scene, t, ix, iy), c10 = texel_fetch(scene, t, ix + 1, iy);
            auto c01 = texel_fetch(scene, t, ix, iy + 1), c11 = texel_fetch(scene, t, ix + 1, iy + 1);
            auto mix = [&](float a, float b, float c, float d) {
                return (a * (1.f - tx) + b * tx) * (1.f - ty) + (c * (1.f - tx) + d * tx) * ty;
            };
            v = {mix(c00.x, c10.x, c01.x, c11.x), mix(c00.y, c10.y, c01.y, c11.y),
                 mix(c00.z, c10.z, c01.z, c11.z), mix(c00.w, c10.w, c01.w, c11.w)};
        }
        auto decode = [&](float c, int ch) {// image.cpp:138-153
            if (t.encoding == LR_TEX_ENC_SRGB) {
                c = c <= 0.04045f ? c * (1.0f / 12.92f) : std::pow((c + 0.055f) * (1.0f / 1.055f), 2.4f);
            } else if (t.encoding == LR_TEX_ENC_GAMMA) {
                c = std::pow(c, t.gamma[std::min(ch, 2)]);
            }
            return t.scale[ch] * c;
        };
        return {decode(v.x, 0), decode(v.y, 1), decode(v.z, 2), decode(v.w, 3)};
    }
    // checkerboard (src/textures/checkerboard.cpp): on/off by parity of floor(uv * scale)
    float2 uv{uv_it.x * t.checker_scale, uv_it.y * t.checker_scale};
    auto parity = (static_cast<int>(std::floor(uv.x)) + static_cast<int>(std::floor(uv.y))) & 1;
    auto child = t.child[parity ? 1 : 0];
    if (child < 0) { return parity ? float4{0.f, 0.f, 0.f, 1.f} : float4{1.f, 1.f, 1.f, 1.f}; }
    return texture_evaluate(scene, child, uv_it);
}

inline float3 extend_color_to_rgb(float4 c, uint32_t n) {// texture.cpp:14-18
    if (n == 1u) { return {c.x, c.x, c.x}; }
    if (n == 2u) { return {c.x, c.y, 1.f}; }
    return {c.x, c.y, c.z};
}

struct Decode {
    Spectrum3 value;
    float strength;
};
// evaluate_albedo_spectrum with the sRGB spectrum (texture.cpp:21-33, srgb.cpp:34-41)
inline Decode albedo_or(const lr_scene &scene, int32_t id, float2 uv, float default_value) {
    if (id < 0) { return {f3(default_value), default_value}; }// Spectrum::Decode::constant
    auto sv = saturate(extend_color_to_rgb(texture_evaluate(scene, id, uv), scene.textures[id].channels));
    return {sv, srgb_to_cie_y(sv)};
}
// evaluate_illuminant_spectrum (texture.cpp:49-60, srgb.cpp:48-54): static (constant) textures are
// channel-extended, dynamic ones use xyz as-is
inline Decode illuminant(const lr_scene &scene, int32_t id, float2 uv) {
    auto &t = scene.textures[id];
    auto v = texture_evaluate(scene, id, uv);
    auto rgb = t.kind == LR_TEX_CONSTANT ? extend_color_to_rgb(v, t.channels) : f3(v.x, v.y, v.z);
    auto sv = max0(rgb);
    return {sv, srgb_to_cie_y(sv)};
}

// ---------------------------------------------------------------- closures

enum : uint32_t { EVENT_REFLECT = 0u, EVENT_ENTER = 1u, EVENT_EXIT = 2u };// surface.h:46-50

struct SurfaceEval {
    Spectrum3 f{0.f, 0.f, 0.f};
    float pdf{0.f};
};
struct SurfaceSample {
    SurfaceEval eval;
    float3 wi{0.f, 0.f, 1.f};
    uint32_t event{EVENT_REFLECT};
};

// validate_surface_sides, surface.cpp:35-43
inline bool validate_surface_sides(float3 ng, float3 ns, float3 wo, float3 wi) {
    auto flip = sign(dot(ng, ns));
    return sign(flip * dot(wo, ns)) == sign(dot(wo, ng)) && sign(flip * dot(wi, ns)) == sign(dot(wi, ng));
}


// ---------------------------------------------------------------- Disney (src/surfaces/disney.cpp)

inline float schlick_weight(float cosTheta) {// :95-98
    auto m = saturate(1.f - cosTheta);
    return sqr(sqr(m)) * m;
}
inline float fr_schlick(float R0, float cosTheta) { return lerp(R0, 1.f, schlick_weight(cosTheta)); }// :100-102
inline float schlick_r0_from_eta(float eta) { return sqr((eta - 1.f) / (eta + 1.f)); }             // :106-108
inline float gtr1(float cosTheta, float alpha) {// :210-214
    auto alpha2 = sqr(alpha);
    auto denom = pi * std::log(alpha2) * (1.f + (alpha2 - 1.f) * sqr(cosTheta));
    return (alpha2 - 1.f) / denom;
}
inline float smith_g_ggx(float cosTheta, float alpha) {// :217-221
    auto alpha2 = sqr(alpha);
    auto cosTheta2 = sqr(cosTheta);
    return 1.f / (cosTheta + std::sqrt(alpha2 + cosTheta2 - alpha2 * cosTheta2));
}

enum : uint32_t {
    DISNEY_LOBE_DIFFUSE = 1u << 0u, DISNEY_LOBE_RETRO = 1u << 1u, DISNEY_LOBE_FAKE_SS = 1u << 2u, DISNEY_LOBE_SHEEN = 1u << 3u,
    DISNEY_LOBE_CLEARCOAT = 1u << 4u, DISNEY_LOBE_SPECULAR = 1u << 5u, DISNEY_LOBE_DIFF_TRANS = 1u << 6u, DISNEY_LOBE_SPEC_TRANS = 1u << 7u
};// :323-330

struct DisneyParams {// DisneyContext, :304-321
    Spectrum3 color;
    float color_lum, metallic, eta_i, eta_t, roughness, specular_tint, anisotropic, sheen, sheen_tint;
    float clearcoat, clearcoat_gloss, specular_trans, flatness, diffuse_trans;
    uint32_t lobes;
    bool thin, transmissive;
};

// DisneyClosureImpl (:351-588) and ThinDisneyClosureImpl (:590-841) on one structure
struct DisneyClosure {
    bool thin{false}, transmissive{false};
    bool has_diffuse{false}, has_fake_ss{false}, has_sheen{false}, has_clearcoat{false}, has_spec_trans{false}, has_diff_trans{false};
    Spectrum3 Cdiff, Css, Csheen, Cspec0, Cst, Cdt;
    float roughness{0.f}, metallic{0.f}, eta{1.f}, eta_i{1.f}, eta_t{1.f}, clearcoat{0.f}, gloss{0.f};
    float2 alpha{0.f, 0.f}, thin_alpha{0.f, 0.f};
    bool two_sided_fresnel{false};
    bool importance{false};// TransportMode::IMPORTANCE (set by Layered's reverse walks)
    float w[5]{0.f, 0.f, 0.f, 0.f, 0.f};
    bool enabled[5]{false, false, false, false, false};
    uint32_t technique_count{0u};

    explicit DisneyClosure(const DisneyParams &c) {
        thin = c.thin, transmissive = c.transmissive;
        auto diffuse_weight = (1.f - c.metallic) * (1.f - c.specular_trans);
        auto diff_refl_weight = thin ? diffuse_weight * (1.f - c.diffuse_trans) : diffuse_weight;
        auto diff_trans_weight = diffuse_weight * c.diffuse_trans;
        auto tint_weight = c.color_lum > 0.f ? 1.f / c.color_lum : 1.f;
        auto tint = saturate(c.color * tint_weight);
        auto tint_lum = c.color_lum * tint_weight;
        roughness = c.roughness, metallic = c.metallic;
        auto diffuse_like = diff_refl_weight * c.color_lum;
        if (c.lobes & (DISNEY_LOBE_DIFFUSE | DISNEY_LOBE_RETRO)) {
            Cdiff = c.color * (diff_refl_weight * (1.f - c.flatness));
            has_diffuse = true, enabled[0] = true;
        }
        if (c.lobes & DISNEY_LOBE_FAKE_SS) {
            auto Css_weight = thin ? diff_refl_weight * c.flatness * (1.f - c.diffuse_trans) : diffuse_weight * c.flatness;
            Css = Css_weight * c.color;
            has_fake_ss = true, enabled[0] = true;
        }
        if (c.lobes & DISNEY_LOBE_SHEEN) {
            auto Csheen_weight = thin ? diff_refl_weight * c.sheen * (1.f - c.diffuse_trans) : diffuse_weight * c.sheen;
            Csheen = Csheen_weight * lerp(f3(1.f), tint, c.sheen_tint);
            has_sheen = true;
            diffuse_like += Csheen_weight * lerp(1.f, tint_lum, c.sheen_tint) * .1f;
            if (!thin) { enabled[0] = true; }// the thin closure does not enable the technique for sheen alone (:641-647)
        }
        w[0] = saturate(diffuse_like);
        eta_i = c.eta_i, eta_t = c.eta_t;
        eta = c.eta_t / c.eta_i;
        auto R0 = schlick_r0_from_eta(eta);
        Cspec0 = lerp(lerp(f3(1.f), tint, c.specular_tint) * R0, c.color, c.metallic);
        two_sided_fresnel = thin ? false : !transmissive;
        auto aspect = std::sqrt(1.f - c.anisotropic * .9f);
        alpha = {std::max(0.001f, c.roughness / aspect), std::max(0.001f, c.roughness * aspect)};
        w[1] = saturate(lerp(lerp(1.f, tint_lum, c.specular_tint) * R0, c.color_lum, c.metallic));
        enabled[1] = true;
        if (c.lobes & DISNEY_LOBE_CLEARCOAT) {
            gloss = lerp(.1f, .001f, c.clearcoat_gloss);
            clearcoat = c.clearcoat;
            has_clearcoat = true;
            w[2] = saturate(c.clearcoat * fr_schlick(.04f, 1.f));
            enabled[2] = true;
        }
        if (thin) {
            technique_count = 5u;
            if (c.lobes & DISNEY_LOBE_SPEC_TRANS) {
                auto rscaled = (.65f * eta - .35f) * c.roughness;
                thin_alpha = {std::max(.001f, rscaled / aspect), std::max(.001f, rscaled * aspect)};
                auto Cst_weight = (1.f - c.metallic) * c.specular_trans;
                Cst = Cst_weight * c.color;
                has_spec_trans = true;
                w[3] = saturate(Cst_weight * c.color_lum);
                enabled[3] = true;
            }
            if (c.lobes & DISNEY_LOBE_DIFF_TRANS) {
                Cdt = diff_trans_weight * c.color;
                has_diff_trans = true;
                w[4] = saturate(diff_trans_weight * c.color_lum);
                enabled[4] = true;
            }
        } else {
            technique_count = transmissive ? 4u : 3u;
            if (transmissive && (c.lobes & DISNEY_LOBE_SPEC_TRANS)) {
                auto Cst_weight = (1.f - c.metallic) * c.specular_trans;
                Cst = Cst_weight * sqrt3(c.color);
                thin_alpha = alpha;
                has_spec_trans = true;
                w[3] = saturate(Cst_weight * std::sqrt(c.color_lum));
                enabled[3] = true;
            }
        }
        auto sum = 0.f;
        for (auto i = 0u; i < technique_count; i++) {
            if (enabled[i]) { sum += w[i]; }
        }
        auto inv = sum == 0.f ? 0.f : 1.f / sum;
        for (auto i = 0u; i < technique_count; i++) {
            if (enabled[i]) { w[i] *= inv; }
        }
    }

    Spectrum3 disney_fresnel(float cosI_in) const {// DisneyFresnel::evaluate, :277-296
        auto cosI = two_sided_fresnel ? std::abs(cosI_in) : cosI_in;
        auto fr = fresnel_dielectric(cosI, 1.f, eta);
        auto f0 = f3(fr_schlick(Cspec0.x, cosI), fr_schlick(Cspec0.y, cosI), fr_schlick(Cspec0.z, cosI));
        return lerp(f3(fr), f0, metallic);
    }
    float clearcoat_eval(float3 wo, float3 wi) const {// :232-249
        auto wh = wi + wo;
        auto valid = wh.x != 0.f || wh.y != 0.f || wh.z != 0.f;
        wh = normalize(wh);
        auto Dr = gtr1(abs_cos_theta(wh), gloss);
        auto Fr = fr_schlick(.04f, dot(wo, wh));
        auto Gr = smith_g_ggx(abs_cos_theta(wo), .25f) * smith_g_ggx(abs_cos_theta(wi), .25f);
        return valid ? clearcoat * Gr * Fr * Dr * .25f : 0.f;
    }
    float clearcoat_pdf(float3 wo, float3 wi) const {// :266-279
        auto wh = wi + wo;
        auto valid = same_hemisphere(wo, wi) && (wh.x != 0.f || wh.y != 0.f || wh.z != 0.f);
        wh = normalize(wh);
        auto Dr = gtr1(abs_cos_theta(wh), gloss);
        return valid ? Dr * abs_cos_theta(wh) / (4.f * dot(wo, wh)) : 0.f;
    }
    SurfaceEval evaluate_local(float3 wo, float3 wi) const {// _evaluate_local, :476-521 / :728-781
        Spectrum3 f{0.f, 0.f, 0.f};
        auto pdf = 0.f;
        if (same_hemisphere(wo, wi)) {
            if (has_diffuse && w[0] > 0.f) {
                auto Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
                f += Cdiff * (inv_pi * (1.f - Fo * .5f) * (1.f - Fi * .5f));// DisneyDiffuse, :118-126
                auto wh = wi + wo;
                auto valid = wh.x != 0.f || wh.y != 0.f || wh.z != 0.f;
                wh = normalize(wh);
                auto cosThetaD = dot(wi, wh);
                auto Rr = 2.f * roughness * cosThetaD * cosThetaD;// DisneyRetro, :173-186
                f += Cdiff * (valid ? inv_pi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1.f)) : 0.f);
                if (has_fake_ss) {// DisneyFakeSS, :143-160
                    auto Fss90 = cosThetaD * cosThetaD * roughness;
                    auto Fss = lerp(1.0f, Fss90, Fo) * lerp(1.0f, Fss90, Fi);
                    auto ss = 1.25f * (Fss * (1.f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - .5f) + .5f);
                    f += Css * (valid ? inv_pi * ss : 0.f);
                }
                if (has_sheen) { f += Csheen * (valid ? schlick_weight(cosThetaD) : 0.f); }// DisneySheen, :198-207
                pdf += w[0] * cosine_pdf(wo, wi);
            }
            if (w[1] > 0.f) {
                TrowbridgeReitz dist{alpha};
                f += microfacet_reflection_eval(f3(1.f), dist, [&](float c) { return disney_fresnel(c); }, wo, wi);
                pdf += w[1] * microfacet_reflection_pdf(dist, wo, wi);
            }
            if (has_clearcoat && w[2] > 0.f) {
                f += f3(clearcoat_eval(wo, wi));
                pdf += w[2] * clearcoat_pdf(wo, wi);
            }
        } else {
            if (has_spec_trans && w[3] > 0.f) {
                TrowbridgeReitz dist{thin_alpha};
                f += microfacet_transmission_eval(Cst, dist, eta_i, eta_t, wo, wi, importance);
                pdf += w[3] * microfacet_transmission_pdf(dist, eta_i, eta_t, wo, wi);
            }
            if (has_diff_trans && w[4] > 0.f) {// LambertianTransmission, scattering.cpp:271-284
                f += Cdt * (!same_hemisphere(wo, wi) ? inv_pi : 0.f);
                pdf += w[4] * (same_hemisphere(wo, wi) ? 0.0f : abs_cos_theta(wi) * inv_pi);
            }
        }
        return {f * abs_cos_theta(wi), pdf};
    }
    // sample, :538-587 / :796-840.  Returns local wi, validity and the event
    void sample_local(float3 wo, float u_lobe, float2 u, float3 &wi, bool &valid, uint32_t &event) const {
        auto tech = 0u;
        auto sum = 0.f;
        for (auto i = 0u; i < technique_count; i++) {
            if (enabled[i]) {
                tech = u_lobe > sum ? i : tech;
                sum += w[i];
            }
        }
        event = EVENT_REFLECT;
        valid = false;
        wi = f3(0.f);
        if (tech == 0u && has_diffuse) {
            wi = cosine_sample_wi(wo, u), valid = true;
        } else if (tech == 1u) {
            TrowbridgeReitz dist{alpha};
            wi = reflect(-wo, dist.sample_wh(wo, u));
            valid = same_hemisphere(wo, wi);
        } else if (tech == 2u && has_clearcoat) {// DisneyClearcoat::sample_wi, :250-265
            auto alpha2 = gloss * gloss;
            auto cosTheta = std::sqrt(std::max(0.f, (1.f - std::pow(alpha2, 1.f - u.x)) / (1.f - alpha2)));
            auto sinTheta = std::sqrt(std::max(0.f, 1.f - cosTheta * cosTheta));
            auto phi = 2.f * pi * u.y;
            auto wh = f3(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta);
            wh = same_hemisphere(wo, wh) ? wh : -wh;
            wi = reflect(-wo, wh);
            valid = same_hemisphere(wo, wi);
        } else if (tech == 3u && has_spec_trans) {
            TrowbridgeReitz dist{thin_alpha};
            auto e = cos_theta(wo) > 0.f ? eta_i / eta_t : eta_t / eta_i;
            auto refr = refract(wo, dist.sample_wh(wo, u), e, wi);
            valid = refr && !same_hemisphere(wo, wi);
            event = thin ? 4u : (cos_theta(wo) > 0.f ? EVENT_ENTER : EVENT_EXIT);
        } else if (tech == 4u && has_diff_trans) {// LambertianTransmission::sample_wi
            wi = sample_cosine_hemisphere(u);
            wi.z *= -sign(cos_theta(wo));
            valid = true;
            event = 4u;// Surface::event_through
        }
    }
};

struct Closure {
    uint32_t kind{LR_SURFACE_NULL};
    float3 ng;
    Frame shading;
    // parameters (meaning per kind)
    Spectrum3 c0{1.f, 1.f, 1.f}, c1{1.f, 1.f, 1.f}, c2{0.f, 0.f, 0.f};
    float2 alpha{0.f, 0.f};
    float s0{0.f}, s1{0.f}, s2{0.f};
    bool has_eta{false};
    float eta_value{1.f};
    DisneyParams disney{};
    // Mix (src/surfaces/mix.cpp): children are full closures with their own (possibly normal-mapped) frames
    const lr_scene *mix_scene{nullptr};
    Interaction mix_it{};
    float3 mix_wo{};
    uint32_t mix_a{0u}, mix_b{0u};
    float mix_eta_i{1.f};
    // Layered (src/surfaces/layered.cpp): mix_a = top, mix_b = bottom, s0 = thickness, s1 = g, c0 = albedo
    uint32_t layer_max_depth{10u}, layer_samples{1u};
    bool importance{false};// transport mode of this evaluation (IMPORTANCE only inside Layered)

    // resolve the `roughness` texture like every closure does (e.g. mirror.cpp:145-154)
    static float2 roughness_alpha(const lr_scene &scene, const lr_surface &s, int32_t tex, float2 uv, float2 dv) {
        if (tex < 0) { return dv; }
        auto r = texture_evaluate(scene, tex, uv);
        auto remap = (s.flags & LR_SURFACE_FLAG_REMAP_ROUGHNESS) != 0u;
        auto r2a = [](float x) { return TrowbridgeReitz::roughness_to_alpha(x); };
        if (scene.textures[tex].channels == 1u) { return remap ? float2{r2a(r.x), r2a(r.x)} : float2{r.x, r.x}; }
        return remap ? float2{r2a(r.x), r2a(r.y)} : float2{r.x, r.y};
    }

    // Surface::Instance::closure -> populate_closure (+ NormalMapWrapper, surface.h:236-254)
    static Closure populate(const lr_scene &scene, const Interaction &it_in, float3 wo, float eta_i) {
        return populate_tag(scene, it_in.surface_tag(), it_in, wo, eta_i);
    }
    static Closure populate_tag(const lr_scene &scene, uint32_t tag, const Interaction &it_in, float3 wo, float eta_i) {
        auto &s = scene.surfaces[tag];
        auto it = it_in;
        if (s.normal_tex >= 0) {
            auto v = texture_evaluate(scene, s.normal_tex, it.uv);
            auto n_local = f3(2.f * v.x - 1.f, 2.f * v.y - 1.f, 2.f * v.z - 1.f);
            if (s.normal_strength != 1.f) { n_local = n_local * f3(s.normal_strength, s.normal_strength, 1.f); }
            auto normal = it.shading.local_to_world(n_local);
            it.shading = Frame::make(clamp_shading_normal(normal, it.ng, wo), it.shading.s);
        }
        Closure c;
        c.kind = s.kind;
        c.ng = it.ng;
        c.shading = it.shading;
        auto uv = it.uv;
        switch (s.kind) {
            case LR_SURFACE_MATTE: {// matte.cpp:119-134
                c.c0 = albedo_or(scene, s.tex[0], uv, 1.f).value;
                auto has_sigma = s.tex[1] >= 0 && !(scene.textures[s.tex[1]].kind == LR_TEX_CONSTANT &&
                                                    scene.textures[s.tex[1]].v[0] == 0.f && scene.textures[s.tex[1]].v[1] == 0.f &&
                                                    scene.textures[s.tex[1]].v[2] == 0.f && scene.textures[s.tex[1]].v[3] == 0.f);
                c.s0 = has_sigma ? saturate(texture_evaluate(scene, s.tex[1], uv).x) * 90.f : 0.f;
                break;
            }
            case LR_SURFACE_MIRROR: {// mirror.cpp:141-163
                c.alpha = roughness_alpha(scene, s, s.tex[1], uv, {0.f, 0.f});
                c.c0 = albedo_or(scene, s.tex[0], uv, 1.f).value;
                break;
            }
            case LR_SURFACE_GLASS: {// glass.cpp:232-285 (fixed spectrum: eta = first channel, non-dispersive)
                c.alpha = roughness_alpha(scene, s, s.tex[2], uv, {0.f, 0.f});
                auto kr = albedo_or(scene, s.tex[0], uv, 1.f), kt = albedo_or(scene, s.tex[1], uv, 1.f);
                c.c0 = kr.value, c.c1 = kt.value;
                c.s2 = kr.strength == 0.f ? 0.f : kr.strength / (kr.strength + kt.strength);// Kr_ratio
                c.s0 = eta_i;
                c.s1 = s.tex[3] >= 0 ? texture_evaluate(scene, s.tex[3], uv).x : 1.5f;
                c.has_eta = true, c.eta_value = c.s1;
                break;
            }
            case LR_SURFACE_PLASTIC: {// plastic.cpp:256-291
                c.alpha = roughness_alpha(scene, s, s.tex[1], uv, {0.f, 0.f});
                auto eta = (s.tex[3] >= 0 ? texture_evaluate(scene, s.tex[3], uv).x : 1.5f) / eta_i;
                auto kd = albedo_or(scene, s.tex[0], uv, 1.f);
                auto sigma_a = albedo_or(scene, s.tex[2], uv, 0.f);
                auto thickness = s.tex[4] >= 0 ? texture_evaluate(scene, s.tex[4], uv).x : 1.f;
                auto average_transmittance = std::exp(-2.f * sigma_a.strength * thickness);
                auto diffuse_fresnel = fresnel_dielectric_integral(eta);
                c.c0 = kd.value / (f3(1.f) - kd.value * diffuse_fresnel);
                c.s0 = kd.strength * average_transmittance;// Kd_weight
                c.c1 = sigma_a.value;                       // note: un-scaled sigma_a is bound (plastic.cpp:285)
                c.s1 = eta;
                break;
            }
            case LR_SURFACE_METAL: {// metal.cpp:273-308
                c.alpha = roughness_alpha(scene, s, s.tex[1], uv, {.5f, .5f});
                c.c0 = f3(s.f[0], s.f[1], s.f[2]);// n
                c.c1 = f3(s.f[3], s.f[4], s.f[5]);// k
                c.c2 = s.tex[0] >= 0 ? albedo_or(scene, s.tex[0], uv, 1.f).value : f3(1.f);
                c.s0 = eta_i;
                break;
            }
            case LR_SURFACE_DISNEY: {// disney.cpp:932-1003
                auto &d = c.disney;
                auto color = albedo_or(scene, s.tex[0], uv, 1.f);
                d.color = color.value, d.color_lum = color.strength;
                auto scalar = [&](int slot, float dv) { return s.tex[slot] >= 0 ? texture_evaluate(scene, s.tex[slot], uv).x : dv; };
                d.metallic = scalar(1, 0.f);
                d.eta_i = eta_i, d.eta_t = scalar(2, 1.5f);
                d.roughness = scalar(3, .5f);
                if (s.flags & LR_SURFACE_FLAG_REMAP_ROUGHNESS) { d.roughness = TrowbridgeReitz::roughness_to_alpha(d.roughness); }
                d.specular_tint = scalar(4, 0.f), d.anisotropic = scalar(5, 0.f), d.sheen = scalar(6, 0.f), d.sheen_tint = scalar(7, 0.f);
                d.clearcoat = scalar(8, 0.f), d.clearcoat_gloss = scalar(9, 1.f), d.specular_trans = scalar(10, 0.f);
                d.flatness = scalar(11, 0.f), d.diffuse_trans = scalar(12, 0.f);
                d.lobes = s.u[0];// union over the scene's Disney surfaces of the same kind (enable_lobes, :856)
                d.thin = (s.flags & LR_SURFACE_FLAG_THIN) != 0u;
                d.transmissive = s.u[1] != 0u;
                // DisneyClosureImpl::eta (:531-533): only the thick closure with a specular-transmission lobe reports one
                c.has_eta = !d.thin && d.transmissive && (d.lobes & DISNEY_LOBE_SPEC_TRANS) != 0u;
                c.eta_value = d.eta_t;
                break;
            }
            case LR_SURFACE_MIX: {// mix.cpp:198-212
                c.s0 = s.tex[0] >= 0 ? clampf(texture_evaluate(scene, s.tex[0], uv).x, 0.f, 1.f) : 0.5f;
                c.mix_scene = &scene, c.mix_it = it, c.mix_wo = wo, c.mix_a = s.u[0], c.mix_b = s.u[1], c.mix_eta_i = eta_i;
                auto a = populate_tag(scene, s.u[0], it, wo, eta_i), b = populate_tag(scene, s.u[1], it, wo, eta_i);
                if (!a.has_eta) { c.has_eta = b.has_eta, c.eta_value = b.eta_value; }// mix.cpp:148-157
                else if (!b.has_eta) { c.has_eta = true, c.eta_value = a.eta_value; }
                else { c.has_eta = true, c.eta_value = lerp(b.eta_value, a.eta_value, c.s0); }
                break;
            }
            case LR_SURFACE_LAYERED: {// layered.cpp:478-500
                c.s0 = s.tex[0] >= 0 ? std::max(texture_evaluate(scene, s.tex[0], uv).x, std::numeric_limits<float>::min()) : 1e-2f;
                c.s1 = s.tex[1] >= 0 ? texture_evaluate(scene, s.tex[1], uv).x : 0.f;
                c.c0 = albedo_or(scene, s.tex[2], uv, 1.f).value;
                c.layer_max_depth = s.u[2], c.layer_samples = s.u[3];
                c.mix_scene = &scene, c.mix_it = it, c.mix_wo = wo, c.mix_a = s.u[0], c.mix_b = s.u[1], c.mix_eta_i = eta_i;
                auto top = populate_tag(scene, s.u[0], it, wo, eta_i);
                auto bottom = populate_tag(scene, s.u[1], it, wo, top.has_eta ? top.eta_value : 1.f);
                c.has_eta = bottom.has_eta, c.eta_value = bottom.eta_value;// LayeredSurfaceClosure::eta, :252
                break;
            }
            default: break;
        }
        return c;
    }

    // ---- Layered (layered.cpp:195-470, a port of PBRT-v4's LayeredBxDF onto nested closures).  The internal random
    // walk draws from an LCG seeded by a hash of position / direction BITS; where the reference evaluates several
    // lcg(seed) calls inside one argument list (C++ leaves their order unspecified) they are taken left to right.
    std::vector<Closure> layers() const {// {top, bottom}
        std::vector<Closure> l;
        l.emplace_back(populate_tag(*mix_scene, mix_a, mix_it, mix_wo, mix_eta_i));
        l.emplace_back(populate_tag(*mix_scene, mix_b, mix_it, mix_wo, l[0].has_eta ? l[0].eta_value : 1.f));
        return l;
    }
    static float layer_tr(float dz, float3 w) {// :214-217
        return std::abs(dz) <= std::numeric_limits<float>::min() ? 1.f : std::exp(-std::abs(dz / w.z));
    }
    static float hg(float cos_theta_, float g) {// HGPhaseFunction::HenyeyGreenstein, :21-24
        auto denom = 1.f + sqr(g) + 2.f * g * cos_theta_;
        return inv_pi / 4.0f * (1.f - sqr(g)) / (denom * std::sqrt(denom));
    }
    static float3 hg_sample(float3 wo, float g, float2 u, float &pdf) {// :25-38
        auto cos_t = std::abs(g) < 1e-3f ? 1.f - 2.f * u.x : -1.f / (2.f * g) * (1.f + sqr(g) - sqr((1.f - sqr(g)) / (1.f + g - 2.f * g * u.x)));
        auto sin_t = std::sqrt(1.f - sqr(cos_t));
        auto phi = 2.f * pi * u.y;
        auto frame = Frame::make(wo);
        auto wi = frame.local_to_world(f3(sin_t * std::cos(phi), sin_t * std::sin(phi), cos_t));
        pdf = hg(cos_t, g);
        return wi;
    }
    static float power_heuristic(float f, float g) {// sampling.cpp:142-159
        auto ff = f * f, gg = g * g;
        auto sum = ff + gg;
        return std::isinf(ff) ? 1.f : (sum == 0.f ? 0.f : ff / sum);
    }
    static bool is_zero(Spectrum3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }
    static uint32_t bits(float x) { return float_bits(x); }

    SurfaceEval layered_evaluate(float3 wo, float3 wi) const {// :256-398
        auto L = layers();
        auto mode = importance, reverse_mode = !importance;
        auto thickness = s0, g = s1;
        auto albedo = c0;
        auto samples = static_cast<float>(layer_samples);
        auto wi_local = shading.world_to_local(wi), wo_local = shading.world_to_local(wo);
        auto entered_top = wo_local.z > 0.f;
        auto sh = same_hemisphere(wo_local, wi_local);
        auto &enter = entered_top ? L[0] : L[1];
        auto &exit = (sh != entered_top) ? L[1] : L[0];
        auto &nonexit = (sh != entered_top) ? L[0] : L[1];
        auto exit_z = (sh != entered_top) ? 0.f : thickness;
        auto f = sh ? samples * enter.evaluate(wo, wi, mode).f : f3(0.f);
        auto seed = xxhash32(bits(mix_it.pg.x), bits(mix_it.pg.y), bits(mix_it.pg.z), xxhash32(bits(wi.x), bits(wi.y), bits(wi.z)));
        auto pdf_sum = sh ? samples * (entered_top ? L[0] : L[1]).evaluate(wo, wi, mode).pdf : 0.f;
        auto draw3 = [&](float &uc, float2 &u) { uc = lcg(seed), u.x = lcg(seed), u.y = lcg(seed); };
        for (auto i = 0u; i < layer_samples; i++) {
            float uc;
            float2 u;
            draw3(uc, u);
            auto wos = enter.sample(wo, uc, u, mode);
            if (is_zero(wos.eval.f) || wos.eval.pdf <= 0.f) { continue; }
            draw3(uc, u);
            auto wis = exit.sample(wi, uc, u, reverse_mode);
            auto wis_wi_local = exit.shading.world_to_local(wis.wi);
            if (is_zero(wis.eval.f) || wis.eval.pdf <= 0.f) { continue; }
            auto beta = wos.eval.f * (1.f / wos.eval.pdf);
            auto z = entered_top ? thickness : 0.f;
            auto w = wos.wi;
            auto w_local = enter.shading.world_to_local(w);
            for (auto depth = 0u; depth < layer_max_depth; depth++) {
                if (depth > 3u && max_component(beta) < 0.25f) {
                    auto q = std::max(0.f, 1.f - max_component(beta));
                    if (lcg(seed) < q) { break; }
                    beta = beta * (1.f / (1.f - q));
                }
                if (is_zero(albedo)) {
                    z = z == thickness ? 0.f : thickness;
                    beta = beta * layer_tr(thickness, w_local);
                } else {
                    auto sigma_t = 1.f;
                    auto dz = -std::log(1.f - lcg(seed)) / (sigma_t / std::abs(w_local.z));
                    auto zp = w_local.z > 0.f ? z + dz : z - dz;
                    if (z == zp) { continue; }
                    if (zp > 0.f && zp < thickness) {
                        auto wt = power_heuristic(wis.eval.pdf, nonexit.evaluate(-w, -wis.wi, mode).pdf);
                        f += beta * albedo * hg(dot(-w_local, -wis_wi_local), g) * wt * layer_tr(zp - exit_z, wis_wi_local) * wis.eval.f *
                             (1.f / wis.eval.pdf);
                        float2 up{lcg(seed), 0.f};
                        up.y = lcg(seed);
                        float ps_pdf;
                        auto ps_wi = hg_sample(-w_local, g, up, ps_pdf);
                        if (ps_pdf <= 0.f || ps_wi.z == 0.f) { continue; }
                        beta = beta * albedo * (ps_pdf / ps_pdf);
                        w_local = ps_wi;
                        w = exit.shading.local_to_world(w_local);
                        z = zp;
                        if ((z < exit_z && w_local.z > 0.f) || (z > exit_z && w_local.z < 0.f)) {
                            auto e = exit.evaluate(-w, wi, mode);
                            if (!is_zero(e.f)) {
                                auto wte = power_heuristic(ps_pdf, e.pdf);
                                f += beta * layer_tr(zp - exit_z, w_local) * e.f * wte;
                            }
                        }
                        continue;
                    }
                    z = clampf(zp, 0.f, thickness);
                }
                if (z == exit_z) {
                    draw3(uc, u);
                    auto bs = exit.sample(-w, uc, u, mode);
                    if (is_zero(bs.eval.f) || bs.eval.pdf <= 0.f) { break; }
                    beta = beta * bs.eval.f * (1.f / bs.eval.pdf);
                    w = bs.wi;
                    w_local = exit.shading.world_to_local(w);
                } else {
                    auto wns = nonexit.evaluate(-w, -wis.wi, mode);
                    auto wt = power_heuristic(wis.eval.pdf, wns.pdf);
                    f += beta * wns.f * wt * layer_tr(thickness, wis_wi_local) * wis.eval.f * (1.f / wis.eval.pdf);
                    draw3(uc, u);
                    auto bs = nonexit.sample(-w, uc, u, mode);
                    if (is_zero(bs.eval.f) || bs.eval.pdf <= 0.f) { break; }
                    beta = beta * bs.eval.f * (1.f / bs.eval.pdf);
                    w = bs.wi;
                    w_local = nonexit.shading.world_to_local(w);
                    auto wes = exit.evaluate(-w, wi, mode);
                    if (!is_zero(wes.f)) {
                        auto wte = power_heuristic(bs.eval.pdf, wes.pdf);
                        f += beta * layer_tr(thickness, nonexit.shading.world_to_local(bs.wi)) * wes.f * wte;
                    }
                }
            }
        }
        for (auto i = 0u; i < layer_samples; i++) {// pdf estimate, :360-395
            float uc;
            float2 u;
            if (sh) {
                auto &r = entered_top ? L[1] : L[0];
                auto &t = entered_top ? L[0] : L[1];
                draw3(uc, u);
                auto wos = t.sample(wo, uc, u, mode);
                draw3(uc, u);
                auto wis = t.sample(wi, uc, u, reverse_mode);
                if (!is_zero(wos.eval.f) && wos.eval.pdf > 0.f && !is_zero(wis.eval.f) && wis.eval.pdf > 0.f) {
                    draw3(uc, u);
                    auto rs = r.sample(-wos.wi, uc, u, mode);
                    if (!is_zero(rs.eval.f) && rs.eval.pdf > 0.f) {
                        auto r_pdf = r.evaluate(-wos.wi, -wis.wi, mode).pdf;
                        pdf_sum += power_heuristic(wis.eval.pdf, r_pdf) * r_pdf;
                        auto t_pdf = t.evaluate(-rs.wi, wi, mode).pdf;
                        pdf_sum += power_heuristic(rs.eval.pdf, t_pdf) * t_pdf;
                    }
                }
            } else {
                auto &ti = entered_top ? L[1] : L[0];
                auto &to = entered_top ? L[0] : L[1];
                draw3(uc, u);
                auto wos = to.sample(wo, uc, u, mode);
                draw3(uc, u);
                auto wis = ti.sample(wi, uc, u, reverse_mode);
                if (is_zero(wos.eval.f) || wos.eval.pdf <= 0.f || is_zero(wis.eval.f) || wis.eval.pdf <= 0.f) { continue; }
                pdf_sum += .5f * (to.evaluate(wo, -wis.wi, mode).pdf + ti.evaluate(-wos.wi, wi, mode).pdf);
            }
        }
        return {f * (1.f / samples), lerp(1.f / (4.f * pi), pdf_sum / samples, 0.9f)};
    }

    SurfaceSample layered_sample(float3 wo, float u_lobe, float2 u) const {// :399-470
        auto L = layers();
        auto mode = importance;
        auto thickness = s0, g = s1;
        auto albedo = c0;
        auto wo_local = shading.world_to_local(wo);
        auto entered_top = wo_local.z > 0.f;
        auto bs = (entered_top ? L[0] : L[1]).sample(wo, u_lobe, u, mode);
        SurfaceSample s;
        s.eval = {f3(0.f), 0.f};
        if (is_zero(bs.eval.f) || bs.eval.pdf == 0.f) { return s; }
        auto wi_local = shading.world_to_local(bs.wi);
        if (same_hemisphere(wi_local, wo_local)) { return bs; }
        auto w = bs.wi;
        auto w_local = wi_local;
        auto seed = xxhash32(bits(u.x), bits(u.y), bits(u_lobe), xxhash32(bits(wo.x), bits(wo.y), bits(wo.z)));
        auto f = bs.eval.f;
        auto pdf = bs.eval.pdf;
        auto z = entered_top ? thickness : 0.f;
        for (auto depth = 0u; depth < layer_max_depth; depth++) {
            auto rr_beta = max_component(f) / pdf;
            if (depth > 3u && rr_beta < 0.25f) {
                auto q = std::max(0.f, 1.f - rr_beta);
                if (lcg(seed) < q) { break; }
                pdf *= 1.f - q;
            }
            if (w_local.z == 0.f) { break; }
            if (!is_zero(albedo)) {
                auto sigma_t = 1.f;
                auto dz = -std::log(1.f - lcg(seed)) / (sigma_t / std::abs(w_local.z));
                auto zp = w_local.z > 0.f ? z + dz : z - dz;
                if (z == zp) { break; }
                if (0.f < zp && zp < thickness) {
                    float2 up{lcg(seed), 0.f};
                    up.y = lcg(seed);
                    float ps_pdf;
                    auto ps_wi = hg_sample(-w_local, g, up, ps_pdf);
                    if (ps_pdf <= 0.f) { break; }
                    f = f * albedo * ps_pdf;
                    pdf *= ps_pdf;
                    w = ps_wi;// (the reference assigns the phase sample, a LOCAL direction, to the world-space w: kept)
                    w_local = shading.world_to_local(w);
                    z = zp;
                    continue;
                }
                z = clampf(zp, 0.f, thickness);
            } else {
                z = z == thickness ? 0.f : thickness;
                f = f * layer_tr(thickness, w_local);
            }
            auto &interface = z == 0.f ? L[1] : L[0];
            auto uc = lcg(seed);
            float2 ub{lcg(seed), 0.f};
            ub.y = lcg(seed);
            auto is = interface.sample(-w, uc, ub, mode);
            if (is_zero(is.eval.f) || is.eval.pdf <= 0.f) { break; }
            f = f * is.eval.f;
            pdf *= is.eval.pdf;
            w = is.wi;
            w_local = shading.world_to_local(w);
            if (is.event == EVENT_ENTER || is.event == EVENT_EXIT) {// (bs.event & Surface::event_transmit) != 0
                s.eval = {f, pdf};
                s.wi = w;
                s.event = same_hemisphere(w_local, wo_local) ? EVENT_REFLECT : (w_local.z > 0.f ? EVENT_EXIT : EVENT_ENTER);
                break;
            }
        }
        return s;
    }

    // ---- per-kind _evaluate / _sample in world space (f already includes |cos theta_i|)
    SurfaceEval evaluate_impl(float3 wo, float3 wi) const {
        auto wo_local = shading.world_to_local(wo);
        auto wi_local = shading.world_to_local(wi);
        switch (kind) {
            case LR_SURFACE_MATTE: {// matte.cpp:86-96
                OrenNayar refl{c0, s0};
                return {refl.evaluate(wo_local, wi_local) * abs_cos_theta(wi_local), cosine_pdf(wo_local, wi_local)};
            }
            case LR_SURFACE_MIRROR: {// mirror.cpp:101-115
                TrowbridgeReitz dist{alpha};
                auto fresnel = [&](float cosI) { return schlick(c0, cosI); };
                auto f = microfacet_reflection_eval(c0, dist, fresnel, wo_local, wi_local);
                return {f * abs_cos_theta(wi_local), microfacet_reflection_pdf(dist, wo_local, wi_local)};
            }
            case LR_SURFACE_GLASS: {// glass.cpp:169-193
                TrowbridgeReitz dist{alpha};
                auto eta_i = s0, eta_t = s1;
                auto ratio = glass_refl_prob(eta_i, eta_t, s2, wo_local);
                Spectrum3 f;
                float pdf;
                if (same_hemisphere(wo_local, wi_local)) {
                    auto fresnel = [&](float cosI) { return f3(fresnel_dielectric(cosI, eta_i, eta_t)); };
                    f = microfacet_reflection_eval(c0, dist, fresnel, wo_local, wi_local);
                    pdf = microfacet_reflection_pdf(dist, wo_local, wi_local) * ratio;
                } else {
                    f = microfacet_transmission_eval(c1, dist, eta_i, eta_t, wo_local, wi_local, importance);
                    pdf = microfacet_transmission_pdf(dist, eta_i, eta_t, wo_local, wi_local) * (1.f - ratio);
                }
                return {f * abs_cos_theta(wi_local), pdf};
            }
            case LR_SURFACE_PLASTIC: {// plastic.cpp:139-166
                auto sgn = cos_theta(wo_local) < 0.f ? f3(1.f, 1.f, -1.f) : f3(1.f, 1.f, 1.f);
                wo_local = wo_local * sgn;
                wi_local = sgn * wi_local;
                return plastic_eval_local(wo_local, wi_local);
            }
            case LR_SURFACE_METAL: {// metal.cpp:228-241
                TrowbridgeReitz dist{alpha};
                auto fresnel = [&](float cosI) { return fresnel_conductor(std::abs(cosI), s0, c0, c1); };
                auto f = microfacet_reflection_eval(f3(1.f), dist, fresnel, wo_local, wi_local) * c2;
                return {f * abs_cos_theta(wi_local), microfacet_reflection_pdf(dist, wo_local, wi_local)};
            }
            case LR_SURFACE_DISNEY: {
                DisneyClosure dc{disney};
                dc.importance = importance;
                return dc.evaluate_local(wo_local, wi_local);
            }
            case LR_SURFACE_LAYERED: return layered_evaluate(wo, wi);
            case LR_SURFACE_MIX: {// mix.cpp:169-177: children through their public evaluate (side validation included)
                // (the transport mode goes down to the children: a Mix can be an interface of a Layered surface)
                auto ea = populate_tag(*mix_scene, mix_a, mix_it, mix_wo, mix_eta_i).evaluate(wo, wi, importance);
                auto eb = populate_tag(*mix_scene, mix_b, mix_it, mix_wo, mix_eta_i).evaluate(wo, wi, importance);
                return mix(ea, eb, s0);
            }
            default: return {};
        }
    }

    static SurfaceEval mix(const SurfaceEval &a, const SurfaceEval &b, float ratio) {// mix.cpp:97-104
        auto t = 1.f - ratio;
        return {lerp(a.f, b.f, t), lerp(a.pdf, b.pdf, t)};
    }

    SurfaceSample sample_impl(float3 wo, float u_lobe, float2 u) const {
        auto wo_local = shading.world_to_local(wo);
        SurfaceSample out;
        switch (kind) {
            case LR_SURFACE_MATTE: {// matte.cpp:98-112
                OrenNayar refl{c0, s0};
                auto wi_local = cosine_sample_wi(wo_local, u);
                auto pdf = cosine_pdf(wo_local, wi_local);
                auto f = refl.evaluate(wo_local, wi_local);
                out.wi = shading.local_to_world(wi_local);
                out.eval = {f * abs_cos_theta(wi_local), pdf};
                return out;
            }
            case LR_SURFACE_MIRROR: {// mirror.cpp:116-135
                TrowbridgeReitz dist{alpha};
                auto fresnel = [&](float cosI) { return schlick(c0, cosI); };
                auto s = microfacet_reflection_sample(c0, dist, fresnel, wo_local, u);
                out.wi = shading.local_to_world(s.wi);
                out.eval = {s.f * abs_cos_theta(s.wi), s.pdf};
                return out;
            }
            case LR_SURFACE_GLASS: {// glass.cpp:195-228
                TrowbridgeReitz dist{alpha};
                auto eta_i = s0, eta_t = s1;
                auto ratio = glass_refl_prob(eta_i, eta_t, s2, wo_local);
                BxdfSample s;
                if (u_lobe < ratio) {
                    auto fresnel = [&](float cosI) { return f3(fresnel_dielectric(cosI, eta_i, eta_t)); };
                    s = microfacet_reflection_sample(c0, dist, fresnel, wo_local, u);
                    s.pdf *= ratio;
                } else {
                    s = microfacet_transmission_sample(c1, dist, eta_i, eta_t, wo_local, u, importance);
                    s.pdf *= (1.f - ratio);
                    out.event = cos_theta(wo_local) > 0.f ? EVENT_ENTER : EVENT_EXIT;
                }
                out.wi = shading.local_to_world(s.wi);
                out.eval = {s.f * abs_cos_theta(s.wi), s.pdf};
                return out;
            }
            case LR_SURFACE_PLASTIC: {// plastic.cpp:168-213
                auto sgn = cos_theta(wo_local) < 0.f ? f3(1.f, 1.f, -1.f) : f3(1.f, 1.f, 1.f);
                wo_local = wo_local * sgn;
                auto eta = s1;
                auto Fo = fresnel_dielectric(abs_cos_theta(wo_local), 1.f, eta);
                auto substrate_weight = plastic_substrate_weight(Fo, s0);
                float3 wi_local;
                bool valid;
                if (u_lobe < substrate_weight) {
                    wi_local = cosine_sample_wi(wo_local, u);
                    valid = true;
                } else {
                    TrowbridgeReitz dist{alpha};
                    auto wh = dist.sample_wh(wo_local, u);
                    wi_local = reflect(-wo_local, wh);
                    valid = same_hemisphere(wo_local, wi_local);
                }
                if (valid) {
                    out.wi = shading.local_to_world(wi_local * sgn);
                    out.eval = plastic_eval_local(wo_local, wi_local);
                }
                return out;
            }
            case LR_SURFACE_METAL: {// metal.cpp:242-260
                TrowbridgeReitz dist{alpha};
                auto fresnel = [&](float cosI) { return fresnel_conductor(std::abs(cosI), s0, c0, c1); };
                auto s = microfacet_reflection_sample(f3(1.f), dist, fresnel, wo_local, u);
                s.f = s.f * c2;
                out.wi = shading.local_to_world(s.wi);
                out.eval = {s.f * abs_cos_theta(s.wi), s.pdf};
                return out;
            }
            case LR_SURFACE_DISNEY: {
                DisneyClosure dc{disney};
                dc.importance = importance;
                float3 wi_local;
                bool valid;
                dc.sample_local(wo_local, u_lobe, u, wi_local, valid, out.event);
                out.wi = shading.local_to_world(wi_local);
                if (valid) { out.eval = dc.evaluate_local(wo_local, wi_local); }
                return out;
            }
            case LR_SURFACE_LAYERED: return layered_sample(wo, u_lobe, u);
            case LR_SURFACE_MIX: {// mix.cpp:178-196 — the "sample b" branch samples A and evaluates B (reference quirk, kept)
                auto a = populate_tag(*mix_scene, mix_a, mix_it, mix_wo, mix_eta_i);
                auto b = populate_tag(*mix_scene, mix_b, mix_it, mix_wo, mix_eta_i);
                auto ratio = s0;
                if (u_lobe < ratio) {
                    auto sa = a.sample(wo, u_lobe / ratio, u, importance);
                    auto eb = b.evaluate(wo, sa.wi, importance);
                    out.eval = mix(sa.eval, eb, ratio);
                    out.wi = sa.wi, out.event = sa.event;
                } else {
                    auto sb = a.sample(wo, (u_lobe - ratio) / (1.f - ratio), u, importance);
                    auto ea = b.evaluate(wo, sb.wi, importance);
                    out.eval = mix(ea, sb.eval, ratio);
                    out.wi = sb.wi, out.event = sb.event;
                }
                return out;
            }
            default: return out;
        }
    }

    // Surface::Closure::evaluate / sample, surface.cpp:45-68
    SurfaceEval evaluate(float3 wo, float3 wi, bool importance_mode) {
        importance = importance_mode;
        return evaluate(wo, wi);
    }
    SurfaceSample sample(float3 wo, float u_lobe, float2 u, bool importance_mode) {
        importance = importance_mode;
        return sample(wo, u_lobe, u);
    }
    SurfaceEval evaluate(float3 wo, float3 wi) const {
        auto e = evaluate_impl(wo, wi);
        if (!validate_surface_sides(ng, shading.n, wo, wi)) { e.f = f3(0.f), e.pdf = 0.f; }
        return e;
    }
    SurfaceSample sample(float3 wo, float u_lobe, float2 u) const {
        auto s = sample_impl(wo, u_lobe, u);
        if (!validate_surface_sides(ng, shading.n, wo, s.wi)) { s.eval.f = f3(0.f), s.eval.pdf = 0.f; }
        return s;
    }

private:
    static Spectrum3 schlick(Spectrum3 R0, float cosI) {// mirror.cpp:67-79
        auto m = saturate(1.f - cosI);
        auto weight = sqr(sqr(m)) * m;
        return (1.f - weight) * R0 + f3(weight);
    }
    static float glass_refl_prob(float eta_i, float eta_t, float kr_ratio, float3 wo_local) {// glass.cpp:160-166
        auto F = fresnel_dielectric(cos_theta(wo_local), eta_i, eta_t);
        auto r = kr_ratio * F;
        auto t = (1.f - kr_ratio) * (1.f - F);
        return r == 0.f ? 0.f : r / (r + t);
    }
    static float plastic_substrate_weight(float Fo, float kd_weight) {// plastic.cpp:126-129
        auto w = kd_weight * (1.0f - Fo);
        return w == 0.f ? 0.f : w / (w + Fo);
    }
    SurfaceEval plastic_eval_local(float3 wo_local, float3 wi_local) const {// plastic.cpp:147-163,191-206
        TrowbridgeReitz dist{alpha};
        auto eta = s1;
        auto fresnel = [&](float cosI) { return f3(fresnel_dielectric(cosI, 1.f, eta)); };
        auto f_coat = microfacet_reflection_eval(f3(1.f), dist, fresnel, wo_local, wi_local);
        auto pdf_coat = microfacet_reflection_pdf(dist, wo_local, wi_local);
        auto Fi = fresnel_dielectric(abs_cos_theta(wi_local), 1.f, eta);
        auto Fo = fresnel_dielectric(abs_cos_theta(wo_local), 1.f, eta);
        auto a = exp3(-(1.f / abs_cos_theta(wi_local) + 1.f / abs_cos_theta(wo_local)) * c1);
        auto f_diffuse = (1.f - Fi) * (1.f - Fo) * sqr(1.f / eta) * a * lambert_eval(c0, wo_local, wi_local);
        auto pdf_diffuse = cosine_pdf(wo_local, wi_local);
        auto substrate_weight = plastic_substrate_weight(Fo, s0);
        return {(f_coat + f_diffuse) * abs_cos_theta(wi_local), lerp(pdf_coat, pdf_diffuse, substrate_weight)};
    }
};

}// namespace oracle
