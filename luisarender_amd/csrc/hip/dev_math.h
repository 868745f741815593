// dev_math.h — fp32 vector helpers for the gfx950 megakernel (device side) and for the host
// code that pre-resolves constant closures at upload time (same arithmetic, LR_HD functions).
//
// Builtin semantics follow the LuisaCompute DSL the reference's device code is written in
// (sign = copysign(1, x), fract = x - floor(x), lerp = a + t (b - a)); see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define LR_HD __host__ __device__ __forceinline__
#define LR_D __device__ __forceinline__
// Wave votes on a BOOL (round 5).  HIP's __any / __ballot take an int: the compiler materialises the predicate in a VGPR and compares
// it with zero again (v_cndmask_b32_e64 v, 0, 1, s[..] + v_cmp_ne_u32 s[..], 0, v -- two half-rate VALU instructions per vote, five votes
// per iteration of the traversal loop: 42 of its ~960 issue cycles).  The ballot builtin takes the lane mask the compare already
// produced (LR_VOTE_BUILTIN=0 restores the library forms for A/B).
#ifndef LR_VOTE_BUILTIN
#define LR_VOTE_BUILTIN 1
#endif
#if LR_VOTE_BUILTIN
__device__ __forceinline__ unsigned long long lr_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lr_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
#else
__device__ __forceinline__ unsigned long long lr_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ bool lr_any(bool p) { return __any(p) != 0; }
#endif
// Out-of-line device functions.  LR_HEAVY: the Disney / Mix / Layered path of the variants that hold Mix or Layered
// (dev_heavy.h) is always a real call.  LR_CALL: the texture lookup and the environment evaluate / sample are real
// calls only in those same variants (one copy instead of one per use: -60 % code, fewer spills in the main loop);
// in the lean ones the call overhead costs more than it saves (measured inline vs call: C2 +2 %, C3 +1.4 %, C4 +7 %).
// Every variant is its own translation unit (megapath_variant.hip defines LR_VARIANT), so this is a preprocessor choice.
// LR_TEX_LAMBDA: the texture lookup behind the callback resolve_closure gets from load_lobe (dev_heavy.h; dev_shade.h: texture_eval_slot -- a
// capturing lambda until round 6), its ~20 uses in ONE place.  In the lean
// variants it is the one real call they make -- only hits on textured closures take it: <0> 73 -> 58 KB, <20> 220 -> 153 KB, and C2,
// which never runs it, +1.5 % from what the rest of the kernel gets out of the smaller function (C3 / C4 unchanged; inlined at every
// use: C4 -1.2 %; EVERY texture lookup through one out-of-line function instead: C2 the same, C3 -0.5 %, C4 -1 %;
// profiles/archive/r03ae_texture_lambda_ab.txt).  Rounds 1-2 had the same call by accident: the inliner left the lambda out
// of line while the texture code still held powf.
#ifndef LR_CALL
#if defined(LR_VARIANT) && ((LR_VARIANT) & (96 | 256)) && !defined(LR_CALL_INLINE)
#define LR_CALL __device__ __noinline__
#define LR_TEX_LAMBDA __device__ __forceinline__
#else
#define LR_CALL __device__ __forceinline__
#define LR_TEX_LAMBDA __device__ __noinline__
#endif
#endif
#ifndef LR_TEX_LAMBDA
#define LR_TEX_LAMBDA __device__ __forceinline__// (LR_CALL given by the translation unit, heavy_variant.hip: texture_eval_tables is a real call itself)
#endif
#ifndef LR_TEX_BY_VALUE
#define LR_TEX_BY_VALUE 0
#endif
#if defined(LR_VARIANT) && !((LR_VARIANT) & (96 | 256)) && !defined(LR_CALL_INLINE)
#define LR_TEX_LAMBDA_ATTR __attribute__((noinline))
#else
#define LR_TEX_LAMBDA_ATTR
#endif
#ifndef LR_HEAVY
#define LR_HEAVY __device__ __noinline__
#endif

// hit reconstruction gathers from the baked per-triangle records (dev_scene.h: DShadeTri): +2.6 % on C2;
// -DLR_BAKED_SHADING=0 restores the instance -> triangle -> vertices chain of the reference for A/B
#ifndef LR_BAKED_SHADING
#define LR_BAKED_SHADING 1
#endif

namespace lrd {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.318309886183790671537767526745028724f;
constexpr float kPiOverTwo = 1.57079632679489661923132169163975144f;
constexpr float kPiOverFour = 0.785398163397448309615660845819875721f;
constexpr float kOneMinusEpsilon = 0x1.fffffep-1f;
constexpr float kFloatMax = 3.402823466e+38f;

struct f2 {
    float x, y;
};
struct f3 {
    float x, y, z;
};

LR_HD f3 mk3(float x, float y, float z) { return {x, y, z}; }
LR_HD f3 mk3(float s) { return {s, s, s}; }
LR_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
LR_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
LR_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
LR_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
LR_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
LR_HD f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
LR_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
LR_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
LR_HD f3 &operator+=(f3 &a, f3 b) { a = a + b; return a; }
LR_HD f3 &operator*=(f3 &a, f3 b) { a = a * b; return a; }
LR_HD f3 &operator*=(f3 &a, float s) { a = a * s; return a; }

LR_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LR_HD f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
LR_HD float length(f3 a) { return sqrtf(dot(a, a)); }
LR_HD f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
LR_HD float sqr(float x) { return x * x; }
LR_HD float sign(float x) { return copysignf(1.0f, x); }
LR_HD float fract(float x) { return x - floorf(x); }
LR_HD float lerp(float a, float b, float t) { return a + t * (b - a); }
LR_HD float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
LR_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
LR_HD f3 saturate(f3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
LR_HD f3 max0(f3 v) { return {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f)}; }
LR_HD float max_component(f3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
// ---- trigonometry without the libm range reduction (round 3).  With the project's flags cosf(x) + sinf(x) is 232 VALU instructions on
// gfx950 (the argument reduction for arbitrary x), powf(x, y) 158; the shading code only ever needs angles of the forms below.
// sin and cos of x, |x| <= pi / 4: Taylor polynomials to x^9 / x^10 in Horner form, max abs error 6.7e-8 (libm: 3.8e-8)
LR_HD void sincos_small(float x, float &s, float &c) {
    const auto x2 = x * x;
    s = x * (1.f + x2 * (-1.66666667e-1f + x2 * (8.33333333e-3f + x2 * (-1.98412698e-4f + x2 * 2.75573192e-6f))));
    c = 1.f + x2 * (-0.5f + x2 * (4.16666667e-2f + x2 * (-1.38888889e-3f + x2 * (2.48015873e-5f + x2 * -2.75573192e-7f))));
}
// sin and cos of 2 pi u, u in [0, 1] (any u: it is reduced exactly): quadrant q = round(4 u), the rest r = u - q / 4 in [-1/8, 1/8] is
// exact, the polynomials above on 2 pi r, a rotation by q quarter turns.  Max abs error 9.8e-8 -- more accurate than sinf(2 pi u)
// evaluated the obvious way, whose argument is rounded to fp32 first (4e-7) -- in ~30 instructions.
LR_HD void sincos_2pi(float u, float &s, float &c) {
    const auto q = floorf(4.f * u + 0.5f);
    float s0, c0;
    sincos_small((u - q * 0.25f) * 6.28318530717958647692f, s0, c0);
    const auto k = static_cast<int>(q) & 3;
    s = k == 0 ? s0 : (k == 1 ? c0 : (k == 2 ? -s0 : -c0));
    c = k == 0 ? c0 : (k == 1 ? -s0 : (k == 2 ? -c0 : s0));
}

LR_HD f3 exp3(f3 v) { return {expf(v.x), expf(v.y), expf(v.z)}; }
LR_HD f3 sqrt3(f3 v) { return {sqrtf(v.x), sqrtf(v.y), sqrtf(v.z)}; }
LR_HD bool any_nan(f3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }
LR_HD bool any_inf(f3 v) { return isinf(v.x) || isinf(v.y) || isinf(v.z); }
LR_HD f3 reflect(f3 i, f3 n) { return i - 2.0f * dot(n, i) * n; }
LR_HD f3 face_forward(f3 v, f3 n) { return dot(v, n) < 0.f ? -v : v; }
LR_HD float cie_y(f3 rgb) { return dot(mk3(0.212671f, 0.715160f, 0.072169f), rgb); }

// orthonormal shading frame (reference: src/util/frame.cpp:21-42)
struct Frame {
    f3 s, t, n;
};
LR_HD Frame frame_from_normal(f3 n) {
    auto sgn = sign(n.z);
    auto a = -1.f / (sgn + n.z);
    auto b = n.x * n.y * a;
    auto s = mk3(1.f + sgn * sqr(n.x) * a, sgn * b, -sgn * n.x);
    auto t = mk3(b, sgn + sqr(n.y) * a, -n.y);
    return {normalize(s), normalize(t), n};
}
LR_HD Frame frame_from_normal_tangent(f3 n, f3 s) {
    auto ss = normalize(s - n * dot(n, s));
    auto tt = normalize(cross(n, ss));
    return {ss, tt, n};
}
LR_HD f3 to_world(const Frame &f, f3 d) { return normalize(d.x * f.s + d.y * f.t + d.z * f.n); }
LR_HD f3 to_local(const Frame &f, f3 d) { return normalize(mk3(dot(d, f.s), dot(d, f.t), dot(d, f.n))); }
LR_HD f3 clamp_shading_normal(f3 ns, f3 ng, f3 w) {// frame.cpp:49-54
    auto w_refl = reflect(-w, ns);
    auto w_refl_clip = dot(w_refl, ng) * dot(w, ng) > 0.f ? w_refl : normalize(w_refl - ng * dot(w_refl, ng));
    return normalize(w_refl_clip + w);
}

// local-frame trigonometry (src/util/frame.h:48-71)
LR_HD float cos_theta(f3 w) { return w.z; }
LR_HD float cos2_theta(f3 w) { return w.z * w.z; }
LR_HD float abs_cos_theta(f3 w) { return fabsf(w.z); }
LR_HD float sin2_theta(f3 w) { return saturate(1.0f - cos2_theta(w)); }
LR_HD float sin_theta(f3 w) { return sqrtf(sin2_theta(w)); }
LR_HD float tan_theta(f3 w) { return sin_theta(w) / cos_theta(w); }
LR_HD float tan2_theta(f3 w) { return sin2_theta(w) / cos2_theta(w); }
LR_HD float cos_phi(f3 w) {
    auto s = sin_theta(w);
    return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
LR_HD float sin_phi(f3 w) {
    auto s = sin_theta(w);
    return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
LR_HD bool same_hemisphere(f3 w, f3 wp) { return w.z * wp.z > 0.0f; }
LR_HD float abs_dot(f3 u, f3 v) { return fabsf(dot(u, v)); }

}// namespace lrd
