// oracle_math.h — scalar fp32 vector helpers for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY: everything under oracle/ is a CPU restatement of the reference
// algorithm used as the parity checker (tests/, __graft_entry__.smoke(), bench.py's
// cpu_baseline leg).  The product path (luisarender_amd/csrc/hip) never includes, links or
// calls it.  Pinned to the reference's own code compiled in place (oracle/_ref, tests/test_oracle_vs_ref.py);
// the builtins below are the ones oracle/ref_shim/lc_types.h gives that code, operation for operation.
//
// Semantics follow the LuisaCompute DSL builtins the reference is written in (absent
// submodule; restated): sign(x) = copysign(1, x) (the published branch-free ONB of
// src/util/frame.cpp:21-28 needs sign(0) = 1), fract(x) = x - floor(x),
// lerp(a, b, t) = a + t * (b - a), saturate = clamp(x, 0, 1).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

namespace oracle {

constexpr float pi = 3.14159265358979323846f;
constexpr float inv_pi = 0.318309886183790671537767526745028724f;
constexpr float pi_over_two = 1.57079632679489661923132169163975144f;
constexpr float pi_over_four = 0.785398163397448309615660845819875721f;
constexpr float one_minus_epsilon = 0x1.fffffep-1f;

struct float2 {
    float x{}, y{};
};
struct float3 {
    float x{}, y{}, z{};
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
};
struct float4 {
    float x{}, y{}, z{}, w{};
};

inline float3 f3(float x, float y, float z) { return {x, y, z}; }
inline float3 f3(float s) { return {s, s, s}; }
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float3 &operator+=(float3 &a, float3 b) { return a = a + b; }
inline float3 &operator*=(float3 &a, float3 b) { return a = a * b; }
inline float3 &operator*=(float3 &a, float s) { return a = a * s; }
inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator-(float2 a, float2 b) { return {a.x - b.x, a.y - b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
inline float2 operator*(float2 a, float2 b) { return {a.x * b.x, a.y * b.y}; }

inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(float3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float sqr(float x) { return x * x; }
inline float sign(float x) { return std::copysign(1.0f, x); }
inline float fract(float x) { return x - std::floor(x); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float3 lerp(float3 a, float3 b, float t) { return a + t * (b - a); }
inline float saturate(float x) { return std::min(std::max(x, 0.0f), 1.0f); }
inline float clampf(float x, float lo, float hi) { return std::min(std::max(x, lo), hi); }
inline float3 saturate(float3 v) { return {saturate(v.x), saturate(v.y), saturate(v.z)}; }
inline float3 max0(float3 v) { return {std::max(v.x, 0.f), std::max(v.y, 0.f), std::max(v.z, 0.f)}; }
inline float max_component(float3 v) { return std::max(v.x, std::max(v.y, v.z)); }
inline float3 abs3(float3 v) { return {std::abs(v.x), std::abs(v.y), std::abs(v.z)}; }
inline float3 exp3(float3 v) { return {std::exp(v.x), std::exp(v.y), std::exp(v.z)}; }
inline float3 sqrt3(float3 v) { return {std::sqrt(v.x), std::sqrt(v.y), std::sqrt(v.z)}; }
inline bool any_nan(float3 v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }
inline bool any_inf(float3 v) { return std::isinf(v.x) || std::isinf(v.y) || std::isinf(v.z); }
inline float3 reflect(float3 i, float3 n) { return i - 2.0f * dot(n, i) * n; }
inline float3 face_forward(float3 v, float3 n) { return dot(v, n) < 0.f ? -v : v; }// scattering.cpp:76-78
inline float srgb_to_cie_y(float3 rgb) { return dot(f3(0.212671f, 0.715160f, 0.072169f), rgb); }// colorspace.h:21-25

inline uint32_t float_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float bits_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// column-major 3x3 / 4x4 (luisa::float3x3 / float4x4)
struct mat3 {
    float3 c[3];
};
inline float3 operator*(const mat3 &m, float3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
inline mat3 transpose(const mat3 &m) {
    return {{{m.c[0].x, m.c[1].x, m.c[2].x}, {m.c[0].y, m.c[1].y, m.c[2].y}, {m.c[0].z, m.c[1].z, m.c[2].z}}};
}
// luisa::inverse(float3x3) (core/mathematics.h): adjugate / determinant, fp32
inline mat3 inverse(const mat3 &m) {
    auto one_over_det = 1.0f / (m.c[0].x * (m.c[1].y * m.c[2].z - m.c[2].y * m.c[1].z) -
                                m.c[1].x * (m.c[0].y * m.c[2].z - m.c[2].y * m.c[0].z) +
                                m.c[2].x * (m.c[0].y * m.c[1].z - m.c[1].y * m.c[0].z));
    mat3 r;
    r.c[0] = f3((m.c[1].y * m.c[2].z - m.c[2].y * m.c[1].z) * one_over_det,
                (m.c[2].y * m.c[0].z - m.c[0].y * m.c[2].z) * one_over_det,
                (m.c[0].y * m.c[1].z - m.c[1].y * m.c[0].z) * one_over_det);
    r.c[1] = f3((m.c[2].x * m.c[1].z - m.c[1].x * m.c[2].z) * one_over_det,
                (m.c[0].x * m.c[2].z - m.c[2].x * m.c[0].z) * one_over_det,
                (m.c[1].x * m.c[0].z - m.c[0].x * m.c[1].z) * one_over_det);
    r.c[2] = f3((m.c[1].x * m.c[2].y - m.c[2].x * m.c[1].y) * one_over_det,
                (m.c[2].x * m.c[0].y - m.c[0].x * m.c[2].y) * one_over_det,
                (m.c[0].x * m.c[1].y - m.c[1].x * m.c[0].y) * one_over_det);
    return r;
}

// ---- src/util/frame.{h,cpp}
struct Frame {
    float3 s{1.f, 0.f, 0.f}, t{0.f, 1.f, 0.f}, n{0.f, 0.f, 1.f};
    static Frame make(float3 n) {// frame.cpp:21-28
        auto sgn = sign(n.z);
        auto a = -1.f / (sgn + n.z);
        auto b = n.x * n.y * a;
        auto s = f3(1.f + sgn * sqr(n.x) * a, sgn * b, -sgn * n.x);
        auto t = f3(b, sgn + sqr(n.y) * a, -n.y);
        return {normalize(s), normalize(t), n};
    }
    static Frame make(float3 n, float3 s) {// frame.cpp:30-34
        auto ss = normalize(s - n * dot(n, s));
        auto tt = normalize(cross(n, ss));
        return {ss, tt, n};
    }
    float3 local_to_world(float3 d) const { return normalize(d.x * s + d.y * t + d.z * n); }          // :36-38
    float3 world_to_local(float3 d) const { return normalize(f3(dot(d, s), dot(d, t), dot(d, n))); }// :40-42
};

inline float3 clamp_shading_normal(float3 ns, float3 ng, float3 w) {// frame.cpp:49-54
    auto w_refl = reflect(-w, ns);
    auto w_refl_clip = dot(w_refl, ng) * dot(w, ng) > 0.f ? w_refl : normalize(w_refl - ng * dot(w_refl, ng));
    return normalize(w_refl_clip + w);
}

// local shading-space trigonometry, frame.h:48-71
inline float cos_theta(float3 w) { return w.z; }
inline float cos2_theta(float3 w) { return sqr(w.z); }
inline float abs_cos_theta(float3 w) { return std::abs(w.z); }
inline float sin2_theta(float3 w) { return saturate(1.0f - cos2_theta(w)); }
inline float sin_theta(float3 w) { return std::sqrt(sin2_theta(w)); }
inline float tan_theta(float3 w) { return sin_theta(w) / cos_theta(w); }
inline float tan2_theta(float3 w) { return sin2_theta(w) / cos2_theta(w); }
inline float cos_phi(float3 w) {
    auto s = sin_theta(w);
    return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
inline float sin_phi(float3 w) {
    auto s = sin_theta(w);
    return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
inline float cos2_phi(float3 w) { return sqr(cos_phi(w)); }
inline float sin2_phi(float3 w) { return sqr(sin_phi(w)); }
inline bool same_hemisphere(float3 w, float3 wp) { return w.z * wp.z > 0.0f; }
inline float abs_dot(float3 u, float3 v) { return std::abs(dot(u, v)); }

}// namespace oracle
