// lr_math.h — minimal host-side vector/matrix types standing in for the (absent)
// LuisaCompute `luisa::float3 / float4x4` (core/basic_types.h).  Column-major 4x4.
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>

namespace lr {

struct float2 { float x{}, y{}; };
struct float3 {
    float x{}, y{}, z{};
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct float4 {
    float x{}, y{}, z{}, w{};
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float s) { return {s, s, s}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline float4 make_float4(float3 v, float w) { return {v.x, v.y, v.z, w}; }

inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline bool operator==(float3 a, float3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(float3 a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(float3 a) { return a * (1.0f / length(a)); }
inline float3 min3(float3 a, float3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline float3 max3(float3 a, float3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }

inline float4 operator+(float4 a, float4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline float4 operator*(float4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline bool operator==(float4 a, float4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

struct float4x4 {
    float4 c[4];// columns
    float4 &operator[](int i) { return c[i]; }
    const float4 &operator[](int i) const { return c[i]; }
    static float4x4 identity() {
        return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
    }
};

inline float4 operator*(const float4x4 &m, float4 v) {
    return m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w;
}
inline float4x4 operator*(const float4x4 &a, const float4x4 &b) {
    return {{a * b[0], a * b[1], a * b[2], a * b[3]}};
}
inline bool is_identity(const float4x4 &m) {
    return m[0] == float4{1, 0, 0, 0} && m[1] == float4{0, 1, 0, 0} &&
           m[2] == float4{0, 0, 1, 0} && m[3] == float4{0, 0, 0, 1};
}
inline float3 transform_point(const float4x4 &m, float3 p) {
    auto v = m * make_float4(p, 1.f);
    return {v.x, v.y, v.z};
}
inline float3 transform_vector(const float4x4 &m, float3 d) {
    return {m[0].x * d.x + m[1].x * d.y + m[2].x * d.z,
            m[0].y * d.x + m[1].y * d.y + m[2].y * d.z,
            m[0].z * d.x + m[1].z * d.y + m[2].z * d.z};
}

inline float4x4 translation(float3 t) {
    auto m = float4x4::identity();
    m[3] = {t.x, t.y, t.z, 1.f};
    return m;
}
inline float4x4 scaling(float3 s) {
    auto m = float4x4::identity();
    m[0].x = s.x, m[1].y = s.y, m[2].z = s.z;
    return m;
}
// Rodrigues rotation, axis normalised, angle in radians (luisa::rotation, core/mathematics.h)
inline float4x4 rotation(float3 axis, float angle) {
    auto c = std::cos(angle), s = std::sin(angle);
    auto a = normalize(axis);
    auto t = (1.0f - c) * a;
    float4x4 m;
    m[0] = {c + t.x * a.x, t.x * a.y + s * a.z, t.x * a.z - s * a.y, 0.f};
    m[1] = {t.y * a.x - s * a.z, c + t.y * a.y, t.y * a.z + s * a.x, 0.f};
    m[2] = {t.z * a.x + s * a.y, t.z * a.y - s * a.x, c + t.z * a.z, 0.f};
    m[3] = {0.f, 0.f, 0.f, 1.f};
    return m;
}
inline float radians(float deg) { return deg * (3.14159265358979323846f / 180.f); }

inline uint32_t next_pow2(uint32_t v) {
    v--;
    v |= v >> 1, v |= v >> 2, v |= v >> 4, v |= v >> 8, v |= v >> 16;
    return v + 1;
}

}// namespace lr
