// lc_types.h -- vector / matrix value types and the math builtins of the scalar LuisaCompute stand-in.
//
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim).  LuisaCompute (src/compute, an unpinned submodule) is absent from the
// reference snapshot, so the reference's render code cannot be compiled as shipped.  This directory provides a SCALAR,
// eagerly evaluated stand-in for the handful of LuisaCompute names that code is written in (`Float`, `Expr<>`, `$if`,
// `ite`, `def`, `Callable`, `make_float3` ...): a DSL "variable" is a plain C++ value and a DSL statement executes
// immediately.  With it the reference's OWN sources under /root/reference/src compile in place into oracle/_ref/libref.so
// (recipe: oracle/Makefile.ref), which tests/test_oracle_vs_ref.py uses to pin oracle/ to the reference's arithmetic.
// Nothing here is reference code; nothing in the product path includes it.
//
// Builtin semantics are restated from LuisaCompute's published meaning and kept IDENTICAL to oracle/oracle_math.h so
// that any difference between oracle and libref comes from the render code, not from the builtins:
// sign(x) = copysign(1, x), fract(x) = x - floor(x), lerp(a, b, t) = a + t * (b - a), saturate = clamp(x, 0, 1),
// normalize(v) = v * (1 / sqrt(dot(v, v))), dot = left-to-right sum, reflect(i, n) = i - 2 * dot(n, i) * n.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <algorithm>
#include <array>
#include <limits>
#include <type_traits>
#include <bit>

#include "lc_swizzles.inl.h"

namespace luisa {

using uint = unsigned int;
using ushort = unsigned short;
using ulong = unsigned long long;
using slong = long long;
using uchar = unsigned char;
using half = unsigned short;// storage only; never computed with here

template<typename T, size_t N>
struct Vector;

namespace detail {
template<typename T, size_t N>
constexpr size_t vector_alignment_v = sizeof(T) * (N == 3 ? 4 : N) > 16 ? 16 : sizeof(T) * (N == 3 ? 4 : N);
}

template<typename T>
struct alignas(detail::vector_alignment_v<T, 2>) Vector<T, 2> {
    using value_type = T;
    static constexpr size_t dimension = 2;
    T x{}, y{};
    constexpr Vector() noexcept = default;
    explicit constexpr Vector(T s) noexcept : x{s}, y{s} {}
    constexpr Vector(T x, T y) noexcept : x{x}, y{y} {}
    template<typename U>
        requires(!std::is_same_v<U, T>)
    explicit constexpr Vector(Vector<U, 2> v) noexcept : x{static_cast<T>(v.x)}, y{static_cast<T>(v.y)} {}
    [[nodiscard]] constexpr T &operator[](size_t i) noexcept { return i == 0 ? x : y; }
    [[nodiscard]] constexpr const T &operator[](size_t i) const noexcept { return i == 0 ? x : y; }
    LC_SHIM_SWIZZLES_2;
};

template<typename T>
struct alignas(detail::vector_alignment_v<T, 3>) Vector<T, 3> {
    using value_type = T;
    static constexpr size_t dimension = 3;
    T x{}, y{}, z{};
    constexpr Vector() noexcept = default;
    explicit constexpr Vector(T s) noexcept : x{s}, y{s}, z{s} {}
    constexpr Vector(T x, T y, T z) noexcept : x{x}, y{y}, z{z} {}
    template<typename U>
        requires(!std::is_same_v<U, T>)
    explicit constexpr Vector(Vector<U, 3> v) noexcept : x{static_cast<T>(v.x)}, y{static_cast<T>(v.y)}, z{static_cast<T>(v.z)} {}
    [[nodiscard]] constexpr T &operator[](size_t i) noexcept { return i == 0 ? x : (i == 1 ? y : z); }
    [[nodiscard]] constexpr const T &operator[](size_t i) const noexcept { return i == 0 ? x : (i == 1 ? y : z); }
    LC_SHIM_SWIZZLES_3;
};

template<typename T>
struct alignas(detail::vector_alignment_v<T, 4>) Vector<T, 4> {
    using value_type = T;
    static constexpr size_t dimension = 4;
    T x{}, y{}, z{}, w{};
    constexpr Vector() noexcept = default;
    explicit constexpr Vector(T s) noexcept : x{s}, y{s}, z{s}, w{s} {}
    constexpr Vector(T x, T y, T z, T w) noexcept : x{x}, y{y}, z{z}, w{w} {}
    template<typename U>
        requires(!std::is_same_v<U, T>)
    explicit constexpr Vector(Vector<U, 4> v) noexcept
        : x{static_cast<T>(v.x)}, y{static_cast<T>(v.y)}, z{static_cast<T>(v.z)}, w{static_cast<T>(v.w)} {}
    [[nodiscard]] constexpr T &operator[](size_t i) noexcept { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    [[nodiscard]] constexpr const T &operator[](size_t i) const noexcept { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
    LC_SHIM_SWIZZLES_4;
};

#define LC_SHIM_VECTOR_ALIASES(T)   \
    using T##2 = Vector<T, 2>;      \
    using T##3 = Vector<T, 3>;      \
    using T##4 = Vector<T, 4>;
LC_SHIM_VECTOR_ALIASES(bool)
LC_SHIM_VECTOR_ALIASES(float)
LC_SHIM_VECTOR_ALIASES(int)
LC_SHIM_VECTOR_ALIASES(uint)
LC_SHIM_VECTOR_ALIASES(short)
LC_SHIM_VECTOR_ALIASES(ushort)
LC_SHIM_VECTOR_ALIASES(slong)
LC_SHIM_VECTOR_ALIASES(ulong)
LC_SHIM_VECTOR_ALIASES(half)
#undef LC_SHIM_VECTOR_ALIASES

static_assert(sizeof(float3) == 16 && alignof(float3) == 16 && sizeof(float2) == 8 && sizeof(float4) == 16);

template<typename T>
using sid = std::type_identity_t<T>;// keeps scalar operands out of template deduction so that Expr<float> / literals convert

// ---- element-wise operators ------------------------------------------------------------------------------------------------
#define LC_SHIM_APPLY1(N, expr)                                           \
    if constexpr (N == 2) { return R{expr(x), expr(y)}; }                 \
    else if constexpr (N == 3) { return R{expr(x), expr(y), expr(z)}; }   \
    else { return R{expr(x), expr(y), expr(z), expr(w)}; }

#define LC_SHIM_BINARY_OP(op, Ret)                                                                                     \
    template<typename T, size_t N>                                                                                     \
    [[nodiscard]] constexpr auto operator op(Vector<T, N> a, Vector<T, N> b) noexcept {                                \
        using R = Vector<Ret, N>;                                                                                      \
        if constexpr (N == 2) { return R{a.x op b.x, a.y op b.y}; }                                                    \
        else if constexpr (N == 3) { return R{a.x op b.x, a.y op b.y, a.z op b.z}; }                                   \
        else { return R{a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; }                                             \
    }                                                                                                                  \
    template<typename T, size_t N>                                                                                     \
    [[nodiscard]] constexpr auto operator op(Vector<T, N> a, sid<T> b) noexcept { return a op Vector<T, N>{b}; }       \
    template<typename T, size_t N>                                                                                     \
    [[nodiscard]] constexpr auto operator op(sid<T> a, Vector<T, N> b) noexcept { return Vector<T, N>{a} op b; }

#define LC_SHIM_ARITH_OP(op)                                                                                           \
    LC_SHIM_BINARY_OP(op, T)                                                                                           \
    template<typename T, size_t N>                                                                                     \
    constexpr Vector<T, N> &operator op##=(Vector<T, N> &a, Vector<T, N> b) noexcept { return a = a op b; }            \
    template<typename T, size_t N>                                                                                     \
    constexpr Vector<T, N> &operator op##=(Vector<T, N> &a, sid<T> b) noexcept { return a = a op Vector<T, N>{b}; }

LC_SHIM_ARITH_OP(+)
LC_SHIM_ARITH_OP(-)
LC_SHIM_ARITH_OP(*)
LC_SHIM_ARITH_OP(/)
LC_SHIM_ARITH_OP(%)
LC_SHIM_ARITH_OP(&)
LC_SHIM_ARITH_OP(|)
LC_SHIM_ARITH_OP(^)
LC_SHIM_ARITH_OP(<<)
LC_SHIM_ARITH_OP(>>)
LC_SHIM_BINARY_OP(==, bool)
LC_SHIM_BINARY_OP(!=, bool)
LC_SHIM_BINARY_OP(<, bool)
LC_SHIM_BINARY_OP(>, bool)
LC_SHIM_BINARY_OP(<=, bool)
LC_SHIM_BINARY_OP(>=, bool)
LC_SHIM_BINARY_OP(&&, bool)
LC_SHIM_BINARY_OP(||, bool)
#undef LC_SHIM_ARITH_OP
#undef LC_SHIM_BINARY_OP

template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> operator-(Vector<T, N> v) noexcept {
    if constexpr (N == 2) { return {-v.x, -v.y}; }
    else if constexpr (N == 3) { return {-v.x, -v.y, -v.z}; }
    else { return {-v.x, -v.y, -v.z, -v.w}; }
}
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> operator+(Vector<T, N> v) noexcept { return v; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> operator~(Vector<T, N> v) noexcept {
    if constexpr (N == 2) { return {~v.x, ~v.y}; }
    else if constexpr (N == 3) { return {~v.x, ~v.y, ~v.z}; }
    else { return {~v.x, ~v.y, ~v.z, ~v.w}; }
}
template<size_t N>
[[nodiscard]] constexpr Vector<bool, N> operator!(Vector<bool, N> v) noexcept {
    if constexpr (N == 2) { return {!v.x, !v.y}; }
    else if constexpr (N == 3) { return {!v.x, !v.y, !v.z}; }
    else { return {!v.x, !v.y, !v.z, !v.w}; }
}

// ---- make_<type>N ----------------------------------------------------------------------------------------------------------
#define LC_SHIM_MAKE_VECTOR(T)                                                                                          \
    [[nodiscard]] constexpr auto make_##T##2() noexcept { return T##2{}; }                                              \
    [[nodiscard]] constexpr auto make_##T##2(T s) noexcept { return T##2{s, s}; }                                       \
    [[nodiscard]] constexpr auto make_##T##2(T x, T y) noexcept { return T##2{x, y}; }                                  \
    template<typename U, size_t N>                                                                                      \
        requires(N >= 2)                                                                                                \
    [[nodiscard]] constexpr auto make_##T##2(Vector<U, N> v) noexcept { return T##2{static_cast<T>(v.x), static_cast<T>(v.y)}; } \
    [[nodiscard]] constexpr auto make_##T##3() noexcept { return T##3{}; }                                              \
    [[nodiscard]] constexpr auto make_##T##3(T s) noexcept { return T##3{s, s, s}; }                                    \
    [[nodiscard]] constexpr auto make_##T##3(T x, T y, T z) noexcept { return T##3{x, y, z}; }                          \
    [[nodiscard]] constexpr auto make_##T##3(T##2 v, T z) noexcept { return T##3{v.x, v.y, z}; }                        \
    [[nodiscard]] constexpr auto make_##T##3(T x, T##2 v) noexcept { return T##3{x, v.x, v.y}; }                        \
    template<typename U, size_t N>                                                                                      \
        requires(N >= 3)                                                                                                \
    [[nodiscard]] constexpr auto make_##T##3(Vector<U, N> v) noexcept {                                                 \
        return T##3{static_cast<T>(v.x), static_cast<T>(v.y), static_cast<T>(v.z)};                                     \
    }                                                                                                                   \
    [[nodiscard]] constexpr auto make_##T##4() noexcept { return T##4{}; }                                              \
    [[nodiscard]] constexpr auto make_##T##4(T s) noexcept { return T##4{s, s, s, s}; }                                 \
    [[nodiscard]] constexpr auto make_##T##4(T x, T y, T z, T w) noexcept { return T##4{x, y, z, w}; }                  \
    [[nodiscard]] constexpr auto make_##T##4(T##2 v, T z, T w) noexcept { return T##4{v.x, v.y, z, w}; }                \
    [[nodiscard]] constexpr auto make_##T##4(T x, T y, T##2 v) noexcept { return T##4{x, y, v.x, v.y}; }                \
    [[nodiscard]] constexpr auto make_##T##4(T x, T##2 v, T w) noexcept { return T##4{x, v.x, v.y, w}; }                \
    [[nodiscard]] constexpr auto make_##T##4(T##2 a, T##2 b) noexcept { return T##4{a.x, a.y, b.x, b.y}; }              \
    [[nodiscard]] constexpr auto make_##T##4(T##3 v, T w) noexcept { return T##4{v.x, v.y, v.z, w}; }                   \
    [[nodiscard]] constexpr auto make_##T##4(T x, T##3 v) noexcept { return T##4{x, v.x, v.y, v.z}; }                   \
    template<typename U>                                                                                                \
    [[nodiscard]] constexpr auto make_##T##4(Vector<U, 4> v) noexcept {                                                 \
        return T##4{static_cast<T>(v.x), static_cast<T>(v.y), static_cast<T>(v.z), static_cast<T>(v.w)};                \
    }
LC_SHIM_MAKE_VECTOR(bool)
LC_SHIM_MAKE_VECTOR(float)
LC_SHIM_MAKE_VECTOR(int)
LC_SHIM_MAKE_VECTOR(uint)
#undef LC_SHIM_MAKE_VECTOR

// ---- matrices (column-major, as luisa::float3x3 / float4x4) ----------------------------------------------------------------
struct float2x2 {
    float2 cols[2];
    constexpr float2x2() noexcept : cols{float2{1.f, 0.f}, float2{0.f, 1.f}} {}
    constexpr float2x2(float2 c0, float2 c1) noexcept : cols{c0, c1} {}
    [[nodiscard]] constexpr float2 &operator[](size_t i) noexcept { return cols[i]; }
    [[nodiscard]] constexpr const float2 &operator[](size_t i) const noexcept { return cols[i]; }
};
struct float3x3 {
    float3 cols[3];
    constexpr float3x3() noexcept : cols{float3{1.f, 0.f, 0.f}, float3{0.f, 1.f, 0.f}, float3{0.f, 0.f, 1.f}} {}
    constexpr float3x3(float3 c0, float3 c1, float3 c2) noexcept : cols{c0, c1, c2} {}
    [[nodiscard]] constexpr float3 &operator[](size_t i) noexcept { return cols[i]; }
    [[nodiscard]] constexpr const float3 &operator[](size_t i) const noexcept { return cols[i]; }
};
struct float4x4 {
    float4 cols[4];
    constexpr float4x4() noexcept
        : cols{float4{1.f, 0.f, 0.f, 0.f}, float4{0.f, 1.f, 0.f, 0.f}, float4{0.f, 0.f, 1.f, 0.f}, float4{0.f, 0.f, 0.f, 1.f}} {}
    constexpr float4x4(float4 c0, float4 c1, float4 c2, float4 c3) noexcept : cols{c0, c1, c2, c3} {}
    [[nodiscard]] constexpr float4 &operator[](size_t i) noexcept { return cols[i]; }
    [[nodiscard]] constexpr const float4 &operator[](size_t i) const noexcept { return cols[i]; }
};

[[nodiscard]] constexpr auto make_float2x2(float2 c0, float2 c1) noexcept { return float2x2{c0, c1}; }
[[nodiscard]] constexpr auto make_float2x2(float m00, float m01, float m10, float m11) noexcept {
    return float2x2{float2{m00, m01}, float2{m10, m11}};
}
[[nodiscard]] constexpr auto make_float3x3() noexcept { return float3x3{}; }
[[nodiscard]] constexpr auto make_float3x3(float s) noexcept {
    return float3x3{float3{s, 0.f, 0.f}, float3{0.f, s, 0.f}, float3{0.f, 0.f, s}};
}
[[nodiscard]] constexpr auto make_float3x3(float3 c0, float3 c1, float3 c2) noexcept { return float3x3{c0, c1, c2}; }
[[nodiscard]] constexpr auto make_float3x3(float m00, float m01, float m02, float m10, float m11, float m12,
                                           float m20, float m21, float m22) noexcept {
    return float3x3{float3{m00, m01, m02}, float3{m10, m11, m12}, float3{m20, m21, m22}};
}
[[nodiscard]] constexpr auto make_float3x3(const float4x4 &m) noexcept {
    return float3x3{make_float3(m[0]), make_float3(m[1]), make_float3(m[2])};
}
[[nodiscard]] constexpr auto make_float3x3(const float3x3 &m) noexcept { return m; }
[[nodiscard]] constexpr auto make_float4x4() noexcept { return float4x4{}; }
[[nodiscard]] constexpr auto make_float4x4(float s) noexcept {
    return float4x4{float4{s, 0.f, 0.f, 0.f}, float4{0.f, s, 0.f, 0.f}, float4{0.f, 0.f, s, 0.f}, float4{0.f, 0.f, 0.f, s}};
}
[[nodiscard]] constexpr auto make_float4x4(float4 c0, float4 c1, float4 c2, float4 c3) noexcept { return float4x4{c0, c1, c2, c3}; }
[[nodiscard]] constexpr auto make_float4x4(float m00, float m01, float m02, float m03, float m10, float m11, float m12, float m13,
                                           float m20, float m21, float m22, float m23, float m30, float m31, float m32, float m33) noexcept {
    return float4x4{float4{m00, m01, m02, m03}, float4{m10, m11, m12, m13}, float4{m20, m21, m22, m23}, float4{m30, m31, m32, m33}};
}
[[nodiscard]] constexpr auto make_float4x4(const float3x3 &m) noexcept {
    return float4x4{make_float4(m[0], 0.f), make_float4(m[1], 0.f), make_float4(m[2], 0.f), float4{0.f, 0.f, 0.f, 1.f}};
}
[[nodiscard]] constexpr auto make_float4x4(const float4x4 &m) noexcept { return m; }

[[nodiscard]] constexpr float2 operator*(const float2x2 &m, float2 v) noexcept { return v.x * m[0] + v.y * m[1]; }
[[nodiscard]] constexpr float3 operator*(const float3x3 &m, float3 v) noexcept { return v.x * m[0] + v.y * m[1] + v.z * m[2]; }
[[nodiscard]] constexpr float4 operator*(const float4x4 &m, float4 v) noexcept {
    return v.x * m[0] + v.y * m[1] + v.z * m[2] + v.w * m[3];
}
[[nodiscard]] constexpr float3x3 operator*(const float3x3 &a, const float3x3 &b) noexcept { return {a * b[0], a * b[1], a * b[2]}; }
[[nodiscard]] constexpr float4x4 operator*(const float4x4 &a, const float4x4 &b) noexcept {
    return {a * b[0], a * b[1], a * b[2], a * b[3]};
}
[[nodiscard]] constexpr float3x3 operator*(const float3x3 &m, float s) noexcept { return {m[0] * s, m[1] * s, m[2] * s}; }
[[nodiscard]] constexpr float3x3 operator*(float s, const float3x3 &m) noexcept { return m * s; }
[[nodiscard]] constexpr float4x4 operator*(const float4x4 &m, float s) noexcept { return {m[0] * s, m[1] * s, m[2] * s, m[3] * s}; }
[[nodiscard]] constexpr float4x4 operator*(float s, const float4x4 &m) noexcept { return m * s; }
[[nodiscard]] constexpr float3x3 operator+(const float3x3 &a, const float3x3 &b) noexcept { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
[[nodiscard]] constexpr float3x3 operator-(const float3x3 &a, const float3x3 &b) noexcept { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
[[nodiscard]] constexpr float4x4 operator-(const float4x4 &a, const float4x4 &b) noexcept {
    return {a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]};
}
[[nodiscard]] constexpr float4x4 operator+(const float4x4 &a, const float4x4 &b) noexcept {
    return {a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
}

// ---- constants (luisa/core/constants.h, mathematics.h) ----------------------------------------------------------------------
constexpr auto pi = 3.14159265358979323846264338327950288f;
constexpr auto pi_over_two = 1.57079632679489661923132169163975144f;
constexpr auto pi_over_four = 0.785398163397448309615660845819875721f;
constexpr auto inv_pi = 0.318309886183790671537767526745028724f;
constexpr auto two_over_pi = 0.636619772367581343075535053490057448f;
constexpr auto sqrt_two = 1.41421356237309504880168872420969808f;
constexpr auto inv_sqrt_two = 0.707106781186547524400844362104849039f;
constexpr auto one_minus_epsilon = 0x1.fffffep-1f;

// ---- scalar builtins: concrete overloads so that Expr<scalar> and literals convert -----------------------------------------
[[nodiscard]] inline float abs(float x) noexcept { return std::fabs(x); }
[[nodiscard]] inline int abs(int x) noexcept { return x < 0 ? -x : x; }
[[nodiscard]] constexpr float max(float a, float b) noexcept { return a > b ? a : (b >= a ? b : (a != a ? b : a)); }
[[nodiscard]] constexpr float min(float a, float b) noexcept { return a < b ? a : (b <= a ? b : (a != a ? b : a)); }
[[nodiscard]] constexpr int max(int a, int b) noexcept { return a > b ? a : b; }
[[nodiscard]] constexpr int min(int a, int b) noexcept { return a < b ? a : b; }
[[nodiscard]] constexpr uint max(uint a, uint b) noexcept { return a > b ? a : b; }
[[nodiscard]] constexpr uint min(uint a, uint b) noexcept { return a < b ? a : b; }
[[nodiscard]] constexpr size_t max(size_t a, size_t b) noexcept { return a > b ? a : b; }
[[nodiscard]] constexpr size_t min(size_t a, size_t b) noexcept { return a < b ? a : b; }
[[nodiscard]] constexpr double max(double a, double b) noexcept { return a > b ? a : b; }
[[nodiscard]] constexpr double min(double a, double b) noexcept { return a < b ? a : b; }
[[nodiscard]] constexpr float clamp(float x, float lo, float hi) noexcept { return min(max(x, lo), hi); }
[[nodiscard]] constexpr int clamp(int x, int lo, int hi) noexcept { return min(max(x, lo), hi); }
[[nodiscard]] constexpr uint clamp(uint x, uint lo, uint hi) noexcept { return min(max(x, lo), hi); }
[[nodiscard]] constexpr double clamp(double x, double lo, double hi) noexcept { return min(max(x, lo), hi); }
[[nodiscard]] constexpr float saturate(float x) noexcept { return clamp(x, 0.f, 1.f); }
[[nodiscard]] constexpr float lerp(float a, float b, float t) noexcept { return a + t * (b - a); }
[[nodiscard]] inline float sign(float x) noexcept { return std::copysign(1.f, x); }
[[nodiscard]] inline float fract(float x) noexcept { return x - std::floor(x); }
[[nodiscard]] constexpr float radians(float deg) noexcept { return deg * (pi / 180.f); }
[[nodiscard]] constexpr float degrees(float rad) noexcept { return rad * (180.f * inv_pi); }
[[nodiscard]] constexpr float fma(float a, float b, float c) noexcept { return a * b + c; }// no hardware fma: -ffp-contract=off tree
[[nodiscard]] inline float sqrt(float x) noexcept { return std::sqrt(x); }
[[nodiscard]] inline double sqrt(double x) noexcept { return std::sqrt(x); }
[[nodiscard]] inline float rsqrt(float x) noexcept { return 1.f / std::sqrt(x); }
[[nodiscard]] inline float sin(float x) noexcept { return std::sin(x); }
[[nodiscard]] inline float cos(float x) noexcept { return std::cos(x); }
[[nodiscard]] inline float tan(float x) noexcept { return std::tan(x); }
[[nodiscard]] inline float asin(float x) noexcept { return std::asin(x); }
[[nodiscard]] inline float acos(float x) noexcept { return std::acos(x); }
[[nodiscard]] inline float atan(float x) noexcept { return std::atan(x); }
[[nodiscard]] inline float atan2(float y, float x) noexcept { return std::atan2(y, x); }
[[nodiscard]] inline float sinh(float x) noexcept { return std::sinh(x); }
[[nodiscard]] inline float cosh(float x) noexcept { return std::cosh(x); }
[[nodiscard]] inline float tanh(float x) noexcept { return std::tanh(x); }
[[nodiscard]] inline float exp(float x) noexcept { return std::exp(x); }
[[nodiscard]] inline float exp2(float x) noexcept { return std::exp2(x); }
[[nodiscard]] inline float log(float x) noexcept { return std::log(x); }
[[nodiscard]] inline float log2(float x) noexcept { return std::log2(x); }
[[nodiscard]] inline float log10(float x) noexcept { return std::log10(x); }
[[nodiscard]] inline float pow(float x, float y) noexcept { return std::pow(x, y); }
[[nodiscard]] inline float floor(float x) noexcept { return std::floor(x); }
[[nodiscard]] inline float ceil(float x) noexcept { return std::ceil(x); }
[[nodiscard]] inline float round(float x) noexcept { return std::round(x); }
[[nodiscard]] inline float trunc(float x) noexcept { return std::trunc(x); }
[[nodiscard]] inline float fmod(float x, float y) noexcept { return std::fmod(x, y); }
[[nodiscard]] inline bool isinf(float x) noexcept { return std::isinf(x); }
[[nodiscard]] inline bool isnan(float x) noexcept { return std::isnan(x); }
[[nodiscard]] constexpr bool any(bool x) noexcept { return x; }
[[nodiscard]] constexpr bool all(bool x) noexcept { return x; }
[[nodiscard]] constexpr float select(float f, float t, bool p) noexcept { return p ? t : f; }
[[nodiscard]] constexpr uint select(uint f, uint t, bool p) noexcept { return p ? t : f; }
[[nodiscard]] constexpr int select(int f, int t, bool p) noexcept { return p ? t : f; }
[[nodiscard]] inline uint popcount(uint x) noexcept { return static_cast<uint>(std::popcount(x)); }
[[nodiscard]] inline uint clz(uint x) noexcept { return static_cast<uint>(std::countl_zero(x)); }
[[nodiscard]] inline uint ctz(uint x) noexcept { return static_cast<uint>(std::countr_zero(x)); }
[[nodiscard]] inline uint reverse(uint x) noexcept {
    x = ((x >> 1u) & 0x55555555u) | ((x & 0x55555555u) << 1u);
    x = ((x >> 2u) & 0x33333333u) | ((x & 0x33333333u) << 2u);
    x = ((x >> 4u) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4u);
    x = ((x >> 8u) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8u);
    return (x >> 16u) | (x << 16u);
}
template<typename T>
    requires std::is_unsigned_v<T>
[[nodiscard]] constexpr T next_pow2(T v) noexcept {
    v--;
    for (auto s = 1u; s < sizeof(T) * 8u; s <<= 1u) { v |= v >> s; }
    return ++v;
}
template<typename To, typename From>
    requires(sizeof(To) == sizeof(From))
[[nodiscard]] inline To bit_cast(const From &from) noexcept {
    To to;
    std::memcpy(&to, &from, sizeof(To));
    return to;
}

// ---- vector builtins -------------------------------------------------------------------------------------------------------
#define LC_SHIM_VECTOR_UNARY(name, Ret)                                              \
    template<typename T, size_t N>                                                   \
    [[nodiscard]] inline auto name(Vector<T, N> v) noexcept {                        \
        using R = Vector<Ret, N>;                                                    \
        if constexpr (N == 2) { return R{name(v.x), name(v.y)}; }                    \
        else if constexpr (N == 3) { return R{name(v.x), name(v.y), name(v.z)}; }    \
        else { return R{name(v.x), name(v.y), name(v.z), name(v.w)}; }               \
    }
LC_SHIM_VECTOR_UNARY(abs, T)
LC_SHIM_VECTOR_UNARY(saturate, T)
LC_SHIM_VECTOR_UNARY(sign, T)
LC_SHIM_VECTOR_UNARY(fract, T)
LC_SHIM_VECTOR_UNARY(sqrt, T)
LC_SHIM_VECTOR_UNARY(rsqrt, T)
LC_SHIM_VECTOR_UNARY(sin, T)
LC_SHIM_VECTOR_UNARY(cos, T)
LC_SHIM_VECTOR_UNARY(tan, T)
LC_SHIM_VECTOR_UNARY(exp, T)
LC_SHIM_VECTOR_UNARY(exp2, T)
LC_SHIM_VECTOR_UNARY(log, T)
LC_SHIM_VECTOR_UNARY(log2, T)
LC_SHIM_VECTOR_UNARY(floor, T)
LC_SHIM_VECTOR_UNARY(ceil, T)
LC_SHIM_VECTOR_UNARY(round, T)
LC_SHIM_VECTOR_UNARY(radians, T)
LC_SHIM_VECTOR_UNARY(degrees, T)
LC_SHIM_VECTOR_UNARY(isinf, bool)
LC_SHIM_VECTOR_UNARY(isnan, bool)
#undef LC_SHIM_VECTOR_UNARY

#define LC_SHIM_VECTOR_BINARY(name)                                                                         \
    template<typename T, size_t N>                                                                          \
    [[nodiscard]] inline auto name(Vector<T, N> a, Vector<T, N> b) noexcept {                               \
        using R = Vector<T, N>;                                                                             \
        if constexpr (N == 2) { return R{name(a.x, b.x), name(a.y, b.y)}; }                                 \
        else if constexpr (N == 3) { return R{name(a.x, b.x), name(a.y, b.y), name(a.z, b.z)}; }            \
        else { return R{name(a.x, b.x), name(a.y, b.y), name(a.z, b.z), name(a.w, b.w)}; }                  \
    }                                                                                                       \
    template<typename T, size_t N>                                                                          \
    [[nodiscard]] inline auto name(Vector<T, N> a, sid<T> b) noexcept { return name(a, Vector<T, N>{b}); }  \
    template<typename T, size_t N>                                                                          \
    [[nodiscard]] inline auto name(sid<T> a, Vector<T, N> b) noexcept { return name(Vector<T, N>{a}, b); }
LC_SHIM_VECTOR_BINARY(max)
LC_SHIM_VECTOR_BINARY(min)
LC_SHIM_VECTOR_BINARY(pow)
LC_SHIM_VECTOR_BINARY(atan2)
LC_SHIM_VECTOR_BINARY(fmod)
#undef LC_SHIM_VECTOR_BINARY

template<typename T, size_t N>
[[nodiscard]] inline auto clamp(Vector<T, N> v, Vector<T, N> lo, Vector<T, N> hi) noexcept { return min(max(v, lo), hi); }
template<typename T, size_t N>
[[nodiscard]] inline auto clamp(Vector<T, N> v, sid<T> lo, sid<T> hi) noexcept { return min(max(v, lo), hi); }
template<typename T, size_t N>
[[nodiscard]] inline auto lerp(Vector<T, N> a, Vector<T, N> b, Vector<T, N> t) noexcept { return a + t * (b - a); }
template<typename T, size_t N>
[[nodiscard]] inline auto lerp(Vector<T, N> a, Vector<T, N> b, sid<T> t) noexcept { return a + t * (b - a); }
template<typename T, size_t N>
[[nodiscard]] inline auto fma(Vector<T, N> a, Vector<T, N> b, Vector<T, N> c) noexcept { return a * b + c; }

template<size_t N>
[[nodiscard]] constexpr bool any(Vector<bool, N> v) noexcept {
    if constexpr (N == 2) { return v.x || v.y; }
    else if constexpr (N == 3) { return v.x || v.y || v.z; }
    else { return v.x || v.y || v.z || v.w; }
}
template<size_t N>
[[nodiscard]] constexpr bool all(Vector<bool, N> v) noexcept {
    if constexpr (N == 2) { return v.x && v.y; }
    else if constexpr (N == 3) { return v.x && v.y && v.z; }
    else { return v.x && v.y && v.z && v.w; }
}
template<size_t N>
[[nodiscard]] constexpr bool none(Vector<bool, N> v) noexcept { return !any(v); }

template<typename T, size_t N>
[[nodiscard]] constexpr T dot(Vector<T, N> a, Vector<T, N> b) noexcept {
    if constexpr (N == 2) { return a.x * b.x + a.y * b.y; }
    else if constexpr (N == 3) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    else { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
}
[[nodiscard]] constexpr float3 cross(float3 a, float3 b) noexcept {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template<size_t N>
[[nodiscard]] inline float length(Vector<float, N> v) noexcept { return std::sqrt(dot(v, v)); }
template<size_t N>
[[nodiscard]] inline float length_squared(Vector<float, N> v) noexcept { return dot(v, v); }
template<size_t N>
[[nodiscard]] inline float distance(Vector<float, N> a, Vector<float, N> b) noexcept { return length(a - b); }
template<size_t N>
[[nodiscard]] inline float distance_squared(Vector<float, N> a, Vector<float, N> b) noexcept { return length_squared(a - b); }
template<size_t N>
[[nodiscard]] inline Vector<float, N> normalize(Vector<float, N> v) noexcept { return v * (1.f / std::sqrt(dot(v, v))); }
[[nodiscard]] inline float3 reflect(float3 i, float3 n) noexcept { return i - 2.f * dot(n, i) * n; }
[[nodiscard]] inline float3 faceforward(float3 n, float3 i, float3 n_ref) noexcept { return dot(n_ref, i) < 0.f ? n : -n; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> select(Vector<T, N> f, Vector<T, N> t, bool p) noexcept { return p ? t : f; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> select(Vector<T, N> f, Vector<T, N> t, Vector<bool, N> p) noexcept {
    Vector<T, N> r;
    for (size_t i = 0; i < N; i++) { r[i] = p[i] ? t[i] : f[i]; }
    return r;
}
template<typename T, size_t N>
[[nodiscard]] constexpr T reduce_sum(Vector<T, N> v) noexcept {
    if constexpr (N == 2) { return v.x + v.y; }
    else if constexpr (N == 3) { return v.x + v.y + v.z; }
    else { return v.x + v.y + v.z + v.w; }
}
template<typename T, size_t N>
[[nodiscard]] constexpr T reduce_max(Vector<T, N> v) noexcept {
    if constexpr (N == 2) { return max(v.x, v.y); }
    else if constexpr (N == 3) { return max(max(v.x, v.y), v.z); }
    else { return max(max(max(v.x, v.y), v.z), v.w); }
}
template<typename T, size_t N>
[[nodiscard]] constexpr T reduce_min(Vector<T, N> v) noexcept {
    if constexpr (N == 2) { return min(v.x, v.y); }
    else if constexpr (N == 3) { return min(min(v.x, v.y), v.z); }
    else { return min(min(min(v.x, v.y), v.z), v.w); }
}

// ---- matrix builtins (luisa/core/mathematics.h: adjugate / determinant, fp32; same expression order as oracle_math.h) --------
[[nodiscard]] constexpr float3x3 transpose(const float3x3 &m) noexcept {
    return {float3{m[0].x, m[1].x, m[2].x}, float3{m[0].y, m[1].y, m[2].y}, float3{m[0].z, m[1].z, m[2].z}};
}
[[nodiscard]] constexpr float4x4 transpose(const float4x4 &m) noexcept {
    return {float4{m[0].x, m[1].x, m[2].x, m[3].x}, float4{m[0].y, m[1].y, m[2].y, m[3].y},
            float4{m[0].z, m[1].z, m[2].z, m[3].z}, float4{m[0].w, m[1].w, m[2].w, m[3].w}};
}
[[nodiscard]] constexpr float determinant(const float3x3 &m) noexcept {
    return m[0].x * (m[1].y * m[2].z - m[2].y * m[1].z) - m[1].x * (m[0].y * m[2].z - m[2].y * m[0].z) +
           m[2].x * (m[0].y * m[1].z - m[1].y * m[0].z);
}
[[nodiscard]] constexpr float3x3 inverse(const float3x3 &m) noexcept {
    const auto one_over_det = 1.0f / determinant(m);
    return {float3{(m[1].y * m[2].z - m[2].y * m[1].z) * one_over_det, (m[2].y * m[0].z - m[0].y * m[2].z) * one_over_det,
                   (m[0].y * m[1].z - m[1].y * m[0].z) * one_over_det},
            float3{(m[2].x * m[1].z - m[1].x * m[2].z) * one_over_det, (m[0].x * m[2].z - m[2].x * m[0].z) * one_over_det,
                   (m[1].x * m[0].z - m[0].x * m[1].z) * one_over_det},
            float3{(m[1].x * m[2].y - m[2].x * m[1].y) * one_over_det, (m[2].x * m[0].y - m[0].x * m[2].y) * one_over_det,
                   (m[0].x * m[1].y - m[1].x * m[0].y) * one_over_det}};
}
[[nodiscard]] constexpr float4x4 inverse(const float4x4 &m) noexcept {// cofactor expansion (glm-style, as luisa::inverse)
    const auto coef00 = m[2].z * m[3].w - m[3].z * m[2].w;
    const auto coef02 = m[1].z * m[3].w - m[3].z * m[1].w;
    const auto coef03 = m[1].z * m[2].w - m[2].z * m[1].w;
    const auto coef04 = m[2].y * m[3].w - m[3].y * m[2].w;
    const auto coef06 = m[1].y * m[3].w - m[3].y * m[1].w;
    const auto coef07 = m[1].y * m[2].w - m[2].y * m[1].w;
    const auto coef08 = m[2].y * m[3].z - m[3].y * m[2].z;
    const auto coef10 = m[1].y * m[3].z - m[3].y * m[1].z;
    const auto coef11 = m[1].y * m[2].z - m[2].y * m[1].z;
    const auto coef12 = m[2].x * m[3].w - m[3].x * m[2].w;
    const auto coef14 = m[1].x * m[3].w - m[3].x * m[1].w;
    const auto coef15 = m[1].x * m[2].w - m[2].x * m[1].w;
    const auto coef16 = m[2].x * m[3].z - m[3].x * m[2].z;
    const auto coef18 = m[1].x * m[3].z - m[3].x * m[1].z;
    const auto coef19 = m[1].x * m[2].z - m[2].x * m[1].z;
    const auto coef20 = m[2].x * m[3].y - m[3].x * m[2].y;
    const auto coef22 = m[1].x * m[3].y - m[3].x * m[1].y;
    const auto coef23 = m[1].x * m[2].y - m[2].x * m[1].y;
    const auto fac0 = float4{coef00, coef00, coef02, coef03};
    const auto fac1 = float4{coef04, coef04, coef06, coef07};
    const auto fac2 = float4{coef08, coef08, coef10, coef11};
    const auto fac3 = float4{coef12, coef12, coef14, coef15};
    const auto fac4 = float4{coef16, coef16, coef18, coef19};
    const auto fac5 = float4{coef20, coef20, coef22, coef23};
    const auto Vec0 = float4{m[1].x, m[0].x, m[0].x, m[0].x};
    const auto Vec1 = float4{m[1].y, m[0].y, m[0].y, m[0].y};
    const auto Vec2 = float4{m[1].z, m[0].z, m[0].z, m[0].z};
    const auto Vec3 = float4{m[1].w, m[0].w, m[0].w, m[0].w};
    const auto inv0 = Vec1 * fac0 - Vec2 * fac1 + Vec3 * fac2;
    const auto inv1 = Vec0 * fac0 - Vec2 * fac3 + Vec3 * fac4;
    const auto inv2 = Vec0 * fac1 - Vec1 * fac3 + Vec3 * fac5;
    const auto inv3 = Vec0 * fac2 - Vec1 * fac4 + Vec2 * fac5;
    constexpr auto sign_a = float4{+1.0f, -1.0f, +1.0f, -1.0f};
    constexpr auto sign_b = float4{-1.0f, +1.0f, -1.0f, +1.0f};
    const auto inv_0 = inv0 * sign_a;
    const auto inv_1 = inv1 * sign_b;
    const auto inv_2 = inv2 * sign_a;
    const auto inv_3 = inv3 * sign_b;
    const auto dot0 = m[0] * float4{inv_0.x, inv_1.x, inv_2.x, inv_3.x};
    const auto dot1 = dot0.x + dot0.y + dot0.z + dot0.w;
    const auto one_over_det = 1.0f / dot1;
    return float4x4{inv_0 * one_over_det, inv_1 * one_over_det, inv_2 * one_over_det, inv_3 * one_over_det};
}

[[nodiscard]] inline float4x4 translation(float3 v) noexcept {
    return {float4{1.f, 0.f, 0.f, 0.f}, float4{0.f, 1.f, 0.f, 0.f}, float4{0.f, 0.f, 1.f, 0.f}, float4{v.x, v.y, v.z, 1.f}};
}
[[nodiscard]] inline float4x4 scaling(float3 s) noexcept {
    return {float4{s.x, 0.f, 0.f, 0.f}, float4{0.f, s.y, 0.f, 0.f}, float4{0.f, 0.f, s.z, 0.f}, float4{0.f, 0.f, 0.f, 1.f}};
}
[[nodiscard]] inline float4x4 scaling(float s) noexcept { return scaling(float3{s, s, s}); }
[[nodiscard]] inline float4x4 rotation(float3 axis, float angle) noexcept {
    auto c = std::cos(angle);
    auto s = std::sin(angle);
    auto a = normalize(axis);
    auto t = (1.0f - c) * a;
    return {float4{c + t.x * a.x, t.x * a.y + s * a.z, t.x * a.z - s * a.y, 0.0f},
            float4{t.y * a.x - s * a.z, c + t.y * a.y, t.y * a.z + s * a.x, 0.0f},
            float4{t.z * a.x + s * a.y, t.z * a.y - s * a.x, c + t.z * a.z, 0.0f},
            float4{0.0f, 0.0f, 0.0f, 1.0f}};
}

template<typename T>
constexpr bool is_vector_v = false;
template<typename T, size_t N>
constexpr bool is_vector_v<Vector<T, N>> = true;
template<typename T>
constexpr bool is_scalar_v = std::is_arithmetic_v<T>;
template<typename T>
struct vector_element { using type = T; };
template<typename T, size_t N>
struct vector_element<Vector<T, N>> { using type = T; };
template<typename T>
using vector_element_t = typename vector_element<std::remove_cvref_t<T>>::type;
template<typename T>
constexpr size_t vector_dimension_v = 1u;
template<typename T, size_t N>
constexpr size_t vector_dimension_v<Vector<T, N>> = N;

}// namespace luisa
