// lc_dsl.h -- the scalar stand-in for LuisaCompute's embedded DSL (namespace luisa::compute).
//
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim; see lc_types.h for what this directory is).
// A DSL variable IS a C++ value (`Float` = float, `Float3` = luisa::float3, `Var<T>` = T); `Expr<T>` is a thin read-only
// view that converts from and to T; `$if / $for / $switch` are the C++ statements; a `Callable` calls its lambda; a
// `Kernel` runs its body once per dispatch id on the calling thread.  One DSL "thread" executes at a time.
#pragma once

#include "lc_types.h"

#include <vector>
#include <functional>
#include <utility>
#include <tuple>
#include <cassert>

namespace luisa::compute {

using namespace luisa;// every builtin lives in ::luisa; the DSL spells them luisa::compute::xxx

// ---- Var / Expr ------------------------------------------------------------------------------------------------------------
// Var<S> for a user struct S carries the methods the LUISA_STRUCT(S, ...) { ... } braces declare (reached through `var->`)
namespace detail {
template<typename S>
struct StructExtension : public S {};// primary: no extension methods; LUISA_STRUCT specialises it
template<typename T>
constexpr bool is_value_type_v = std::is_arithmetic_v<T> || luisa::is_vector_v<T> || std::is_same_v<T, float2x2> ||
                                 std::is_same_v<T, float3x3> || std::is_same_v<T, float4x4>;
}// namespace detail

template<typename S>
struct StructVar : public detail::StructExtension<S> {
    [[nodiscard]] StructVar *operator->() noexcept { return this; }
    [[nodiscard]] const StructVar *operator->() const noexcept { return this; }
};

template<typename T>
struct Expr;
template<typename T>
    requires std::is_arithmetic_v<T>
struct Expr<T> {
    T _v{};
    constexpr Expr() noexcept = default;
    // Constructor TEMPLATES only: as in the DSL, a value of the same type or a literal converts, a Float variable never
    // becomes an Expr<uint> (through float -> uint -> Expr<uint>); overload sets such as sample_alias_table (util/sampling.h)
    // rely on that.
    template<typename U>
        requires(std::is_same_v<U, T> ||
                 (std::is_integral_v<U> && !std::is_same_v<U, bool> && !std::is_same_v<T, bool>) ||
                 (std::is_same_v<T, bool> && std::is_same_v<U, int>) ||// `a & b` of two Bools is an int here
                 (std::is_floating_point_v<U> && std::is_floating_point_v<T>))
    constexpr Expr(U v) noexcept : _v{static_cast<T>(v)} {}
    constexpr operator T() const noexcept { return _v; }
};

namespace detail {
template<typename T, bool value = is_value_type_v<T>>
struct var_of { using type = T; };// scalars, vectors and matrices ARE their values
template<typename T>
struct var_of<T, false> { using type = StructVar<T>; };
}// namespace detail

template<typename T>
using Var = typename detail::var_of<T>::type;
template<typename T>
using VarOf = Var<T>;

template<typename T>
    requires std::is_class_v<T>
struct Expr<T> : public Var<T> {
    constexpr Expr() noexcept = default;
    constexpr Expr(const Var<T> &v) noexcept : Var<T>{v} {}
    [[nodiscard]] constexpr const Expr *operator->() const noexcept { return this; }
};
template<typename T>
Expr(StructVar<T>) -> Expr<T>;

template<typename T>
Expr(T) -> Expr<T>;

template<typename T>
Expr(Expr<T>) -> Expr<T>;

template<typename T>
struct expr_value { using type = T; };
template<typename T>
struct expr_value<Expr<T>> { using type = T; };

template<typename T>
using expr_value_t = typename expr_value<std::remove_cvref_t<T>>::type;

// DSL scalar variables are plain C++ scalars.  The DSL zero-initialises a default-constructed variable (the render code relies
// on it: the default Shape::Handle of Interaction{p} has intersection-offset factor 0, src/base/interaction.h:77-78); here
// that comes from the build: -ftrivial-auto-var-init=zero for automatic objects and a zeroing operator new (ref_api.cpp) for
// heap objects -- see oracle/Makefile.ref.
using Bool = bool;
using Float = float;
using Int = int;
using UInt = uint;
using Short = short;
using UShort = ushort;
using ULong = ulong;
using SLong = slong;
#define LC_SHIM_DSL_VECTOR_ALIASES(Name, T) \
    using Name##2 = Vector<T, 2>;           \
    using Name##3 = Vector<T, 3>;           \
    using Name##4 = Vector<T, 4>;
LC_SHIM_DSL_VECTOR_ALIASES(Bool, bool)
LC_SHIM_DSL_VECTOR_ALIASES(Float, float)
LC_SHIM_DSL_VECTOR_ALIASES(Int, int)
LC_SHIM_DSL_VECTOR_ALIASES(UInt, uint)
#undef LC_SHIM_DSL_VECTOR_ALIASES
using Float2x2 = float2x2;
using Float3x3 = float3x3;
using Float4x4 = float4x4;

// ---- def / cast / as / ite ------------------------------------------------------------------------------------------------
template<typename S>
struct expr_value<StructVar<S>> { using type = S; };
template<typename T>
[[nodiscard]] constexpr auto def(T &&x) noexcept {
    using V = expr_value_t<T>;
    if constexpr (detail::is_value_type_v<V>) { return V(std::forward<T>(x)); }
    else {
        Var<V> v{};
        static_cast<V &>(v) = static_cast<const V &>(x);
        return v;
    }
}
template<typename T, typename... Args>
[[nodiscard]] constexpr Var<T> def(Args &&...args) noexcept {
    if constexpr (detail::is_value_type_v<T>) {
        if constexpr (sizeof...(Args) == 0u) { return T{}; }
        else if constexpr (sizeof...(Args) == 1u) { return T(static_cast<T>(args)...); }
        else { return T{std::forward<Args>(args)...}; }
    } else {
        Var<T> v{};
        static_cast<T &>(v) = T{std::forward<Args>(args)...};
        return v;
    }
}

template<typename To>
[[nodiscard]] constexpr To cast(float v) noexcept { return static_cast<To>(v); }
template<typename To>
[[nodiscard]] constexpr To cast(int v) noexcept { return static_cast<To>(v); }
template<typename To>
[[nodiscard]] constexpr To cast(uint v) noexcept { return static_cast<To>(v); }
template<typename To>
[[nodiscard]] constexpr To cast(bool v) noexcept { return static_cast<To>(v); }
template<typename To>
[[nodiscard]] constexpr To cast(ulong v) noexcept { return static_cast<To>(v); }
template<typename To, typename T, size_t N>
[[nodiscard]] constexpr auto cast(Vector<T, N> v) noexcept { return Vector<To, N>{v}; }

template<typename To, typename From>
[[nodiscard]] inline To as(const From &v) noexcept { return luisa::bit_cast<To>(static_cast<expr_value_t<From>>(v)); }

// ite(p, t, f): scalars by concrete overloads (Expr<> and literals convert), vectors by templates, anything else generically
[[nodiscard]] constexpr float ite(bool p, float t, float f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr int ite(bool p, int t, int f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr uint ite(bool p, uint t, uint f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr bool ite(bool p, bool t, bool f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr float ite(bool p, float t, int f) noexcept { return p ? t : static_cast<float>(f); }
[[nodiscard]] constexpr float ite(bool p, int t, float f) noexcept { return p ? static_cast<float>(t) : f; }
[[nodiscard]] constexpr uint ite(bool p, uint t, int f) noexcept { return p ? t : static_cast<uint>(f); }
[[nodiscard]] constexpr uint ite(bool p, int t, uint f) noexcept { return p ? static_cast<uint>(t) : f; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(bool p, Vector<T, N> t, Vector<T, N> f) noexcept { return p ? t : f; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(bool p, Vector<T, N> t, sid<T> f) noexcept { return p ? t : Vector<T, N>{f}; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(bool p, sid<T> t, Vector<T, N> f) noexcept { return p ? Vector<T, N>{t} : f; }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(Vector<bool, N> p, Vector<T, N> t, Vector<T, N> f) noexcept { return luisa::select(f, t, p); }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(Vector<bool, N> p, Vector<T, N> t, sid<T> f) noexcept { return luisa::select(Vector<T, N>{f}, t, p); }
template<typename T, size_t N>
[[nodiscard]] constexpr Vector<T, N> ite(Vector<bool, N> p, sid<T> t, Vector<T, N> f) noexcept { return luisa::select(f, Vector<T, N>{t}, p); }
template<size_t N>
[[nodiscard]] constexpr Vector<float, N> ite(Vector<bool, N> p, float t, float f) noexcept {
    return luisa::select(Vector<float, N>{f}, Vector<float, N>{t}, p);
}
template<typename S>
[[nodiscard]] constexpr StructVar<S> ite(bool p, const StructVar<S> &t, const StructVar<S> &f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr float3x3 ite(bool p, const float3x3 &t, const float3x3 &f) noexcept { return p ? t : f; }
[[nodiscard]] constexpr float4x4 ite(bool p, const float4x4 &t, const float4x4 &f) noexcept { return p ? t : f; }

// dsl-only spellings of builtins
using luisa::make_float2x2;
using luisa::make_float3x3;
using luisa::make_float4x4;
[[nodiscard]] inline bool isinf(Expr<float> x) noexcept { return std::isinf(static_cast<float>(x)); }
[[nodiscard]] inline bool isnan(Expr<float> x) noexcept { return std::isnan(static_cast<float>(x)); }
[[nodiscard]] inline float face_forward(float) = delete;
inline void unreachable() noexcept {}
template<typename... A>
inline void assume(A &&...) noexcept {}

// ---- arrays ----------------------------------------------------------------------------------------------------------------
template<typename T>
class Local {// a DSL array is a handle: element access through a const Local still yields an lvalue
    static constexpr size_t inline_capacity = 4u;// spectra have 1, 3 or 4 samples: no heap traffic for them
    mutable std::array<Var<T>, inline_capacity> _inline{};
    mutable std::vector<Var<T>> _heap;
    size_t _size{0u};
    [[nodiscard]] Var<T> *_ptr() const noexcept { return _size <= inline_capacity ? _inline.data() : _heap.data(); }

public:
    Local() noexcept = default;
    explicit Local(size_t n) noexcept : _size{n} {
        if (n > inline_capacity) { _heap.resize(n); }
    }
    [[nodiscard]] auto size() const noexcept { return _size; }
    [[nodiscard]] Var<T> &operator[](size_t i) const noexcept { return _ptr()[i]; }
    [[nodiscard]] Var<T> read(size_t i) const noexcept { return _ptr()[i]; }
    void write(size_t i, const Var<T> &v) const noexcept { _ptr()[i] = v; }
    [[nodiscard]] Local *operator->() noexcept { return this; }
    [[nodiscard]] const Local *operator->() const noexcept { return this; }
};

template<typename T, size_t N>
class ArrayVar {
    std::vector<Var<T>> _data;// a vector, not std::array: T may still be incomplete here (LUISA_STRUCT(T) can come later)

public:
    ArrayVar() noexcept : _data(N) {}
    [[nodiscard]] constexpr auto size() const noexcept { return N; }
    [[nodiscard]] Var<T> &operator[](size_t i) noexcept { return _data[i]; }
    [[nodiscard]] const Var<T> &operator[](size_t i) const noexcept { return _data[i]; }
    [[nodiscard]] ArrayVar *operator->() noexcept { return this; }
    [[nodiscard]] const ArrayVar *operator->() const noexcept { return this; }
};
template<typename T, size_t N>
using ArrayFloat = ArrayVar<float, N>;

template<typename T>
class Constant {
    std::vector<T> _data;

public:
    Constant() noexcept = default;
    template<typename C>
        requires requires(const C &c) { c.data(); c.size(); }
    Constant(const C &c) noexcept : _data(c.data(), c.data() + c.size()) {}
    Constant(const T *p, size_t n) noexcept : _data(p, p + n) {}
    Constant(std::initializer_list<T> l) noexcept : _data{l} {}
    [[nodiscard]] auto size() const noexcept { return _data.size(); }
    [[nodiscard]] const T &operator[](size_t i) const noexcept { return _data[i]; }
    [[nodiscard]] T read(size_t i) const noexcept { return _data[i]; }
    [[nodiscard]] const Constant *operator->() const noexcept { return this; }
};
template<typename C>
Constant(const C &) -> Constant<std::remove_cvref_t<decltype(*std::declval<const C &>().data())>>;

// ---- Callable / outline --------------------------------------------------------------------------------------------------
template<typename F>
class Callable {
    F _f;

public:
    Callable(F f) noexcept : _f{std::move(f)} {}
    template<typename... Args>
    decltype(auto) operator()(Args &&...args) const noexcept { return _f(std::forward<Args>(args)...); }
};
template<typename F>
Callable(F) -> Callable<F>;
template<typename R, typename... A>
class Callable<R(A...)> {// the explicit-signature spelling: Callable<uint2(uint2, uint)>
    std::function<Var<R>(Var<A>...)> _f;

public:
    template<typename F>
    Callable(F f) noexcept : _f{std::move(f)} {}
    template<typename... Args>
    decltype(auto) operator()(Args &&...args) const noexcept { return _f(std::forward<Args>(args)...); }
};

template<typename F>
inline void outline(F &&f) noexcept { std::forward<F>(f)(); }

// ---- statements --------------------------------------------------------------------------------------------------------------
namespace detail {
template<typename T>
struct SwitchState {
    T value;
    bool matched{false};
    [[nodiscard]] bool match(T c) noexcept {
        if (!matched && value == c) { return matched = true; }
        return false;
    }
    [[nodiscard]] bool unmatched() const noexcept { return !matched; }
};
template<typename T>
SwitchState(T) -> SwitchState<T>;
template<typename T>
struct Range {// `$for(i, n)`: the loop variable may carry attributes (`$for(i [[maybe_unused]], n)`), hence a range-for
    T b, e, s;
    struct Iter {
        T v, s;
        [[nodiscard]] T operator*() const noexcept { return v; }
        Iter &operator++() noexcept { v += s; return *this; }
        [[nodiscard]] bool operator!=(const Iter &o) const noexcept { return v < o.v; }
    };
    [[nodiscard]] Iter begin() const noexcept { return {b, s}; }
    [[nodiscard]] Iter end() const noexcept { return {e, s}; }
};
template<typename B, typename E>
[[nodiscard]] inline auto make_range(B b, E e) noexcept {
    using T = std::common_type_t<expr_value_t<B>, expr_value_t<E>>;
    return Range<T>{static_cast<T>(b), static_cast<T>(e), T(1)};
}
template<typename B, typename E, typename S>
[[nodiscard]] inline auto make_range(B b, E e, S s) noexcept {
    using T = std::common_type_t<expr_value_t<B>, expr_value_t<E>>;
    return Range<T>{static_cast<T>(b), static_cast<T>(e), static_cast<T>(s)};
}
struct OutlineTag {
    template<typename F>
    void operator%(F &&f) const noexcept { std::forward<F>(f)(); }
};
}// namespace detail

}// namespace luisa::compute

#define LC_SHIM_CAT_(a, b) a##b
#define LC_SHIM_CAT(a, b) LC_SHIM_CAT_(a, b)
#define LC_SHIM_NARGS_(_1, _2, _3, _4, N, ...) N
#define LC_SHIM_NARGS(...) LC_SHIM_NARGS_(__VA_ARGS__, 4, 3, 2, 1)

#define $if(...) if (__VA_ARGS__)
#define $else else
#define $elif(...) else if (__VA_ARGS__)
#define $while(...) while (__VA_ARGS__)
#define $loop for (;;)
#define $break break
#define $continue continue
#define $return(...) return __VA_ARGS__
#define $for_2(i, n) for (::luisa::uint i : ::luisa::compute::detail::Range<::luisa::uint>{0u, static_cast<::luisa::uint>(n), 1u})
#define $for_3(i, b, e) for (auto i : ::luisa::compute::detail::make_range(b, e))
#define $for_4(i, b, e, s) for (auto i : ::luisa::compute::detail::make_range(b, e, s))
#define $for(...) LC_SHIM_CAT($for_, LC_SHIM_NARGS(__VA_ARGS__))(__VA_ARGS__)
#define $switch(...) if (auto lc_shim_switch_state = ::luisa::compute::detail::SwitchState{::luisa::compute::def(__VA_ARGS__)}; true)
#define $case(...) if (lc_shim_switch_state.match(__VA_ARGS__))
#define $default if (lc_shim_switch_state.unmatched())
#define $outline ::luisa::compute::detail::OutlineTag{} % [&]() noexcept
#define $lambda(...) [&] __VA_ARGS__
#define $comment(...) static_cast<void>(0)
#define $comment_with_location(...) static_cast<void>(0)

#define LUISA_DISABLE_DSL_ADDRESS_OF_OPERATOR(...)
#define LUISA_DISABLE_DSL_ADDRESS_OF_MESSAGE ""

// LUISA_STRUCT(S, members...) { extension methods };  -- in the real DSL the braces add methods reached through `var->`:
// they become the body of StructExtension<S>, which derives from S (so the members are in scope) and is the base of Var<S>.
#define LUISA_STRUCT(S, ...) \
    template<>               \
    struct luisa::compute::detail::StructExtension<S> : public S
#define LUISA_BINDING_GROUP(S, ...) \
    template<>                      \
    struct luisa::compute::detail::StructExtension<S> : public S

// ---- if_(...).else_(...) / loop_ / while_ builder spellings (util/u64.h uses them) --------------------------------------------
namespace luisa::compute {
namespace detail {
struct IfResult {
    bool taken;
    template<typename F>
    void else_(F &&f) const noexcept {
        if (!taken) { std::forward<F>(f)(); }
    }
    template<typename F>
    [[nodiscard]] IfResult elif_(bool cond, F &&f) const noexcept {
        if (!taken && cond) { std::forward<F>(f)(); return {true}; }
        return {taken};
    }
};
}// namespace detail
template<typename F>
inline detail::IfResult if_(bool cond, F &&f) noexcept {
    if (cond) { std::forward<F>(f)(); }
    return {cond};
}
}// namespace luisa::compute

