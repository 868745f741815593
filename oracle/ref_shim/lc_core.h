// lc_core.h -- stand-ins for luisa/core (stl aliases, logging, fmt-style formatting, platform macros).
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim; see lc_types.h).
#pragma once

#include "lc_types.h"

#include <string>
#include <string_view>
#include <vector>
#include <span>
#include <memory>
#include <optional>
#include <variant>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <map>
#include <sstream>
#include <filesystem>
#include <cstdio>
#include <cstdlib>
#include <cassert>
#include <mutex>
#include <numeric>
#include <tuple>
#include <bitset>
#include <numbers>
#include <random>
#include <fstream>

// ---- fmt::format: "{}" substitution (format specs inside the braces are ignored) ------------------------------------------
namespace fmt {
namespace detail {
template<typename T>
inline void put(std::ostringstream &os, const T &v) {
    if constexpr (std::is_same_v<T, std::filesystem::path>) {
        os << v.string();
    } else if constexpr (std::is_enum_v<T>) {
        os << static_cast<long long>(v);
    } else if constexpr (requires { os << v; }) {
        os << v;
    } else if constexpr (luisa::is_vector_v<T>) {
        os << "(";
        for (size_t i = 0; i < T::dimension; i++) { os << (i ? ", " : "") << v[i]; }
        os << ")";
    } else {
        os << "<?>";
    }
}
inline void format_to(std::ostringstream &os, std::string_view f) { os << f; }
template<typename A, typename... R>
inline void format_to(std::ostringstream &os, std::string_view f, const A &a, const R &...rest) {
    auto open = f.find('{');
    if (open == std::string_view::npos) { os << f; return; }
    if (open + 1 < f.size() && f[open + 1] == '{') {
        os << f.substr(0, open + 1);
        format_to(os, f.substr(open + 2), a, rest...);
        return;
    }
    auto close = f.find('}', open);
    os << f.substr(0, open);
    put(os, a);
    format_to(os, close == std::string_view::npos ? std::string_view{} : f.substr(close + 1), rest...);
}
}// namespace detail
template<typename... Args>
[[nodiscard]] inline std::string format(std::string_view f, const Args &...args) {
    std::ostringstream os;
    detail::format_to(os, f, args...);
    return os.str();
}
}// namespace fmt

namespace luisa {

using std::string;
using std::string_view;
// luisa::vector is EASTL's: vector<bool> is a plain array of bool (sdl/scene_node_desc.h takes spans of it)
template<typename T>
class vector : public std::vector<T> {
public:
    using std::vector<T>::vector;
    vector() noexcept = default;
    vector(const std::vector<T> &v) : std::vector<T>{v} {}
    vector(std::vector<T> &&v) noexcept : std::vector<T>{std::move(v)} {}
    [[nodiscard]] auto cbegin() const noexcept { return this->begin(); }
    [[nodiscard]] auto cend() const noexcept { return this->end(); }
};
template<>
class vector<bool> {
    std::unique_ptr<bool[]> _data;
    size_t _size{0u}, _capacity{0u};

public:
    using value_type = bool;
    vector() noexcept = default;
    vector(std::initializer_list<bool> l) { for (auto b : l) { push_back(b); } }
    explicit vector(size_t n, bool v = false) { resize(n, v); }
    vector(const vector &o) { for (auto b : o) { push_back(b); } }
    vector(vector &&) noexcept = default;
    vector &operator=(const vector &o) { if (this != &o) { clear(); for (auto b : o) { push_back(b); } } return *this; }
    vector &operator=(vector &&) noexcept = default;
    void reserve(size_t n) {
        if (n <= _capacity) { return; }
        auto p = std::make_unique<bool[]>(n);
        std::copy_n(_data.get(), _size, p.get());
        _data = std::move(p), _capacity = n;
    }
    void resize(size_t n, bool v = false) { reserve(n); for (auto i = _size; i < n; i++) { _data[i] = v; } _size = n; }
    void push_back(bool b) { if (_size == _capacity) { reserve(_capacity ? _capacity * 2u : 8u); } _data[_size++] = b; }
    bool &emplace_back(bool b = false) { push_back(b); return _data[_size - 1u]; }
    void clear() noexcept { _size = 0u; }
    void pop_back() noexcept { _size--; }
    [[nodiscard]] size_t size() const noexcept { return _size; }
    [[nodiscard]] bool empty() const noexcept { return _size == 0u; }
    [[nodiscard]] bool *data() noexcept { return _data.get(); }
    [[nodiscard]] const bool *data() const noexcept { return _data.get(); }
    [[nodiscard]] bool &operator[](size_t i) noexcept { return _data[i]; }
    [[nodiscard]] const bool &operator[](size_t i) const noexcept { return _data[i]; }
    [[nodiscard]] bool *begin() noexcept { return _data.get(); }
    [[nodiscard]] bool *end() noexcept { return _data.get() + _size; }
    [[nodiscard]] const bool *begin() const noexcept { return _data.get(); }
    [[nodiscard]] const bool *end() const noexcept { return _data.get() + _size; }
    [[nodiscard]] const bool *cbegin() const noexcept { return begin(); }
    [[nodiscard]] const bool *cend() const noexcept { return end(); }
    [[nodiscard]] bool &front() noexcept { return _data[0]; }
    [[nodiscard]] bool &back() noexcept { return _data[_size - 1u]; }
    [[nodiscard]] const bool &front() const noexcept { return _data[0]; }
    [[nodiscard]] const bool &back() const noexcept { return _data[_size - 1u]; }
};
template<typename T, size_t E = std::dynamic_extent>
class span {// luisa::span (EASTL): pointer + size, with cbegin / cend (std::span of C++20 has none)
    T *_p{nullptr};
    size_t _n{0u};

public:
    using element_type = T;
    using value_type = std::remove_cv_t<T>;
    constexpr span() noexcept = default;
    constexpr span(T *p, size_t n) noexcept : _p{p}, _n{n} {}
    constexpr span(T *b, T *e) noexcept : _p{b}, _n{static_cast<size_t>(e - b)} {}
    template<typename C>
        requires requires(C &c) { { c.data() } -> std::convertible_to<T *>; c.size(); }
    constexpr span(C &&c) noexcept : _p{c.data()}, _n{c.size()} {}
    template<size_t N>
    constexpr span(T (&a)[N]) noexcept : _p{a}, _n{N} {}
    [[nodiscard]] constexpr T *data() const noexcept { return _p; }
    [[nodiscard]] constexpr size_t size() const noexcept { return _n; }
    [[nodiscard]] constexpr size_t size_bytes() const noexcept { return _n * sizeof(T); }
    [[nodiscard]] constexpr bool empty() const noexcept { return _n == 0u; }
    [[nodiscard]] constexpr T *begin() const noexcept { return _p; }
    [[nodiscard]] constexpr T *end() const noexcept { return _p + _n; }
    [[nodiscard]] constexpr const T *cbegin() const noexcept { return _p; }
    [[nodiscard]] constexpr const T *cend() const noexcept { return _p + _n; }
    [[nodiscard]] constexpr T &operator[](size_t i) const noexcept { return _p[i]; }
    [[nodiscard]] constexpr T &front() const noexcept { return _p[0]; }
    [[nodiscard]] constexpr T &back() const noexcept { return _p[_n - 1u]; }
    [[nodiscard]] constexpr span subspan(size_t o, size_t n = std::dynamic_extent) const noexcept {
        return {_p + o, n == std::dynamic_extent ? _n - o : n};
    }
    [[nodiscard]] constexpr span first(size_t n) const noexcept { return {_p, n}; }
    [[nodiscard]] constexpr span last(size_t n) const noexcept { return {_p + _n - n, n}; }
};
template<typename C>
span(C &c) -> span<std::remove_reference_t<decltype(*c.data())>>;
template<typename C>
span(const C &c) -> span<std::remove_reference_t<decltype(*c.data())>>;
template<typename T>
span(T *, size_t) -> span<T>;
using std::unique_ptr;
using std::shared_ptr;
using std::weak_ptr;
using std::make_unique;
using std::make_shared;
using std::optional;
using std::nullopt;
using std::make_optional;
using std::variant;
using std::get;
using std::get_if;
using std::holds_alternative;
using std::visit;
using std::function;
using std::move;
using std::monostate;
using std::pair;
using std::make_pair;
using std::to_string;
using std::unordered_set;
using std::map;
using fmt::format;

struct string_hash {
    using is_transparent = void;
    [[nodiscard]] size_t operator()(std::string_view s) const noexcept { return std::hash<std::string_view>{}(s); }
    [[nodiscard]] size_t operator()(const std::string &s) const noexcept { return std::hash<std::string_view>{}(s); }
    [[nodiscard]] size_t operator()(const char *s) const noexcept { return std::hash<std::string_view>{}(s); }
};
template<typename T>
struct hash : std::hash<T> {};
template<>
struct hash<std::string> : string_hash {};
template<>
struct hash<std::string_view> : string_hash {};

[[nodiscard]] inline uint64_t hash64(const void *p, size_t n, uint64_t seed) noexcept {// FNV-1a; only used for cache keys
    auto h = 1469598103934665603ull ^ seed;
    for (size_t i = 0; i < n; i++) { h = (h ^ static_cast<const unsigned char *>(p)[i]) * 1099511628211ull; }
    return h;
}
[[nodiscard]] inline uint64_t hash64(std::string_view s, uint64_t seed = 19980810ull) noexcept { return hash64(s.data(), s.size(), seed); }
template<typename T>
[[nodiscard]] inline uint64_t hash_value(const T &v, uint64_t seed = 19980810ull) noexcept {
    if constexpr (requires { std::string_view{v}; }) { return hash64(std::string_view{v}, seed); }
    else { return hash64(&v, sizeof(T), seed); }
}

// std::unordered_map with the heterogeneous string lookups / emplacements luisa's (EASTL / unordered_dense based) map allows
template<typename K, typename V, typename Hash = std::conditional_t<std::is_same_v<K, std::string>, string_hash, std::hash<K>>,
         typename Eq = std::equal_to<>>
class unordered_map : public std::unordered_map<K, V, Hash, Eq> {
    using Base = std::unordered_map<K, V, Hash, Eq>;

public:
    using Base::Base;
    template<typename Key, typename... Args>
    auto try_emplace(Key &&key, Args &&...args) { return Base::try_emplace(K{std::forward<Key>(key)}, std::forward<Args>(args)...); }
    template<typename Key, typename... Args>
    auto emplace(Key &&key, Args &&...args) { return Base::emplace(K{std::forward<Key>(key)}, std::forward<Args>(args)...); }
    template<typename Key>
    [[nodiscard]] auto find(const Key &key) { return Base::find(K{key}); }
    template<typename Key>
    [[nodiscard]] auto find(const Key &key) const { return Base::find(K{key}); }
    template<typename Key>
    [[nodiscard]] bool contains(const Key &key) const { return Base::find(K{key}) != Base::end(); }
};

template<typename K, typename V, size_t N, typename Cmp = std::less<>>
class fixed_map : public std::map<K, V, Cmp> {
public:
    using std::map<K, V, Cmp>::map;
};
template<typename T, size_t N, bool = true>
using fixed_vector = vector<T>;

template<typename T, typename... Args>
[[nodiscard]] inline T *new_with_allocator(Args &&...args) { return new T(std::forward<Args>(args)...); }
template<typename T>
inline void delete_with_allocator(T *p) noexcept { delete p; }
template<typename T>
[[nodiscard]] inline T *allocate_with_allocator(size_t n) noexcept { return static_cast<T *>(::operator new(n * sizeof(T), std::align_val_t{alignof(T)})); }
template<typename T>
inline void deallocate_with_allocator(T *p) noexcept { ::operator delete(p, std::align_val_t{alignof(T)}); }

template<typename T, bool = true, bool = true>
class Pool {// luisa::Pool: object pool; objects live until the pool dies
    std::vector<std::unique_ptr<T>> _objects;

public:
    template<typename... A>
    [[nodiscard]] T *create(A &&...a) { return _objects.emplace_back(std::make_unique<T>(std::forward<A>(a)...)).get(); }
    void destroy(T *) noexcept {}
};

template<typename... T>
constexpr bool always_false_v = false;

class spin_mutex {// luisa::spin_mutex
    std::mutex _m;

public:
    void lock() noexcept { _m.lock(); }
    void unlock() noexcept { _m.unlock(); }
    bool try_lock() noexcept { return _m.try_lock(); }
};

template<typename E>
[[nodiscard]] constexpr auto to_underlying(E e) noexcept { return static_cast<std::underlying_type_t<E>>(e); }

template<typename F>
class LazyConstructor {
    mutable F _f;

public:
    explicit LazyConstructor(F f) noexcept : _f{std::move(f)} {}
    [[nodiscard]] operator auto() const noexcept { return _f(); }
};
template<typename F>
[[nodiscard]] inline auto lazy_construct(F f) noexcept { return LazyConstructor<F>{std::move(f)}; }

template<typename T>
[[nodiscard]] inline auto align(T v, size_t a) noexcept { return (v + (a - 1u)) / a * a; }

}// namespace luisa

namespace fmt {
template<typename T>
[[nodiscard]] inline const void *ptr(T p) noexcept { return static_cast<const void *>(p); }
}// namespace fmt

namespace eastl {
using std::make_pair;
using std::pair;
}// namespace eastl

// ---- logging: warnings to stderr (silenced by LC_SHIM_QUIET), errors abort as LUISA_ERROR does -----------------------------
namespace luisa::detail {
inline bool &shim_quiet() noexcept {
    static bool q = std::getenv("LC_SHIM_VERBOSE") == nullptr;
    return q;
}
template<typename... Args>
inline void shim_log(const char *level, std::string_view f, const Args &...args) {
    if (shim_quiet() && level[0] != 'E') { return; }
    std::fprintf(stderr, "[libref %s] %s\n", level, fmt::format(f, args...).c_str());
}
template<typename... Args>
[[noreturn]] inline void shim_error(std::string_view f, const Args &...args) {
    shim_log("E", f, args...);
    std::abort();
}
}// namespace luisa::detail

#define LUISA_VERBOSE(...) ::luisa::detail::shim_log("V", __VA_ARGS__)
#define LUISA_VERBOSE_WITH_LOCATION(...) ::luisa::detail::shim_log("V", __VA_ARGS__)
#define LUISA_INFO(...) ::luisa::detail::shim_log("I", __VA_ARGS__)
#define LUISA_INFO_WITH_LOCATION(...) ::luisa::detail::shim_log("I", __VA_ARGS__)
#define LUISA_WARNING(...) ::luisa::detail::shim_log("W", __VA_ARGS__)
#define LUISA_WARNING_WITH_LOCATION(...) ::luisa::detail::shim_log("W", __VA_ARGS__)
#define LUISA_ERROR(...) ::luisa::detail::shim_error(__VA_ARGS__)
#define LUISA_ERROR_WITH_LOCATION(...) ::luisa::detail::shim_error(__VA_ARGS__)
#define LUISA_NOT_IMPLEMENTED() ::luisa::detail::shim_error("not implemented: {}", __func__)
#define LUISA_ASSERT(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) { ::luisa::detail::shim_error(__VA_ARGS__); }  \
    } while (false)

#define LUISA_NOEXCEPT noexcept
#define LUISA_EXPORT_API extern "C" __attribute__((visibility("default")))
#define LUISA_IMPORT_API extern "C"
#define LUISA_FORCE_INLINE inline __attribute__((always_inline))
#define LUISA_NEVER_INLINE __attribute__((noinline))
