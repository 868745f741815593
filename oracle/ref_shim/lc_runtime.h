// lc_runtime.h -- host-runtime stand-ins (Device, Stream, Buffer, Image, BindlessArray, Mesh, Accel, Kernel / Shader ...)
// of the scalar LuisaCompute shim.
//
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim; see lc_types.h for what this directory is).
// "Device memory" is host memory, a command executes when it is created (C++17 sequences the operands of `<<` left to
// right, so `cb << a.copy_from(p) << accel.build()` still runs in order), a Shader runs its kernel body once per dispatch id
// on the calling thread.  The ray-tracing unit the reference delegates to LuisaCompute's backends (Embree / OptiX: absent) is
// restated here as a per-mesh median-split BVH + Moeller-Trumbore in object space (rays transformed un-normalised, so t is
// shared between spaces) -- the same published interface the oracle restates in oracle/oracle_bvh.h.
#pragma once

#include "lc_core.h"
#include "lc_dsl.h"

#include <future>
#include <chrono>
#include <typeinfo>

namespace luisa::compute {

// =============================================================================================================================
// runtime basics
// =============================================================================================================================
class Resource {
public:
    virtual ~Resource() noexcept = default;
};

struct ShimCommand {};// what copy_from / build / dispatch return: the work is already done

class CommandList {
public:
    [[nodiscard]] bool empty() const noexcept { return true; }
    [[nodiscard]] CommandList &commit() noexcept { return *this; }
    template<typename T>
    CommandList &operator<<(T &&) noexcept { return *this; }
};

class Stream : public Resource {
public:
    struct Commit {};
    struct Synchronize {};
    void synchronize() noexcept {}
    template<typename T>
    Stream &operator<<(T &&cmd) noexcept {
        if constexpr (std::is_invocable_v<T>) { std::forward<T>(cmd)(); }
        return *this;
    }
};
[[nodiscard]] inline auto commit() noexcept { return Stream::Commit{}; }
[[nodiscard]] inline auto synchronize() noexcept { return Stream::Synchronize{}; }

class Clock {
    std::chrono::steady_clock::time_point _t0{std::chrono::steady_clock::now()};

public:
    void tic() noexcept { _t0 = std::chrono::steady_clock::now(); }
    [[nodiscard]] double toc() const noexcept {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - _t0).count();
    }
};

[[nodiscard]] constexpr size_t operator""_M(unsigned long long x) noexcept { return x * 1024u * 1024u; }
[[nodiscard]] constexpr size_t operator""_k(unsigned long long x) noexcept { return x * 1024u; }

// ---- dispatch state of the one DSL thread that runs at a time ---------------------------------------------------------------
namespace detail {
struct DispatchState {
    uint3 id{0u, 0u, 0u};
    uint3 size{1u, 1u, 1u};
};
[[nodiscard]] inline DispatchState &dispatch_state() noexcept {
    static thread_local DispatchState s;
    return s;
}
}// namespace detail
[[nodiscard]] inline uint3 dispatch_id() noexcept { return detail::dispatch_state().id; }
[[nodiscard]] inline uint3 dispatch_size() noexcept { return detail::dispatch_state().size; }
[[nodiscard]] inline uint dispatch_x() noexcept { return detail::dispatch_state().id.x; }
[[nodiscard]] inline uint dispatch_y() noexcept { return detail::dispatch_state().id.y; }
[[nodiscard]] inline uint dispatch_z() noexcept { return detail::dispatch_state().id.z; }
[[nodiscard]] inline uint3 thread_id() noexcept { return uint3{0u, 0u, 0u}; }
[[nodiscard]] inline uint3 block_id() noexcept { return detail::dispatch_state().id; }
[[nodiscard]] inline uint3 block_size() noexcept { return uint3{1u, 1u, 1u}; }
inline void set_block_size(uint, uint = 1u, uint = 1u) noexcept {}
inline void sync_block() noexcept {}
template<typename... A>
inline void device_log(A &&...) noexcept {}

// =============================================================================================================================
// buffers
// =============================================================================================================================
template<typename T>
struct AtomicRef {
    T *p;
    T fetch_add(T v) const noexcept { auto old = *p; *p = old + v; return old; }
    T fetch_sub(T v) const noexcept { auto old = *p; *p = old - v; return old; }
    T fetch_min(T v) const noexcept { auto old = *p; *p = luisa::min(old, v); return old; }
    T fetch_max(T v) const noexcept { auto old = *p; *p = luisa::max(old, v); return old; }
    T exchange(T v) const noexcept { auto old = *p; *p = v; return old; }
    T compare_exchange(T expected, T desired) const noexcept { auto old = *p; if (old == expected) { *p = desired; } return old; }
};
template<typename T>
struct AtomicRef<Vector<T, 2>> { AtomicRef<T> x, y; };
template<typename T>
struct AtomicRef<Vector<T, 3>> { AtomicRef<T> x, y, z; };
template<typename T>
struct AtomicRef<Vector<T, 4>> { AtomicRef<T> x, y, z, w; };

// device-side view of a buffer (the DSL's BufferVar<T> / Expr<Buffer<T>>)
template<typename T>
class BufferVar {
    T *_data{nullptr};
    size_t _size{0u};

public:
    BufferVar() noexcept = default;
    BufferVar(T *data, size_t size) noexcept : _data{data}, _size{size} {}
    [[nodiscard]] VarOf<T> read(size_t i) const noexcept {
        assert(i < _size);
        if constexpr (detail::is_value_type_v<T>) { return _data[i]; }
        else { VarOf<T> v; static_cast<T &>(v) = _data[i]; return v; }
    }
    void write(size_t i, const T &v) const noexcept { assert(i < _size); _data[i] = v; }
    [[nodiscard]] auto atomic(size_t i) const noexcept {
        assert(i < _size);
        if constexpr (luisa::is_vector_v<T>) {
            using E = typename T::value_type;
            if constexpr (T::dimension == 2) { return AtomicRef<T>{{&_data[i].x}, {&_data[i].y}}; }
            else if constexpr (T::dimension == 3) { return AtomicRef<T>{{&_data[i].x}, {&_data[i].y}, {&_data[i].z}}; }
            else { return AtomicRef<T>{{&_data[i].x}, {&_data[i].y}, {&_data[i].z}, {&_data[i].w}}; }
        } else {
            return AtomicRef<T>{&_data[i]};
        }
    }
    [[nodiscard]] size_t size() const noexcept { return _size; }
    [[nodiscard]] size_t device_size() const noexcept { return _size; }
    [[nodiscard]] const BufferVar *operator->() const noexcept { return this; }
};
using BufferFloat = BufferVar<float>;
using BufferFloat2 = BufferVar<float2>;
using BufferFloat3 = BufferVar<float3>;
using BufferFloat4 = BufferVar<float4>;
using BufferUInt = BufferVar<uint>;
using BufferUInt2 = BufferVar<uint2>;
using BufferUInt4 = BufferVar<uint4>;
using BufferInt = BufferVar<int>;

template<typename T>
class Buffer;

template<typename T>
class BufferView {
    std::shared_ptr<std::vector<T>> _storage;
    size_t _offset{0u};
    size_t _size{0u};

public:
    BufferView() noexcept = default;
    BufferView(std::shared_ptr<std::vector<T>> s, size_t offset, size_t size) noexcept
        : _storage{std::move(s)}, _offset{offset}, _size{size} {}
    BufferView(const Buffer<T> &b) noexcept;
    [[nodiscard]] explicit operator bool() const noexcept { return _storage != nullptr; }
    [[nodiscard]] size_t size() const noexcept { return _size; }
    [[nodiscard]] size_t offset() const noexcept { return _offset; }
    [[nodiscard]] size_t size_bytes() const noexcept { return _size * sizeof(T); }
    [[nodiscard]] static constexpr size_t stride() noexcept { return sizeof(T); }
    [[nodiscard]] T *data() const noexcept { return _storage ? _storage->data() + _offset : nullptr; }
    [[nodiscard]] BufferView subview(size_t offset, size_t n) const noexcept { return {_storage, _offset + offset, n}; }
    [[nodiscard]] BufferView view() const noexcept { return *this; }
    ShimCommand copy_from(const void *p) const noexcept { std::memcpy(data(), p, size_bytes()); return {}; }
    ShimCommand copy_to(void *p) const noexcept { std::memcpy(p, data(), size_bytes()); return {}; }
    ShimCommand copy_from(BufferView src) const noexcept { std::memcpy(data(), src.data(), size_bytes()); return {}; }
    [[nodiscard]] BufferVar<T> operator->() const noexcept { return {data(), _size}; }
    [[nodiscard]] operator BufferVar<T>() const noexcept { return {data(), _size}; }
};

template<typename T>
class Buffer : public Resource {
    std::shared_ptr<std::vector<T>> _storage;

public:
    Buffer() noexcept = default;
    explicit Buffer(size_t n) noexcept : _storage{std::make_shared<std::vector<T>>(std::max<size_t>(n, 1u))} {}
    Buffer(Buffer &&) noexcept = default;
    Buffer(const Buffer &) noexcept = delete;
    Buffer &operator=(Buffer &&) noexcept = default;
    Buffer &operator=(const Buffer &) noexcept = delete;
    [[nodiscard]] explicit operator bool() const noexcept { return _storage != nullptr; }
    [[nodiscard]] size_t size() const noexcept { return _storage ? _storage->size() : 0u; }
    [[nodiscard]] size_t size_bytes() const noexcept { return size() * sizeof(T); }
    [[nodiscard]] static constexpr size_t stride() noexcept { return sizeof(T); }
    [[nodiscard]] auto &storage() const noexcept { return _storage; }
    [[nodiscard]] BufferView<T> view() const noexcept { return {_storage, 0u, size()}; }
    [[nodiscard]] BufferView<T> view(size_t offset, size_t n) const noexcept { return {_storage, offset, n}; }
    ShimCommand copy_from(const void *p) const noexcept { return view().copy_from(p); }
    ShimCommand copy_to(void *p) const noexcept { return view().copy_to(p); }
    ShimCommand copy_from(BufferView<T> src) const noexcept { return view().copy_from(src); }
    [[nodiscard]] BufferVar<T> operator->() const noexcept { return {_storage->data(), size()}; }
    [[nodiscard]] operator BufferVar<T>() const noexcept { return {_storage->data(), size()}; }
};
template<typename T>
BufferView<T>::BufferView(const Buffer<T> &b) noexcept : BufferView{b.view()} {}

class BufferArena {
public:
    template<typename D>
    BufferArena(D &, size_t) noexcept {}
    template<typename T>
    [[nodiscard]] BufferView<T> allocate(size_t n) noexcept { return Buffer<T>{n}.view(); }
};
class Command {};

// =============================================================================================================================
// images, samplers
// =============================================================================================================================
enum struct PixelStorage : uint {
    BYTE1, BYTE2, BYTE4,
    SHORT1, SHORT2, SHORT4,
    INT1, INT2, INT4,
    HALF1, HALF2, HALF4,
    FLOAT1, FLOAT2, FLOAT4,
};
[[nodiscard]] constexpr uint pixel_storage_channel_count(PixelStorage s) noexcept {
    switch (s) {
        case PixelStorage::BYTE1: case PixelStorage::SHORT1: case PixelStorage::INT1: case PixelStorage::HALF1: case PixelStorage::FLOAT1: return 1u;
        case PixelStorage::BYTE2: case PixelStorage::SHORT2: case PixelStorage::INT2: case PixelStorage::HALF2: case PixelStorage::FLOAT2: return 2u;
        default: return 4u;
    }
}
[[nodiscard]] constexpr size_t pixel_storage_size(PixelStorage s) noexcept {
    switch (s) {
        case PixelStorage::BYTE1: return 1u; case PixelStorage::BYTE2: return 2u; case PixelStorage::BYTE4: return 4u;
        case PixelStorage::SHORT1: case PixelStorage::HALF1: return 2u;
        case PixelStorage::SHORT2: case PixelStorage::HALF2: return 4u;
        case PixelStorage::SHORT4: case PixelStorage::HALF4: return 8u;
        case PixelStorage::INT1: case PixelStorage::FLOAT1: return 4u;
        case PixelStorage::INT2: case PixelStorage::FLOAT2: return 8u;
        default: return 16u;
    }
}
[[nodiscard]] inline size_t pixel_storage_size(PixelStorage s, uint3 size) noexcept {
    return pixel_storage_size(s) * size.x * size.y * size.z;
}

[[nodiscard]] inline float shim_half_to_float(uint16_t h) noexcept {
    auto sign = (h >> 15u) & 1u, exp = (h >> 10u) & 0x1fu, man = h & 0x3ffu;
    float v;
    if (exp == 0u) { v = std::ldexp(static_cast<float>(man), -24); }
    else if (exp == 31u) { v = man ? std::numeric_limits<float>::quiet_NaN() : std::numeric_limits<float>::infinity(); }
    else { v = std::ldexp(static_cast<float>(man | 0x400u), static_cast<int>(exp) - 25); }
    return sign ? -v : v;
}

class Sampler {
public:
    enum struct Filter : uint8_t { POINT, LINEAR_POINT, LINEAR_LINEAR, ANISOTROPIC };
    enum struct Address : uint8_t { EDGE, REPEAT, MIRROR, ZERO };

private:
    Filter _filter{Filter::POINT};
    Address _address{Address::EDGE};

public:
    constexpr Sampler() noexcept = default;
    constexpr Sampler(Filter f, Address a) noexcept : _filter{f}, _address{a} {}
    [[nodiscard]] constexpr auto filter() const noexcept { return _filter; }
    [[nodiscard]] constexpr auto address() const noexcept { return _address; }
    [[nodiscard]] static constexpr auto point_edge() noexcept { return Sampler{Filter::POINT, Address::EDGE}; }
    [[nodiscard]] static constexpr auto point_repeat() noexcept { return Sampler{Filter::POINT, Address::REPEAT}; }
    [[nodiscard]] static constexpr auto point_mirror() noexcept { return Sampler{Filter::POINT, Address::MIRROR}; }
    [[nodiscard]] static constexpr auto point_zero() noexcept { return Sampler{Filter::POINT, Address::ZERO}; }
    [[nodiscard]] static constexpr auto linear_point_edge() noexcept { return Sampler{Filter::LINEAR_POINT, Address::EDGE}; }
    [[nodiscard]] static constexpr auto linear_point_repeat() noexcept { return Sampler{Filter::LINEAR_POINT, Address::REPEAT}; }
    [[nodiscard]] static constexpr auto linear_point_mirror() noexcept { return Sampler{Filter::LINEAR_POINT, Address::MIRROR}; }
    [[nodiscard]] static constexpr auto linear_point_zero() noexcept { return Sampler{Filter::LINEAR_POINT, Address::ZERO}; }
    [[nodiscard]] static constexpr auto linear_linear_edge() noexcept { return Sampler{Filter::LINEAR_LINEAR, Address::EDGE}; }
    [[nodiscard]] static constexpr auto linear_linear_repeat() noexcept { return Sampler{Filter::LINEAR_LINEAR, Address::REPEAT}; }
    [[nodiscard]] static constexpr auto linear_linear_mirror() noexcept { return Sampler{Filter::LINEAR_LINEAR, Address::MIRROR}; }
    [[nodiscard]] static constexpr auto linear_linear_zero() noexcept { return Sampler{Filter::LINEAR_LINEAR, Address::ZERO}; }
    [[nodiscard]] static constexpr auto anisotropic_repeat() noexcept { return Sampler{Filter::ANISOTROPIC, Address::REPEAT}; }
    [[nodiscard]] static constexpr auto anisotropic_edge() noexcept { return Sampler{Filter::ANISOTROPIC, Address::EDGE}; }
};

struct ImageStorage {// level 0 only (the reference generates no mip chain: textures/image.cpp "TODO")
    PixelStorage storage{PixelStorage::FLOAT4};
    uint2 size{0u, 0u};
    uint mip_levels{1u};
    std::vector<float4> texels;// decoded to float4 on upload

    [[nodiscard]] float4 fetch(int x, int y, Sampler::Address a) const noexcept {
        auto wrap = [a](int v, int n, bool &zero) noexcept {
            switch (a) {
                case Sampler::Address::EDGE: return std::clamp(v, 0, n - 1);
                case Sampler::Address::REPEAT: { auto m = v % n; return m < 0 ? m + n : m; }
                case Sampler::Address::MIRROR: {
                    auto p = 2 * n; auto m = v % p; if (m < 0) { m += p; }
                    return m < n ? m : p - 1 - m;
                }
                default: if (v < 0 || v >= n) { zero = true; return 0; } return v;
            }
        };
        auto zero = false;
        auto xx = wrap(x, static_cast<int>(size.x), zero);
        auto yy = wrap(y, static_cast<int>(size.y), zero);
        if (zero) { return float4{}; }
        return texels[static_cast<size_t>(yy) * size.x + xx];
    }
    // bilinear filtering at texel centres (the convention of every LuisaCompute backend's hardware / ISPC sampler)
    [[nodiscard]] float4 sample(float2 uv, Sampler s) const noexcept {
        auto fx = uv.x * static_cast<float>(size.x);
        auto fy = uv.y * static_cast<float>(size.y);
        if (s.filter() == Sampler::Filter::POINT) {
            return fetch(static_cast<int>(std::floor(fx)), static_cast<int>(std::floor(fy)), s.address());
        }
        fx -= .5f, fy -= .5f;
        auto x0 = std::floor(fx), y0 = std::floor(fy);
        auto tx = fx - x0, ty = fy - y0;
        auto ix = static_cast<int>(x0), iy = static_cast<int>(y0);
        auto c00 = fetch(ix, iy, s.address()), c10 = fetch(ix + 1, iy, s.address());
        auto c01 = fetch(ix, iy + 1, s.address()), c11 = fetch(ix + 1, iy + 1, s.address());
        return (c00 * (1.f - tx) + c10 * tx) * (1.f - ty) + (c01 * (1.f - tx) + c11 * tx) * ty;
    }
    void upload(const void *pixels) noexcept {
        auto n = static_cast<size_t>(size.x) * size.y;
        auto c = pixel_storage_channel_count(storage);
        texels.assign(n, float4{0.f, 0.f, 0.f, 0.f});
        for (size_t i = 0; i < n; i++) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (auto k = 0u; k < c; k++) {
                switch (storage) {
                    case PixelStorage::BYTE1: case PixelStorage::BYTE2: case PixelStorage::BYTE4:
                        v[k] = static_cast<const uint8_t *>(pixels)[i * c + k] * (1.f / 255.f); break;
                    case PixelStorage::SHORT1: case PixelStorage::SHORT2: case PixelStorage::SHORT4:
                        v[k] = static_cast<const uint16_t *>(pixels)[i * c + k] * (1.f / 65535.f); break;
                    case PixelStorage::HALF1: case PixelStorage::HALF2: case PixelStorage::HALF4:
                        v[k] = shim_half_to_float(static_cast<const uint16_t *>(pixels)[i * c + k]); break;
                    case PixelStorage::FLOAT1: case PixelStorage::FLOAT2: case PixelStorage::FLOAT4:
                        v[k] = static_cast<const float *>(pixels)[i * c + k]; break;
                    default: v[k] = static_cast<float>(static_cast<const int *>(pixels)[i * c + k]); break;
                }
            }
            texels[i] = float4{v[0], v[1], v[2], v[3]};
        }
    }
};

template<typename T>
class ImageVar {
    ImageStorage *_s{nullptr};

public:
    ImageVar() noexcept = default;
    explicit ImageVar(ImageStorage *s) noexcept : _s{s} {}
    [[nodiscard]] Vector<T, 4> read(uint2 p) const noexcept { return Vector<T, 4>{_s->texels[static_cast<size_t>(p.y) * _s->size.x + p.x]}; }
    void write(uint2 p, Vector<T, 4> v) const noexcept { _s->texels[static_cast<size_t>(p.y) * _s->size.x + p.x] = float4{v}; }
    [[nodiscard]] uint2 size() const noexcept { return _s->size; }
    [[nodiscard]] const ImageVar *operator->() const noexcept { return this; }
};
using ImageFloat = ImageVar<float>;
using ImageUInt = ImageVar<uint>;

template<typename T>
class Image : public Resource {
    std::shared_ptr<ImageStorage> _s;

public:
    Image() noexcept = default;
    Image(PixelStorage storage, uint2 size, uint mips = 1u) noexcept : _s{std::make_shared<ImageStorage>()} {
        _s->storage = storage, _s->size = size, _s->mip_levels = std::max(mips, 1u);
        _s->texels.assign(static_cast<size_t>(size.x) * size.y, float4{});
        _s->mip_levels = 1u;
    }
    Image(PixelStorage storage, uint w, uint h, uint mips = 1u) noexcept : Image{storage, uint2{w, h}, mips} {}
    Image(Image &&) noexcept = default;
    Image &operator=(Image &&) noexcept = default;
    [[nodiscard]] explicit operator bool() const noexcept { return _s != nullptr; }
    [[nodiscard]] auto size() const noexcept { return _s->size; }
    [[nodiscard]] auto mip_levels() const noexcept { return _s->mip_levels; }
    [[nodiscard]] auto storage() const noexcept { return _s->storage; }
    [[nodiscard]] auto &shared_storage() const noexcept { return _s; }
    [[nodiscard]] const Image &view() const noexcept { return *this; }
    [[nodiscard]] const Image &view(uint) const noexcept { return *this; }
    ShimCommand copy_from(const void *pixels) const noexcept { _s->upload(pixels); return {}; }
    ShimCommand copy_to(void *pixels) const noexcept {
        std::memcpy(pixels, _s->texels.data(), _s->texels.size() * sizeof(float4));// FLOAT4 images only
        return {};
    }
    [[nodiscard]] ImageVar<T> operator->() const noexcept { return ImageVar<T>{_s.get()}; }
    [[nodiscard]] operator ImageVar<T>() const noexcept { return ImageVar<T>{_s.get()}; }
};
template<typename T>
using ImageView = Image<T>;
template<typename T>
class Volume : public Resource {};
template<typename T>
class VolumeView {};

// =============================================================================================================================
// bindless array
// =============================================================================================================================
template<typename T>
class BindlessBuffer {
    const T *_data{nullptr};
    size_t _size{0u};

public:
    BindlessBuffer(const void *data, size_t bytes) noexcept : _data{static_cast<const T *>(data)}, _size{bytes / sizeof(T)} {}
    [[nodiscard]] VarOf<T> read(size_t i) const noexcept {
        if (i >= _size) { LUISA_ERROR("libref: bindless buffer read out of range: index {} of {} elements of {} bytes", i, _size, sizeof(T)); }
        if constexpr (detail::is_value_type_v<T>) { return _data[i]; }
        else { VarOf<T> v; static_cast<T &>(v) = _data[i]; return v; }
    }
    [[nodiscard]] size_t size() const noexcept { return _size; }
    [[nodiscard]] const BindlessBuffer *operator->() const noexcept { return this; }
};

class BindlessTexture2D {
    const ImageStorage *_s;
    Sampler _sampler;

public:
    BindlessTexture2D(const ImageStorage *s, Sampler sampler) noexcept : _s{s}, _sampler{sampler} {}
    [[nodiscard]] float4 sample(float2 uv) const noexcept { return _s->sample(uv, _sampler); }
    [[nodiscard]] float4 sample(float2 uv, float) const noexcept { return _s->sample(uv, _sampler); }
    [[nodiscard]] float4 sample(float2 uv, float2, float2) const noexcept { return _s->sample(uv, _sampler); }
    [[nodiscard]] float4 read(uint2 p) const noexcept { return _s->texels[static_cast<size_t>(p.y) * _s->size.x + p.x]; }
    [[nodiscard]] float4 read(uint2 p, uint) const noexcept { return read(p); }
    [[nodiscard]] uint2 size() const noexcept { return _s->size; }
    [[nodiscard]] uint2 size(uint) const noexcept { return _s->size; }
    [[nodiscard]] const BindlessTexture2D *operator->() const noexcept { return this; }
};
class BindlessTexture3D {
public:
    [[nodiscard]] float4 sample(float3) const noexcept { return float4{}; }
    [[nodiscard]] float4 read(uint3) const noexcept { return float4{}; }
    [[nodiscard]] uint3 size() const noexcept { return uint3{}; }
    [[nodiscard]] const BindlessTexture3D *operator->() const noexcept { return this; }
};

class BindlessVar {
public:
    struct BufferSlot { std::shared_ptr<void> keep; const void *data{nullptr}; size_t bytes{0u}; };
    struct Tex2DSlot { std::shared_ptr<ImageStorage> image; Sampler sampler; };

private:
    const std::vector<BufferSlot> *_buffers;
    const std::vector<Tex2DSlot> *_tex2d;

public:
    BindlessVar(const std::vector<BufferSlot> *b, const std::vector<Tex2DSlot> *t) noexcept : _buffers{b}, _tex2d{t} {}
    template<typename T>
    [[nodiscard]] BindlessBuffer<T> buffer(size_t id) const noexcept {
        assert(id < _buffers->size());
        return {(*_buffers)[id].data, (*_buffers)[id].bytes};
    }
    [[nodiscard]] BindlessTexture2D tex2d(size_t id) const noexcept {
        assert(id < _tex2d->size());
        return {(*_tex2d)[id].image.get(), (*_tex2d)[id].sampler};
    }
    [[nodiscard]] BindlessTexture3D tex3d(size_t) const noexcept { return {}; }
    [[nodiscard]] const BindlessVar *operator->() const noexcept { return this; }
};

class BindlessArray : public Resource {
    std::vector<BindlessVar::BufferSlot> _buffers;
    std::vector<BindlessVar::Tex2DSlot> _tex2d;
    bool _dirty{false};

public:
    BindlessArray() noexcept = default;
    explicit BindlessArray(size_t) noexcept {}
    BindlessArray(BindlessArray &&) noexcept = default;
    BindlessArray &operator=(BindlessArray &&) noexcept = default;
    template<typename T>
    void emplace_on_update(size_t id, BufferView<T> v) noexcept {
        if (_buffers.size() <= id) { _buffers.resize(id + 1u); }
        _buffers[id] = {nullptr, v.data(), v.size_bytes()};
        _keep.emplace_back(std::make_shared<BufferView<T>>(v));
        _dirty = true;
    }
    template<typename T>
    void emplace_on_update(size_t id, const Buffer<T> &b) noexcept { emplace_on_update(id, b.view()); }
    template<typename T>
    void emplace_on_update(size_t id, const Image<T> &image, Sampler sampler) noexcept {
        if (_tex2d.size() <= id) { _tex2d.resize(id + 1u); }
        _tex2d[id] = {image.shared_storage(), sampler};
        _dirty = true;
    }
    template<typename T>
    void emplace_on_update(size_t, const Volume<T> &, Sampler) noexcept { _dirty = true; }
    [[nodiscard]] bool dirty() const noexcept { return _dirty; }
    ShimCommand update() noexcept { _dirty = false; return {}; }
    [[nodiscard]] BindlessVar operator->() const noexcept { return {&_buffers, &_tex2d}; }

private:
    std::vector<std::shared_ptr<void>> _keep;
};

// =============================================================================================================================
// ray tracing
// =============================================================================================================================
struct Triangle {
    uint i0, i1, i2;
};
struct Ray {
    std::array<float, 3> compressed_origin{};
    float compressed_t_min{0.f};
    std::array<float, 3> compressed_direction{};
    float compressed_t_max{0.f};
};
struct SurfaceHit {
    uint inst{~0u};
    uint prim{~0u};
    float2 bary{};
    float committed_ray_t{0.f};
};
using TriangleHit = SurfaceHit;
enum struct HitType : uint { Miss = 0u, Surface = 1u, Triangle = 1u, Procedural = 2u };
struct CommittedHit {
    uint inst{~0u};
    uint prim{~0u};
    float2 bary{};
    uint hit_type{0u};
    float committed_ray_t{0.f};
};
struct AccelOption {
    enum struct UsageHint : uint { FAST_TRACE, FAST_BUILD };
    UsageHint hint{UsageHint::FAST_TRACE};
    bool allow_compaction{true};
    bool allow_update{false};
};
struct AccelTraceOptions {};

namespace detail {
template<>
struct StructExtension<Ray> : public Ray {
    [[nodiscard]] float3 origin() const noexcept { return {compressed_origin[0], compressed_origin[1], compressed_origin[2]}; }
    [[nodiscard]] float3 direction() const noexcept { return {compressed_direction[0], compressed_direction[1], compressed_direction[2]}; }
    [[nodiscard]] float t_min() const noexcept { return compressed_t_min; }
    [[nodiscard]] float t_max() const noexcept { return compressed_t_max; }
    void set_origin(float3 o) noexcept { compressed_origin = {o.x, o.y, o.z}; }
    void set_direction(float3 d) noexcept { compressed_direction = {d.x, d.y, d.z}; }
    void set_t_min(float t) noexcept { compressed_t_min = t; }
    void set_t_max(float t) noexcept { compressed_t_max = t; }
};
template<>
struct StructExtension<SurfaceHit> : public SurfaceHit {
    [[nodiscard]] bool miss() const noexcept { return inst == ~0u; }
    [[nodiscard]] bool hit() const noexcept { return inst != ~0u; }
    [[nodiscard]] float distance() const noexcept { return committed_ray_t; }
};
template<>
struct StructExtension<CommittedHit> : public CommittedHit {
    [[nodiscard]] bool miss() const noexcept { return hit_type == 0u; }
    [[nodiscard]] bool is_triangle() const noexcept { return hit_type == 1u; }
    [[nodiscard]] bool is_surface() const noexcept { return hit_type == 1u; }
    [[nodiscard]] bool is_procedural() const noexcept { return hit_type == 2u; }
    [[nodiscard]] float distance() const noexcept { return committed_ray_t; }
};
}// namespace detail

[[nodiscard]] inline StructVar<Ray> make_ray(float3 origin, float3 direction, float t_min, float t_max) noexcept {
    StructVar<Ray> r;
    r.set_origin(origin), r.set_direction(direction), r.set_t_min(t_min), r.set_t_max(t_max);
    return r;
}
[[nodiscard]] inline StructVar<Ray> make_ray(float3 origin, float3 direction) noexcept {
    return make_ray(origin, direction, 0.f, std::numeric_limits<float>::max());
}

// LuisaCompute `offset_ray_origin` (absent submodule): the integer-ULP offset of Ray Tracing Gems ch. 6, restated from the
// published algorithm -- the same restatement as oracle/oracle.cpp's.
[[nodiscard]] inline float3 offset_ray_origin(float3 p, float3 n) noexcept {
    constexpr auto origin = 1.0f / 32.0f;
    constexpr auto float_scale = 1.0f / 65536.0f;
    constexpr auto int_scale = 256.0f;
    float3 out;
    for (size_t i = 0; i < 3; i++) {
        auto of_i = static_cast<int32_t>(int_scale * n[i]);
        auto pi_bits = luisa::bit_cast<int32_t>(p[i]);
        auto p_i = luisa::bit_cast<float>(pi_bits + (p[i] < 0.f ? -of_i : of_i));
        out[i] = std::abs(p[i]) < origin ? p[i] + float_scale * n[i] : p_i;
    }
    return out;
}
[[nodiscard]] inline float3 offset_ray_origin(float3 p, float3 n, float3 w) noexcept {
    return offset_ray_origin(p, luisa::dot(n, w) < 0.f ? -n : n);
}

class Mesh : public Resource {
    std::shared_ptr<void> _keep_v, _keep_t;
    const std::byte *_vertices{nullptr};
    size_t _vertex_stride{0u};
    size_t _vertex_count{0u};
    const Triangle *_triangles{nullptr};
    size_t _triangle_count{0u};
    struct Node {
        float3 lo, hi;
        uint left;  // inner: index of the left child (right = left + 1); leaf: first primitive
        uint count; // 0 = inner
    };
    std::vector<Node> _nodes;
    std::vector<uint> _prims;

    void _bounds(uint first, uint count, float3 &lo, float3 &hi) const noexcept {
        lo = float3{std::numeric_limits<float>::max()}, hi = float3{-std::numeric_limits<float>::max()};
        for (auto i = first; i < first + count; i++) {
            auto t = _triangles[_prims[i]];
            for (auto v : {t.i0, t.i1, t.i2}) { lo = luisa::min(lo, position(v)), hi = luisa::max(hi, position(v)); }
        }
    }
    void _split(uint node_index) noexcept {
        auto first = _nodes[node_index].left, count = _nodes[node_index].count;
        if (count <= 4u) { return; }
        auto e = _nodes[node_index].hi - _nodes[node_index].lo;
        size_t axis = e.x > e.y ? (e.x > e.z ? 0u : 2u) : (e.y > e.z ? 1u : 2u);
        auto centroid = [&](uint p) noexcept {
            auto t = _triangles[p];
            return position(t.i0)[axis] + position(t.i1)[axis] + position(t.i2)[axis];
        };
        auto mid = first + count / 2u;
        std::nth_element(_prims.begin() + first, _prims.begin() + mid, _prims.begin() + first + count,
                         [&](uint a, uint b) noexcept { return centroid(a) < centroid(b); });
        auto left = static_cast<uint>(_nodes.size());
        _nodes.emplace_back(), _nodes.emplace_back();
        _nodes[left].left = first, _nodes[left].count = mid - first;
        _nodes[left + 1u].left = mid, _nodes[left + 1u].count = first + count - mid;
        _bounds(first, mid - first, _nodes[left].lo, _nodes[left].hi);
        _bounds(mid, first + count - mid, _nodes[left + 1u].lo, _nodes[left + 1u].hi);
        _nodes[node_index].left = left, _nodes[node_index].count = 0u;
        _split(left), _split(left + 1u);
    }

public:
    Mesh() noexcept = default;
    template<typename V>
    Mesh(const Buffer<V> &vertices, const Buffer<Triangle> &triangles, const AccelOption & = {}) noexcept
        : _keep_v{vertices.storage()}, _keep_t{triangles.storage()},
          _vertices{reinterpret_cast<const std::byte *>(vertices.storage()->data())}, _vertex_stride{sizeof(V)},
          _vertex_count{vertices.size()}, _triangles{triangles.storage()->data()}, _triangle_count{triangles.size()} {}
    [[nodiscard]] float3 position(uint v) const noexcept {
        float p[3];
        std::memcpy(p, _vertices + v * _vertex_stride, sizeof(p));
        return {p[0], p[1], p[2]};
    }
    [[nodiscard]] auto triangle_count() const noexcept { return static_cast<uint>(_triangle_count); }
    [[nodiscard]] auto vertex_count() const noexcept { return static_cast<uint>(_vertex_count); }
    ShimCommand build() noexcept {
        _prims.resize(_triangle_count);
        std::iota(_prims.begin(), _prims.end(), 0u);
        _nodes.clear();
        _nodes.emplace_back();
        _nodes[0].left = 0u, _nodes[0].count = static_cast<uint>(_triangle_count);
        _bounds(0u, _nodes[0].count, _nodes[0].lo, _nodes[0].hi);
        _split(0u);
        return {};
    }
    // candidates in BVH order; visit(prim, t, u, v) returns the new t_max (commit) or the old one (skip); stop = any-hit
    template<typename Visit>
    void traverse(float3 o, float3 d, float t_min, float &t_max, Visit &&visit) const noexcept {
        if (_nodes.empty()) { return; }
        auto inv = float3{1.f / d.x, 1.f / d.y, 1.f / d.z};
        uint stack[64];
        auto sp = 0u;
        stack[sp++] = 0u;
        while (sp != 0u) {
            auto &n = _nodes[stack[--sp]];
            auto t0 = (n.lo - o) * inv, t1 = (n.hi - o) * inv;
            auto tn = luisa::min(t0, t1), tf = luisa::max(t0, t1);
            auto near = std::max(std::max(tn.x, tn.y), std::max(tn.z, t_min));
            auto far = std::min(std::min(tf.x, tf.y), std::min(tf.z, t_max));
            if (!(near <= far * 1.0000003f)) { continue; }
            if (n.count == 0u) {
                stack[sp++] = n.left, stack[sp++] = n.left + 1u;
                continue;
            }
            for (auto i = n.left; i < n.left + n.count; i++) {
                auto tri = _triangles[_prims[i]];
                auto p0 = position(tri.i0);
                auto e1 = position(tri.i1) - p0, e2 = position(tri.i2) - p0;
                auto pvec = luisa::cross(d, e2);
                auto det = luisa::dot(e1, pvec);
                if (det == 0.f) { continue; }
                auto inv_det = 1.0f / det;
                auto tvec = o - p0;
                auto u = luisa::dot(tvec, pvec) * inv_det;
                auto qvec = luisa::cross(tvec, e1);
                auto v = luisa::dot(d, qvec) * inv_det;
                auto t = luisa::dot(e2, qvec) * inv_det;
                if (u >= 0.f && v >= 0.f && u + v <= 1.f && t > t_min && t < t_max) {
                    if (visit(_prims[i], t, u, v)) { return; }
                }
            }
        }
    }
};

class SurfaceCandidate {
    StructVar<Ray> _ray;
    StructVar<SurfaceHit> _hit;
    bool _committed{false};

public:
    SurfaceCandidate(const StructVar<Ray> &ray, const StructVar<SurfaceHit> &hit) noexcept : _ray{ray}, _hit{hit} {}
    [[nodiscard]] const StructVar<Ray> &ray() const noexcept { return _ray; }
    [[nodiscard]] const StructVar<SurfaceHit> &hit() const noexcept { return _hit; }
    void commit() noexcept { _committed = true; }
    [[nodiscard]] bool committed() const noexcept { return _committed; }
};
using TriangleCandidate = SurfaceCandidate;
class ProceduralCandidate {};

class AccelVar;

template<bool any_hit>
class RayQuery {
    const AccelVar *_accel;
    StructVar<Ray> _ray;
    std::function<void(SurfaceCandidate &)> _on_surface;

public:
    RayQuery(const AccelVar *accel, const StructVar<Ray> &ray) noexcept : _accel{accel}, _ray{ray} {}
    template<typename F>
    [[nodiscard]] RayQuery &on_surface_candidate(F &&f) noexcept { _on_surface = std::forward<F>(f); return *this; }
    template<typename F>
    [[nodiscard]] RayQuery &on_triangle_candidate(F &&f) noexcept { _on_surface = std::forward<F>(f); return *this; }
    template<typename F>
    [[nodiscard]] RayQuery &on_procedural_candidate(F &&) noexcept { return *this; }
    [[nodiscard]] StructVar<CommittedHit> trace() const noexcept;
};

// World -> object matrix of an acceleration-structure instance.  The ray-tracing unit is LuisaCompute's (absent from the
// snapshot); this stand-in inverts the affine instance transform in double precision and rounds once, exactly as the oracle's
// ray-tracing unit does (oracle/oracle_bvh.h), so that the two intersect instanced geometry identically and the per-sample
// comparison of instanced scenes is bit for bit (round 3; a glm-style fp32 inverse here left them 1 ulp apart).
[[nodiscard]] inline float4x4 accel_world_to_object(const float4x4 &m) noexcept {
    double a[3][3], inv[3][3];
    for (auto r = 0; r < 3; r++) {
        for (auto c = 0; c < 3; c++) { a[r][c] = m[c][r]; }
    }
    auto det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
               a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    auto id = 1.0 / det;
    inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) * id, inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
    inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id, inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) * id;
    inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id, inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
    inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) * id, inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
    inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
    float4x4 out{};
    for (auto r = 0; r < 3; r++) {
        for (auto c = 0; c < 3; c++) { out[c][r] = static_cast<float>(inv[r][c]); }
        out[3][r] = static_cast<float>(-(inv[r][0] * m[3].x + inv[r][1] * m[3].y + inv[r][2] * m[3].z));
    }
    out[0].w = 0.f, out[1].w = 0.f, out[2].w = 0.f, out[3].w = 1.f;
    return out;
}

class AccelVar {
public:
    struct Instance {
        const Mesh *mesh;
        float4x4 to_world;
        float4x4 to_object;
        bool visible;
        bool opaque;
    };

private:
    const std::vector<Instance> *_instances;

public:
    explicit AccelVar(const std::vector<Instance> *instances) noexcept : _instances{instances} {}
    template<typename Visit>
    void traverse_all(const StructVar<Ray> &ray, float &t_max, Visit &&visit) const noexcept {
        auto o = ray.origin(), d = ray.direction();
        for (auto i = 0u; i < _instances->size(); i++) {
            auto &inst = (*_instances)[i];
            if (!inst.visible) { continue; }
            auto oo = make_float3(inst.to_object * make_float4(o, 1.f));
            auto dd = make_float3x3(inst.to_object) * d;
            auto stop = false;
            inst.mesh->traverse(oo, dd, ray.t_min(), t_max, [&](uint prim, float t, float u, float v) noexcept {
                stop = visit(i, prim, t, u, v, inst.opaque);
                return stop;
            });
            if (stop) { return; }
        }
    }
    [[nodiscard]] StructVar<SurfaceHit> intersect(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept {
        StructVar<SurfaceHit> hit;
        auto t_max = ray.t_max();
        traverse_all(ray, t_max, [&](uint inst, uint prim, float t, float u, float v, bool) noexcept {
            hit.inst = inst, hit.prim = prim, hit.bary = float2{u, v}, hit.committed_ray_t = t;
            t_max = t;
            return false;
        });
        return hit;
    }
    [[nodiscard]] StructVar<SurfaceHit> trace_closest(const StructVar<Ray> &ray, const AccelTraceOptions &o = {}) const noexcept { return intersect(ray, o); }
    [[nodiscard]] bool intersect_any(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept {
        auto any = false;
        auto t_max = ray.t_max();
        traverse_all(ray, t_max, [&](uint, uint, float, float, float, bool) noexcept { return any = true; });
        return any;
    }
    [[nodiscard]] bool trace_any(const StructVar<Ray> &ray, const AccelTraceOptions &o = {}) const noexcept { return intersect_any(ray, o); }
    [[nodiscard]] RayQuery<false> traverse(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept { return {this, ray}; }
    [[nodiscard]] RayQuery<true> traverse_any(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept { return {this, ray}; }
    [[nodiscard]] RayQuery<false> query_all(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept { return {this, ray}; }
    [[nodiscard]] RayQuery<true> query_any(const StructVar<Ray> &ray, const AccelTraceOptions & = {}) const noexcept { return {this, ray}; }
    [[nodiscard]] float4x4 instance_transform(uint i) const noexcept { return (*_instances)[i].to_world; }
    [[nodiscard]] const AccelVar *operator->() const noexcept { return this; }
};

template<bool any_hit>
StructVar<CommittedHit> RayQuery<any_hit>::trace() const noexcept {
    StructVar<CommittedHit> best;
    auto t_max = _ray.t_max();
    _accel->traverse_all(_ray, t_max, [&](uint inst, uint prim, float t, float u, float v, bool opaque) noexcept {
        auto commit = opaque;
        if (!opaque) {
            StructVar<SurfaceHit> h;
            h.inst = inst, h.prim = prim, h.bary = float2{u, v}, h.committed_ray_t = t;
            SurfaceCandidate c{_ray, h};
            if (_on_surface) { _on_surface(c); }
            commit = c.committed();
        }
        if (commit) {
            best.inst = inst, best.prim = prim, best.bary = float2{u, v}, best.hit_type = 1u, best.committed_ray_t = t;
            t_max = t;
            return any_hit;
        }
        return false;
    });
    return best;
}

class Accel : public Resource {
    std::vector<AccelVar::Instance> _instances;

public:
    Accel() noexcept = default;
    Accel(Accel &&) noexcept = default;
    Accel &operator=(Accel &&) noexcept = default;
    [[nodiscard]] size_t size() const noexcept { return _instances.size(); }
    void emplace_back(const Mesh &mesh, float4x4 transform = float4x4{}, bool visible = true, bool opaque = true) noexcept {
        _instances.push_back({&mesh, transform, accel_world_to_object(transform), visible, opaque});
    }
    void set_transform_on_update(size_t i, float4x4 m) noexcept { _instances[i].to_world = m, _instances[i].to_object = accel_world_to_object(m); }
    void set_visibility_on_update(size_t i, bool v) noexcept { _instances[i].visible = v; }
    ShimCommand build() noexcept { return {}; }
    ShimCommand update() noexcept { return {}; }
    [[nodiscard]] AccelVar operator->() const noexcept { return AccelVar{&_instances}; }
};

// =============================================================================================================================
// kernels and shaders
// =============================================================================================================================
namespace detail {
template<typename T>
struct kernel_arg { using type = std::remove_cvref_t<T>; };
template<typename T>
struct kernel_arg<BufferVar<T>> { using type = Buffer<T>; };
template<typename T>
struct kernel_arg<ImageVar<T>> { using type = Image<T>; };
template<>
struct kernel_arg<BindlessVar> { using type = BindlessArray; };
template<>
struct kernel_arg<AccelVar> { using type = Accel; };
template<typename T>
using kernel_arg_t = typename kernel_arg<std::remove_cvref_t<T>>::type;

template<typename F>
struct lambda_traits : lambda_traits<decltype(&F::operator())> {};
template<typename C, typename R, typename... A>
struct lambda_traits<R (C::*)(A...) const> { using args = std::tuple<A...>; };
template<typename C, typename R, typename... A>
struct lambda_traits<R (C::*)(A...) const noexcept> { using args = std::tuple<A...>; };
template<typename C, typename R, typename... A>
struct lambda_traits<R (C::*)(A...)> { using args = std::tuple<A...>; };
template<typename C, typename R, typename... A>
struct lambda_traits<R (C::*)(A...) noexcept> { using args = std::tuple<A...>; };

template<typename P, typename A>
[[nodiscard]] inline decltype(auto) to_device_arg(A &&a) noexcept {// host resource -> what the kernel lambda takes
    using PD = std::remove_cvref_t<P>;
    if constexpr (std::is_convertible_v<A &&, PD>) { return static_cast<PD>(std::forward<A>(a)); }
    else { return PD{a.operator->()}; }
}
}// namespace detail

template<size_t N, typename... Args>
class Shader;

template<size_t N, typename... Args>
class ShaderInvoke {
    const Shader<N, Args...> *_shader;
    std::function<void()> _run;

public:
    ShaderInvoke(const Shader<N, Args...> *s, std::function<void()> run) noexcept : _shader{s}, _run{std::move(run)} {}
    ShimCommand dispatch(uint3 size) const noexcept {
        auto &state = detail::dispatch_state();
        auto saved = state;
        state.size = size;
        for (auto z = 0u; z < size.z; z++) {
            for (auto y = 0u; y < size.y; y++) {
                for (auto x = 0u; x < size.x; x++) {
                    state.id = uint3{x, y, z};
                    _run();
                }
            }
        }
        state = saved;
        return {};
    }
    ShimCommand dispatch(uint x) const noexcept { return dispatch(uint3{x, 1u, 1u}); }
    ShimCommand dispatch(uint x, uint y) const noexcept { return dispatch(uint3{x, y, 1u}); }
    ShimCommand dispatch(uint x, uint y, uint z) const noexcept { return dispatch(uint3{x, y, z}); }
    ShimCommand dispatch(uint2 s) const noexcept { return dispatch(uint3{s.x, s.y, 1u}); }
};

template<size_t N, typename... Args>
class Shader : public Resource {
    std::function<void(const Args &...)> _f;

public:
    Shader() noexcept = default;
    explicit Shader(std::function<void(const Args &...)> f) noexcept : _f{std::move(f)} {}
    Shader(Shader &&) noexcept = default;
    Shader(const Shader &) noexcept = default;
    Shader &operator=(Shader &&) noexcept = default;
    Shader &operator=(const Shader &) noexcept = default;
    [[nodiscard]] explicit operator bool() const noexcept { return static_cast<bool>(_f); }
    template<typename... A>
    [[nodiscard]] auto operator()(A &&...a) const noexcept {
        auto args = std::make_shared<std::tuple<const Args &...>>(std::forward<A>(a)...);
        // scalars passed by value must outlive the call: copy what is not a resource
        auto values = std::make_shared<std::tuple<std::conditional_t<std::is_base_of_v<Resource, Args>, const Args *, Args>...>>(
            [](auto &&x) noexcept -> decltype(auto) {
                using X = std::remove_cvref_t<decltype(x)>;
                if constexpr (std::is_base_of_v<Resource, X>) { return &x; }
                else { return std::forward<decltype(x)>(x); }
            }(std::forward<A>(a))...);
        auto f = _f;
        return ShaderInvoke<N, Args...>{this, [f, values] {
                                            std::apply([&f](auto &&...v) {
                                                f([](auto &&x) noexcept -> decltype(auto) {
                                                    using X = std::remove_cvref_t<decltype(x)>;
                                                    if constexpr (std::is_pointer_v<X>) { return *x; }
                                                    else { return (x); }
                                                }(v)...);
                                            }, *values);
                                        }};
    }
};
template<typename... Args>
using Shader1D = Shader<1, Args...>;
template<typename... Args>
using Shader2D = Shader<2, Args...>;
template<typename... Args>
using Shader3D = Shader<3, Args...>;

template<size_t N, typename F>
class Kernel {
    F _f;

public:
    Kernel(F f) noexcept : _f{std::move(f)} {}
    [[nodiscard]] const F &function() const noexcept { return _f; }
};
template<typename F>
struct Kernel1D : Kernel<1, F> { Kernel1D(F f) noexcept : Kernel<1, F>{std::move(f)} {} };
template<typename F>
struct Kernel2D : Kernel<2, F> { Kernel2D(F f) noexcept : Kernel<2, F>{std::move(f)} {} };
template<typename F>
struct Kernel3D : Kernel<3, F> { Kernel3D(F f) noexcept : Kernel<3, F>{std::move(f)} {} };
template<typename F>
Kernel1D(F) -> Kernel1D<F>;
template<typename F>
Kernel2D(F) -> Kernel2D<F>;
template<typename F>
Kernel3D(F) -> Kernel3D<F>;

class Printer {// the DSL's device-side printf; here it prints at once when LIBREF_PRINTER is set (a free trace of the reference)
    static bool _enabled() noexcept {
        static bool e = std::getenv("LIBREF_PRINTER") != nullptr;
        return e;
    }
    template<typename... A>
    static void _print(std::string_view f, const A &...a) noexcept {
        if (_enabled()) { std::fprintf(stderr, "[ref] %s\n", fmt::format(f, a...).c_str()); }
    }

public:
    template<typename D>
    explicit Printer(D &) noexcept {}
    ShimCommand reset() noexcept { return {}; }
    ShimCommand retrieve() noexcept { return {}; }
    [[nodiscard]] bool empty() const noexcept { return true; }
    template<typename... A>
    void info(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void verbose(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void error(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void warning(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void info_with_location(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void verbose_with_location(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void warning_with_location(std::string_view f, const A &...a) noexcept { _print(f, a...); }
    template<typename... A>
    void error_with_location(std::string_view f, const A &...a) noexcept { _print(f, a...); }
};

// =============================================================================================================================
// device, context
// =============================================================================================================================
class Device {
    template<size_t N, typename F, typename... P>
    [[nodiscard]] static auto _compile(const F &f, std::tuple<P...> *) noexcept {
        using S = Shader<N, detail::kernel_arg_t<P>...>;
        return S{[f](const detail::kernel_arg_t<P> &...args) { f(detail::to_device_arg<P>(args)...); }};
    }

public:
    [[nodiscard]] luisa::string_view backend_name() const noexcept { return "scalar-shim"; }
    template<typename T>
    [[nodiscard]] Buffer<T> create_buffer(size_t n) noexcept { return Buffer<T>{n}; }
    template<typename T>
    [[nodiscard]] Image<T> create_image(PixelStorage s, uint w, uint h, uint mips = 1u) noexcept { return Image<T>{s, uint2{w, h}, mips}; }
    template<typename T>
    [[nodiscard]] Image<T> create_image(PixelStorage s, uint2 size, uint mips = 1u) noexcept { return Image<T>{s, size, mips}; }
    [[nodiscard]] BindlessArray create_bindless_array(size_t n = 65536u) noexcept { return BindlessArray{n}; }
    [[nodiscard]] Accel create_accel(const AccelOption & = {}) noexcept { return Accel{}; }
    template<typename V>
    [[nodiscard]] Mesh create_mesh(const Buffer<V> &v, const Buffer<Triangle> &t, const AccelOption &o = {}) noexcept { return Mesh{v, t, o}; }
    [[nodiscard]] Stream create_stream() noexcept { return Stream{}; }
    template<typename T, typename... A>
    [[nodiscard]] T create(A &&...a) noexcept { return T{std::forward<A>(a)...}; }
    template<size_t N, typename F>
    [[nodiscard]] auto compile(const Kernel<N, F> &k) noexcept {
        return _compile<N>(k.function(), static_cast<typename detail::lambda_traits<F>::args *>(nullptr));
    }
    template<size_t N, typename F>
        requires(!std::is_base_of_v<Kernel<N, std::remove_cvref_t<decltype(std::declval<F>())>>, F>)
    [[nodiscard]] auto compile(F &&f) noexcept {
        using FD = std::remove_cvref_t<F>;
        if constexpr (requires { f.function(); }) {
            return _compile<N>(f.function(), static_cast<typename detail::lambda_traits<std::remove_cvref_t<decltype(f.function())>>::args *>(nullptr));
        } else {
            return _compile<N>(f, static_cast<typename detail::lambda_traits<FD>::args *>(nullptr));
        }
    }
};

class Context {
    std::filesystem::path _runtime_dir;

public:
    Context() noexcept = default;
    explicit Context(std::filesystem::path p) noexcept : _runtime_dir{std::move(p)} {}
    [[nodiscard]] const std::filesystem::path &runtime_directory() const noexcept { return _runtime_dir; }
    [[nodiscard]] Device create_device(luisa::string_view = {}, const void * = nullptr) noexcept { return Device{}; }
};

// =============================================================================================================================
// Polymorphic<T>: tag -> implementation, dispatched by a switch in the DSL, by a call here
// =============================================================================================================================
namespace detail {
struct SwitchStack {
    std::vector<std::pair<uint, bool>> values;
};
[[nodiscard]] inline SwitchStack &switch_stack() noexcept {
    static thread_local SwitchStack s;
    return s;
}
struct SwitchStmtBuilder {
    uint value;
    explicit SwitchStmtBuilder(uint v) noexcept : value{v} {}
    template<typename F>
    void operator%(F &&body) const noexcept {
        switch_stack().values.emplace_back(value, false);
        std::forward<F>(body)();
        switch_stack().values.pop_back();
    }
};
struct SwitchCaseStmtBuilder {
    uint value;
    explicit SwitchCaseStmtBuilder(uint v) noexcept : value{v} {}
    template<typename F>
    void operator%(F &&body) const noexcept {
        auto &top = switch_stack().values.back();
        if (!top.second && top.first == value) {
            top.second = true;
            std::forward<F>(body)();
        }
    }
};
struct SwitchDefaultStmtBuilder {
    template<typename F>
    void operator%(F &&body) const noexcept {
        auto &top = switch_stack().values.back();
        if (!top.second) { top.second = true; std::forward<F>(body)(); }
    }
};
}// namespace detail

template<typename T>
class Polymorphic {
    luisa::vector<luisa::unique_ptr<T>> _impl;

public:
    [[nodiscard]] auto empty() const noexcept { return _impl.empty(); }
    [[nodiscard]] auto size() const noexcept { return _impl.size(); }
    [[nodiscard]] auto impl(size_t i) noexcept { return _impl[i].get(); }
    [[nodiscard]] auto impl(size_t i) const noexcept { return const_cast<const T *>(_impl[i].get()); }
    uint emplace(luisa::unique_ptr<T> p) noexcept {
        _impl.emplace_back(std::move(p));
        return static_cast<uint>(_impl.size() - 1u);
    }
    template<typename Impl, typename... A>
    uint create(A &&...a) noexcept { return emplace(luisa::make_unique<Impl>(std::forward<A>(a)...)); }
    template<typename F>
    void dispatch(uint tag, F &&f) const noexcept {// a DSL switch over the tags: no case, no effect
        if (tag < _impl.size()) { std::forward<F>(f)(impl(tag)); }
    }
    template<typename F>
    void dispatch_range(uint tag, uint lo, uint hi, F &&f) const noexcept {
        if (tag >= lo && tag < hi) { std::forward<F>(f)(impl(tag)); }
    }
};

}// namespace luisa::compute

namespace luisa {
constexpr uint64_t hash64_default_seed = 19980810ull;
}

// ---- name injection -------------------------------------------------------------------------------------------------------
// The render code calls builtins unqualified (`abs(u.x)`, `saturate(x)`) and relies on argument-dependent lookup through the
// DSL's class types.  Scalars are fundamental types here, so the names are injected into luisa::render once, up front; the
// render code's own overloads (e.g. abs(const SampledSpectrum &), util/spec.h) then join the same overload sets.
#define LC_SHIM_INJECT_BUILTINS                                                                                                  \
    using luisa::abs; using luisa::sqrt; using luisa::rsqrt; using luisa::exp; using luisa::exp2; using luisa::log;              \
    using luisa::log2; using luisa::log10; using luisa::pow; using luisa::sin; using luisa::cos; using luisa::tan;               \
    using luisa::asin; using luisa::acos; using luisa::atan; using luisa::atan2; using luisa::sinh; using luisa::cosh;           \
    using luisa::tanh; using luisa::floor; using luisa::ceil; using luisa::round; using luisa::trunc; using luisa::fract;        \
    using luisa::fmod; using luisa::saturate; using luisa::clamp; using luisa::min; using luisa::max; using luisa::lerp;         \
    using luisa::fma; using luisa::sign; using luisa::isinf; using luisa::isnan; using luisa::any; using luisa::all;             \
    using luisa::none; using luisa::dot; using luisa::cross; using luisa::length; using luisa::length_squared;                   \
    using luisa::distance; using luisa::distance_squared; using luisa::normalize; using luisa::reflect; using luisa::select;     \
    using luisa::radians; using luisa::degrees; using luisa::transpose; using luisa::inverse; using luisa::determinant;          \
    using luisa::reduce_sum; using luisa::reduce_max; using luisa::reduce_min; using luisa::popcount; using luisa::clz;          \
    using luisa::ctz; using luisa::reverse;                                                                                      \
    using luisa::compute::ite; using luisa::compute::def; using luisa::compute::cast; using luisa::compute::as;                  \
    using luisa::compute::if_; using luisa::compute::make_ray; using luisa::compute::offset_ray_origin;
namespace luisa::render {
using namespace luisa::compute;
LC_SHIM_INJECT_BUILTINS
}// namespace luisa::render
