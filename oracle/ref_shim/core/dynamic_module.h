// stand-in for luisa/core/dynamic_module.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): dlopen / dlsym, as the original.
#pragma once
#include "../lc_core.h"
#include <dlfcn.h>
namespace luisa {
class DynamicModule {
    void *_handle{nullptr};
    explicit DynamicModule(void *h) noexcept : _handle{h} {}

public:
    DynamicModule() noexcept = default;
    DynamicModule(DynamicModule &&o) noexcept : _handle{o._handle} { o._handle = nullptr; }
    DynamicModule &operator=(DynamicModule &&o) noexcept { std::swap(_handle, o._handle); return *this; }
    DynamicModule(const DynamicModule &) = delete;
    ~DynamicModule() noexcept = default;// plugins stay loaded for the life of the process, as in the reference's registry
    [[nodiscard]] static DynamicModule load(const std::filesystem::path &dir, std::string_view name) noexcept {
        auto path = dir / ("lib" + std::string{name} + ".so");
        auto h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) { LUISA_ERROR("Failed to load plugin '{}': {}", path.string(), dlerror()); }
        return DynamicModule{h};
    }
    [[nodiscard]] explicit operator bool() const noexcept { return _handle != nullptr; }
    template<typename F>
    [[nodiscard]] F *function(std::string_view name) const noexcept {
        auto p = dlsym(_handle, std::string{name}.c_str());
        if (p == nullptr) { LUISA_ERROR("Symbol '{}' not found in plugin.", name); }
        return reinterpret_cast<F *>(p);
    }
    template<typename F, typename... A>
    decltype(auto) invoke(std::string_view name, A &&...a) const noexcept { return function<F>(name)(std::forward<A>(a)...); }
};
}// namespace luisa
