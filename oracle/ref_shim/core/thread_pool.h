// stand-in for luisa/core/thread_pool.h (oracle/ref_shim, TEST INFRASTRUCTURE ONLY): tasks run on the calling thread, at once.
#pragma once
#include "../lc_core.h"
#include <future>
#include <thread>
namespace luisa {
class ThreadPool {
public:
    explicit ThreadPool(size_t = 0u) noexcept {}
    template<typename F>
    [[nodiscard]] auto async(F &&f) noexcept {
        using R = std::invoke_result_t<F>;
        std::promise<R> p;
        if constexpr (std::is_void_v<R>) { std::forward<F>(f)(); p.set_value(); }
        else { p.set_value(std::forward<F>(f)()); }
        return p.get_future().share();
    }
    template<typename F>
    void parallel(uint n, F &&f) noexcept { for (auto i = 0u; i < n; i++) { f(i); } }
    template<typename F>
    void parallel(uint nx, uint ny, F &&f) noexcept { for (auto y = 0u; y < ny; y++) { for (auto x = 0u; x < nx; x++) { f(x, y); } } }
    void synchronize() noexcept {}
    void barrier() noexcept {}
    [[nodiscard]] size_t size() const noexcept { return 1u; }
};
}// namespace luisa
