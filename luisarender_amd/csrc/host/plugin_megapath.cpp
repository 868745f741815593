// plugin_megapath.cpp — `libluisa-render-integrator-megapath.so`: the MegaPath integrator plugin.
//
// Same plugin contract as the reference's src/integrators/mega_path.cpp:13-32,167 (class
// MegakernelPathTracing, props depth/rr_depth/rr_threshold, exported create/destroy), but
// Instance::render drives the hand-written gfx950 megakernel through the C ABI of include/lrhip.h
// instead of JIT-compiling a LuisaCompute kernel.  Host flow = ProgressiveIntegrator::Instance::
// render / _render_one_camera (src/base/integrator.cpp:34-113): per camera prepare film -> render
// spp -> download (convert) -> save_image, logging "Rendering finished in {} ms.".
//
// Compiled three times: LR_PLUGIN_IMPL = "megapath" (default), "direct" (DirectLighting, src/integrators/direct.cpp) and
// "normal" (NormalVisualizer, src/integrators/normal.cpp) — the sibling integrators of SURVEY §8 f4 are run-time modes of
// the same megakernel (lr_integrator.kind, set by the scene loader from the node's impl type), so their plugins differ
// only in the name they register and in NormalVisualizer not needing a light.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../../include/lrhip.h"
#include "luisa_render_shim.h"

#ifndef LR_PLUGIN_IMPL
#define LR_PLUGIN_IMPL "megapath"
#endif

namespace luisa::render {

namespace {

struct HipApi {
    void *module{nullptr};
    decltype(&lrhip_create) create{};
    decltype(&lrhip_destroy) destroy{};
    decltype(&lrhip_upload_scene) upload_scene{};
    decltype(&lrhip_update_scene) update_scene{};
    decltype(&lrhip_film_clear) film_clear{};
    decltype(&lrhip_render) render{};
    decltype(&lrhip_synchronize) synchronize{};
    decltype(&lrhip_film_download) film_download{};
    decltype(&lrhip_last_error) last_error{};

    bool load(const std::filesystem::path &runtime_dir) {
        for (auto &dir : {runtime_dir, runtime_dir / ".." / "lib"}) {
            auto path = dir / "liblrhip.so";
            if ((module = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) != nullptr) { break; }
        }
        if (module == nullptr) { return false; }
#define LR_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(module, name))
        LR_SYM(create, "lrhip_create");
        LR_SYM(destroy, "lrhip_destroy");
        LR_SYM(upload_scene, "lrhip_upload_scene");
        LR_SYM(update_scene, "lrhip_update_scene");
        LR_SYM(film_clear, "lrhip_film_clear");
        LR_SYM(render, "lrhip_render");
        LR_SYM(synchronize, "lrhip_synchronize");
        LR_SYM(film_download, "lrhip_film_download");
        LR_SYM(last_error, "lrhip_last_error");
#undef LR_SYM
        return create && destroy && upload_scene && update_scene && film_clear && render && synchronize && film_download && last_error;
    }
};

}// namespace

class MegakernelPathTracing final : public Integrator {
    uint32_t _max_depth, _rr_depth;
    float _rr_threshold;

public:
    MegakernelPathTracing(Scene *scene, const SceneNodeDesc *desc) noexcept
        : Integrator{scene, desc},
          _max_depth{std::max(desc->uint_or("depth", 10u), 1u)},
          _rr_depth{desc->uint_or("rr_depth", 0u)},
          _rr_threshold{std::max(desc->float_or("rr_threshold", 0.95f), 0.05f)} {}
    [[nodiscard]] auto max_depth() const noexcept { return _max_depth; }
    [[nodiscard]] auto rr_depth() const noexcept { return _rr_depth; }
    [[nodiscard]] auto rr_threshold() const noexcept { return _rr_threshold; }
    [[nodiscard]] std::string_view impl_type() const noexcept override { return LR_PLUGIN_IMPL; }
    [[nodiscard]] std::unique_ptr<Integrator::Instance> build(Pipeline &pipeline, CommandBuffer &command_buffer) const noexcept override;
};

class MegakernelPathTracingInstance final : public Integrator::Instance {
public:
    using Integrator::Instance::Instance;

    void render(Stream &stream) noexcept override {
        auto &scene = pipeline().scene();
        auto &data = scene.data();
        HipApi api;
        if (!api.load(scene.runtime_directory())) {
            std::fprintf(stderr, "[error] Failed to load liblrhip.so (the gfx950 megakernel library): %s\n", dlerror());
            std::abort();// LUISA_ERROR semantics: log + abort
        }
        lrhip_ctx *ctx = nullptr;
        auto device_index = stream.device != nullptr ? std::max(stream.device->index, 0) : 0;
        if (api.create(device_index, &ctx) != LRHIP_OK) {
            std::fprintf(stderr, "[error] lrhip_create: %s\n", api.last_error());
            std::abort();
        }
        for (size_t i = 0; i < data.cameras.size(); i++) {
            auto &camera = data.cameras[i];
            auto width = camera.camera.width, height = camera.camera.height;
            std::vector<float> pixels(static_cast<size_t>(width) * height * 4u, 0.f);
            if (!pipeline().has_lighting() && std::string_view{LR_PLUGIN_IMPL} != "normal") {// mega_path.cpp:40-46, direct.cpp:57-63: warn, still write a black image
                lr::log_warning("No lights in scene. Rendering aborted.");
                for (size_t p = 0; p < static_cast<size_t>(width) * height; p++) { pixels[p * 4u + 3u] = 1.f; }
            } else {
                std::fprintf(stderr, "[info] Rendering to '%s' of resolution %ux%u at %uspp.\n", camera.file.c_str(), width, height, camera.camera.spp);
                auto tiles = ((width + 7u) / 8u) * ((height + 7u) / 8u);
                auto t0 = std::chrono::steady_clock::now();
                // the loop over shutter samples of _render_one_camera (src/base/integrator.cpp:86-107): pipeline().update(time),
                // then `spp` launches of render(sample_id++, time, weight) — here one persistent launch per shutter sample
                auto sample_id = 0u;
                auto first = true;
                auto &shutter = camera.shutter_samples;
                for (auto &s : shutter) {
                    auto moved = lr::set_scene_time(data, s.time);
                    if (first || moved) {
                        auto view = data.view(i);
                        if ((first ? api.upload_scene(ctx, &view) : api.update_scene(ctx, &view)) != LRHIP_OK) {
                            std::fprintf(stderr, "[error] lrhip_upload_scene: %s\n", api.last_error());
                            std::abort();
                        }
                    }
                    if (first) { api.film_clear(ctx); }
                    first = false;
                    lrhip_render_params params{sample_id, sample_id + s.spp, 0u, tiles, 1u, shutter.size() > 1u ? LRHIP_RENDER_SHUTTER_WEIGHT : 0u, 1u, s.weight};
                    sample_id += s.spp;
                    if (api.render(ctx, &params) != LRHIP_OK || api.synchronize(ctx) != LRHIP_OK) {
                        std::fprintf(stderr, "[error] lrhip_render: %s\n", api.last_error());
                        std::abort();
                    }
                }
                auto ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                std::fprintf(stderr, "[info] Rendering finished in %g ms.\n", ms);
                auto samples = static_cast<double>(width) * height * camera.camera.spp;
                std::fprintf(stderr, "[info] %.2f Msamples/s on HIP device %d.\n", samples / ms * 1e-3, device_index);
                if (api.film_download(ctx, pixels.data(), 1) != LRHIP_OK) {
                    std::fprintf(stderr, "[error] lrhip_film_download: %s\n", api.last_error());
                    std::abort();
                }
            }
            lr::save_image(camera.file, pixels.data(), width, height);
        }
        api.destroy(ctx);
    }
};

std::unique_ptr<Integrator::Instance> MegakernelPathTracing::build(Pipeline &pipeline, CommandBuffer &command_buffer) const noexcept {
    return std::make_unique<MegakernelPathTracingInstance>(pipeline, command_buffer, this);
}

}// namespace luisa::render

LUISA_RENDER_MAKE_SCENE_NODE_PLUGIN(luisa::render::MegakernelPathTracing)
