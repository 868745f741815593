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
#include <string_view>
#include <algorithm>

#include <thread>

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
    // multi-GPU: only looked up when the frame is sharded
    decltype(&lrhip_comm_init_all) comm_init_all{};
    decltype(&lrhip_comm_destroy) comm_destroy{};
    decltype(&lrhip_film_reduce_group) film_reduce_group{};

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
        LR_SYM(comm_init_all, "lrhip_comm_init_all");
        LR_SYM(comm_destroy, "lrhip_comm_destroy");
        LR_SYM(film_reduce_group, "lrhip_film_reduce_group");
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
        // One lrhip_ctx per GPU.  `-d 0,1,2,3` / LR_DEVICES shards every frame by screen tile over the listed devices:
        // this process flattens the scene and builds the BVH ONCE, one host thread per GPU uploads the same host tables and renders
        // the tiles {r, r + W, ...} (lrhip.h: diagonals of the frame) with balance_shards = W, and ONE collective -- lrhip_film_reduce
        // = ncclReduce over xGMI, communicators from ncclCommInitAll -- sums the films on the first device, which converts and saves.
        // Every pixel is owned by one GPU and the others hold exact zeros there, and the work items are ALWAYS sized for a frame cut into
        // kNominalShards shards (lrhip_render_params.balance_shards: the chunking, and with it the order of a pixel's float adds, is a
        // function of that number, not of the device count), so `-d 0`, `-d 0,1` ... `-d 0,..,7` write the same image bit for bit.
        // (A full frame is insensitive to the finer items: C2 at 7 / 14 / 28 chunks 1992 / 1987 / 1987 ms)
        // LR_FORCE_COLLECTIVE=1 runs the communicator + reduce with ONE device as well (tests: the standalone binary must find RCCL).
        constexpr uint32_t kNominalShards = 8u;
        std::vector<int> devices;
        if (stream.device != nullptr && stream.device->indices.size() > 1u) { devices = stream.device->indices; }
        else { devices = {stream.device != nullptr ? std::max(stream.device->index, 0) : 0}; }
        const auto world = static_cast<uint32_t>(devices.size());
        std::vector<lrhip_ctx *> ctxs(world, nullptr);
        std::vector<void *> comms(world, nullptr);
        for (auto r = 0u; r < world; r++) {
            if (api.create(devices[r], &ctxs[r]) != LRHIP_OK) {
                std::fprintf(stderr, "[error] lrhip_create(device %d): %s\n", devices[r], api.last_error());
                std::abort();
            }
        }
        const auto force_collective = std::getenv("LR_FORCE_COLLECTIVE") != nullptr && std::string_view{std::getenv("LR_FORCE_COLLECTIVE")} == "1";
        const auto collective = world > 1u || force_collective;
        const auto balance_shards = std::max(world, kNominalShards);
        if (collective) {
            if (api.comm_init_all == nullptr || api.film_reduce_group == nullptr || api.comm_init_all(static_cast<int>(world), devices.data(), comms.data()) != LRHIP_OK) {
                std::fprintf(stderr, "[error] lrhip_comm_init_all: %s\n", api.last_error());
                std::abort();
            }
        }
        auto ctx = ctxs[0];
        auto device_index = devices[0];
        // runs f(rank) on one host thread per GPU (the calling thread takes rank 0) and joins; a context is driven by one thread at a time
        auto on_every_gpu = [&](auto &&f) {
            std::vector<std::thread> threads;
            std::vector<std::string> errors(world);
            for (auto r = 1u; r < world; r++) { threads.emplace_back([&, r] { errors[r] = f(r); }); }
            errors[0] = f(0u);
            for (auto &t : threads) { t.join(); }
            for (auto r = 0u; r < world; r++) {
                if (!errors[r].empty()) {
                    std::fprintf(stderr, "[error] HIP device %d: %s\n", devices[r], errors[r].c_str());
                    std::abort();
                }
            }
        };
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
                    auto view = data.view(i);// the host tables of this shutter sample: read-only while the GPUs upload them
                    const auto upload = first || moved;
                    const auto was_first = first;
                    const auto begin = sample_id;
                    on_every_gpu([&](uint32_t r) -> std::string {
                        if (upload && (was_first ? api.upload_scene(ctxs[r], &view) : api.update_scene(ctxs[r], &view)) != LRHIP_OK) { return std::string{"lrhip_upload_scene: "} + api.last_error(); }
                        if (was_first && api.film_clear(ctxs[r]) != LRHIP_OK) { return std::string{"lrhip_film_clear: "} + api.last_error(); }
                        lrhip_render_params params{begin, begin + s.spp, r, tiles, world, shutter.size() > 1u ? LRHIP_RENDER_SHUTTER_WEIGHT : 0u, balance_shards, s.weight};
                        if (api.render(ctxs[r], &params) != LRHIP_OK || api.synchronize(ctxs[r]) != LRHIP_OK) { return std::string{"lrhip_render: "} + api.last_error(); }
                        return {};
                    });
                    first = false;
                    sample_id += s.spp;
                }
                if (collective) {// the path's one collective: per-GPU films -> devices[0]
                    if (api.film_reduce_group(static_cast<int>(world), ctxs.data(), comms.data(), 0) != LRHIP_OK) {
                        std::fprintf(stderr, "[error] lrhip_film_reduce: %s\n", api.last_error());
                        std::abort();
                    }
                    on_every_gpu([&](uint32_t r) -> std::string { return api.synchronize(ctxs[r]) != LRHIP_OK ? std::string{"lrhip_synchronize: "} + api.last_error() : std::string{}; });
                }
                auto ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                std::fprintf(stderr, "[info] Rendering finished in %g ms.\n", ms);
                auto samples = static_cast<double>(width) * height * camera.camera.spp;
                std::fprintf(stderr, "[info] %.2f Msamples/s on %u HIP device(s), first %d%s.\n", samples / ms * 1e-3, world, device_index,
                             collective ? ", films reduced over RCCL" : "");
                if (api.film_download(ctx, pixels.data(), 1) != LRHIP_OK) {
                    std::fprintf(stderr, "[error] lrhip_film_download: %s\n", api.last_error());
                    std::abort();
                }
            }
            lr::save_image(camera.file, pixels.data(), width, height);
        }
        for (auto r = 0u; r < world; r++) {
            if (comms[r] != nullptr) { api.comm_destroy(comms[r]); }
            api.destroy(ctxs[r]);
        }
    }
};

std::unique_ptr<Integrator::Instance> MegakernelPathTracing::build(Pipeline &pipeline, CommandBuffer &command_buffer) const noexcept {
    return std::make_unique<MegakernelPathTracingInstance>(pipeline, command_buffer, this);
}

}// namespace luisa::render

LUISA_RENDER_MAKE_SCENE_NODE_PLUGIN(luisa::render::MegakernelPathTracing)
