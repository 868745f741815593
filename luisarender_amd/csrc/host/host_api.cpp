// host_api.cpp — C ABI of liblrhost.so (include/lrhost.h).
#include "../../../include/lrhost.h"

#include <cstdlib>
#include <cstring>

#include "scene.h"

struct lrhost_scene {
    std::unique_ptr<lr::SceneData> data;
};

namespace {

thread_local std::string g_last_error;

lr::MacroMap make_macros(const char *const *keys, const char *const *values, int count) {
    lr::MacroMap macros;
    for (auto i = 0; i < count; i++) { macros[keys[i]] = values[i]; }
    return macros;
}

template<typename F>
int guarded(F &&f) {
    try {
        f();
        return LRHOST_OK;
    } catch (const std::exception &e) {
        g_last_error = e.what();
    } catch (...) {
        g_last_error = "unknown error";
    }
    return LRHOST_ERROR;
}

}// namespace

extern "C" {

int lrhost_scene_load_file(const char *path, const char *const *macro_keys, const char *const *macro_values,
                           int macro_count, lrhost_scene **out) {
    return guarded([&] {
        auto desc = lr::parse_scene_file(path, make_macros(macro_keys, macro_values, macro_count));
        auto s = new lrhost_scene{};
        try {
            s->data = lr::build_scene(*desc);
        } catch (...) {
            delete s;
            throw;
        }
        *out = s;
    });
}

int lrhost_scene_load_string(const char *source, const char *virtual_path, int is_json,
                             const char *const *macro_keys, const char *const *macro_values,
                             int macro_count, lrhost_scene **out) {
    return guarded([&] {
        auto desc = lr::parse_scene_string(source, virtual_path ? virtual_path : "", make_macros(macro_keys, macro_values, macro_count), is_json != 0);
        auto s = new lrhost_scene{};
        try {
            s->data = lr::build_scene(*desc);
        } catch (...) {
            delete s;
            throw;
        }
        *out = s;
    });
}

int lrhost_scene_build_accel(lrhost_scene *scene) {
    return guarded([&] { lr::build_accel(*scene->data); });
}

int lrhost_scene_set_time(lrhost_scene *scene, float time, int *updated) {
    return guarded([&] {
        auto moved = lr::set_scene_time(*scene->data, time);
        if (updated != nullptr) { *updated = moved ? 1 : 0; }
    });
}

int lrhost_scene_shutter_sample_count(const lrhost_scene *scene, int camera_index) {
    if (camera_index < 0 || static_cast<size_t>(camera_index) >= scene->data->cameras.size()) { return 0; }
    return static_cast<int>(scene->data->cameras[static_cast<size_t>(camera_index)].shutter_samples.size());
}

int lrhost_scene_shutter_sample(const lrhost_scene *scene, int camera_index, int sample_index, float *time, float *weight, uint32_t *spp) {
    return guarded([&] {
        if (camera_index < 0 || static_cast<size_t>(camera_index) >= scene->data->cameras.size()) { throw lr::Error{"Camera index out of range."}; }
        auto &samples = scene->data->cameras[static_cast<size_t>(camera_index)].shutter_samples;
        if (sample_index < 0 || static_cast<size_t>(sample_index) >= samples.size()) { throw lr::Error{"Shutter sample index out of range."}; }
        auto &s = samples[static_cast<size_t>(sample_index)];
        if (time) { *time = s.time; }
        if (weight) { *weight = s.weight; }
        if (spp) { *spp = s.spp; }
    });
}

int lrhost_scene_camera_count(const lrhost_scene *scene) { return static_cast<int>(scene->data->cameras.size()); }

int lrhost_scene_view(const lrhost_scene *scene, int camera_index, lr_scene *out) {
    return guarded([&] { *out = scene->data->view(static_cast<size_t>(camera_index)); });
}

const char *lrhost_scene_camera_file(const lrhost_scene *scene, int camera_index) {
    if (camera_index < 0 || static_cast<size_t>(camera_index) >= scene->data->cameras.size()) { return nullptr; }
    return scene->data->cameras[static_cast<size_t>(camera_index)].file.c_str();
}

int lrhost_scene_has_lighting(const lrhost_scene *scene) { return scene->data->has_lighting() ? 1 : 0; }

void lrhost_scene_destroy(lrhost_scene *scene) { delete scene; }

int lrhost_save_image(const char *path, const float *rgba, uint32_t width, uint32_t height) {
    return guarded([&] { lr::save_image(path, rgba, width, height); });
}

int lrhost_load_image(const char *path, float **rgba, uint32_t *width, uint32_t *height, uint32_t *channels) {
    return guarded([&] {
        auto img = lr::load_image(path);
        auto bytes = img.pixels.size() * sizeof(float);
        auto p = static_cast<float *>(std::malloc(bytes));
        std::memcpy(p, img.pixels.data(), bytes);
        *rgba = p, *width = img.width, *height = img.height, *channels = img.channels;
    });
}

void lrhost_free(void *p) { std::free(p); }

uint64_t lrhost_sizeof(const char *name) {
#define LR_SIZEOF(T) if (std::strcmp(name, #T) == 0) { return sizeof(T); }
    LR_SIZEOF(lr_scene) LR_SIZEOF(lr_vertex) LR_SIZEOF(lr_triangle) LR_SIZEOF(lr_alias_entry) LR_SIZEOF(lr_mesh)
    LR_SIZEOF(lr_instance) LR_SIZEOF(lr_texture) LR_SIZEOF(lr_surface) LR_SIZEOF(lr_light) LR_SIZEOF(lr_environment)
    LR_SIZEOF(lr_camera) LR_SIZEOF(lr_filter) LR_SIZEOF(lr_film) LR_SIZEOF(lr_sampler) LR_SIZEOF(lr_integrator)
    LR_SIZEOF(lr_bvh4_node) LR_SIZEOF(lr_bvh_triangle) LR_SIZEOF(lr_accel) LR_SIZEOF(lr_light_handle)
    LR_SIZEOF(lr_medium)
#undef LR_SIZEOF
    return 0u;
}

void lrhost_set_log_level(int level) { lr::set_log_level(level); }

const char *lrhost_last_error(void) { return g_last_error.c_str(); }

}// extern "C"
