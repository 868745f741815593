// luisa_render_shim.cpp — Scene / Pipeline halves of the class boundary (see luisa_render_shim.h);
// lives in liblrhost.so so that both luisa-render-cli and the integrator plugin resolve it.
#include <dlfcn.h>

#include <mutex>
#include <unordered_map>

#include "luisa_render_shim.h"

namespace luisa::render {

Scene::~Scene() noexcept {
    if (_integrator != nullptr && _integrator_deleter != nullptr) { _integrator_deleter(_integrator); }
}

Integrator *Scene::load_integrator(const SceneNodeDesc *desc) {
    if (desc == nullptr || !desc->is_defined()) { throw lr::Error{"Undefined scene description node for the integrator."}; }
    // "luisa-render-<tag>-<impl>", lower-cased (scene.cpp:64-75); modules stay loaded for the process lifetime
    static std::unordered_map<std::string, void *> registry;
    static std::mutex mutex;
    std::scoped_lock lock{mutex};
    auto name = std::string{"luisa-render-integrator-"} + desc->impl_type();
    for (auto &c : name) { c = static_cast<char>(std::tolower(c)); }
    auto &module = registry[name];
    if (module == nullptr) {
        auto path = _runtime_directory / ("lib" + name + ".so");
        module = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (module == nullptr) {
            throw lr::Error{"Failed to load plugin '" + path.string() + "': " + dlerror() +
                            " (this framework ships the MegaPath integrator only)"};
        }
    }
    auto create = reinterpret_cast<NodeCreater *>(dlsym(module, "create"));
    auto destroy = reinterpret_cast<NodeDeleter *>(dlsym(module, "destroy"));
    if (create == nullptr || destroy == nullptr) { throw lr::Error{"Plugin '" + name + "' does not export create/destroy."}; }
    _integrator = dynamic_cast<Integrator *>(create(this, desc));
    _integrator_deleter = destroy;
    if (_integrator == nullptr) { throw lr::Error{"Plugin '" + name + "' did not create an Integrator."}; }
    return _integrator;
}

std::unique_ptr<Scene> Scene::create(const std::filesystem::path &runtime_directory, std::unique_ptr<lr::SceneDesc> desc) {
    auto data = lr::build_scene(*desc);// Scene::create + Pipeline::create host halves
    auto scene = std::make_unique<Scene>(runtime_directory, std::move(desc), std::move(data));
    scene->load_integrator(scene->desc().root()->node("integrator"));
    return scene;
}

bool Pipeline::has_lighting() const noexcept { return _scene.data().has_lighting(); }

std::unique_ptr<Pipeline> Pipeline::create(Device &device, Stream &stream, Scene &scene) {
    auto pipeline = std::make_unique<Pipeline>(device, scene);
    lr::build_accel(scene.data());// Geometry::build (geometry.cpp:12-27): BLAS/TLAS -> our baked BVH4
    CommandBuffer command_buffer{&stream};
    pipeline->_integrator = scene.integrator()->build(*pipeline, command_buffer);
    return pipeline;
}

}// namespace luisa::render
