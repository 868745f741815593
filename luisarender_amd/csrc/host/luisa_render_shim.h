// luisa_render_shim.h — the slice of the reference's C++ class boundary that the MegaPath plugin
// and luisa-render-cli meet at, on a minimal `luisa::render` shim (the LuisaCompute headers the
// original declarations are written against — core/stl.h, dsl/syntax.h, runtime/*.h — are absent).
// Same class / method names and call order as:
//   SceneNode + plugin ABI           src/base/scene_node.h:33-67
//   Scene::load_node / plugin lookup src/base/scene.cpp:54-131  ("luisa-render-<tag>-<impl>")
//   Integrator / Instance            src/base/integrator.h:19-79
//   Pipeline::create / render        src/base/pipeline.cpp:44-99,115-117
#pragma once
#include <vector>
#include <filesystem>
#include <memory>
#include <string>
#include <string_view>

#include "scene.h"

namespace luisa::compute {

// `-b <backend> -d <index>` (src/apps/cli.cpp:166-172): the only backend is "hip" on gfx950
struct Device {
    std::string backend;
    int index{0};
    // multi-GPU extension of `-d` (the reference drives one device, src/apps/cli.cpp:172): `-d 0,1,2,3` / LR_DEVICES=0,1,2,3 shards
    // the frame by screen tile over these HIP devices; empty = the single device `index`
    std::vector<int> indices;
};
struct Stream {
    Device *device{nullptr};
    void synchronize() noexcept {}
};
struct CommandBuffer {
    Stream *stream{nullptr};
};

}// namespace luisa::compute

namespace luisa::render {

using SceneNodeDesc = lr::NodeDesc;
using SceneNodeTag = lr::Tag;
using compute::CommandBuffer;
using compute::Device;
using compute::Stream;

class Scene;
class Pipeline;

class SceneNode {
    const Scene *_scene;
    SceneNodeTag _tag;

public:
    SceneNode(const Scene *scene, const SceneNodeDesc *, SceneNodeTag tag) noexcept : _scene{scene}, _tag{tag} {}
    SceneNode(SceneNode &&) = delete;
    SceneNode(const SceneNode &) = delete;
    virtual ~SceneNode() noexcept = default;
    [[nodiscard]] auto scene() const noexcept { return _scene; }
    [[nodiscard]] auto tag() const noexcept { return _tag; }
    [[nodiscard]] virtual std::string_view impl_type() const noexcept = 0;
};

using NodeCreater = SceneNode *(Scene *, const SceneNodeDesc *);
using NodeDeleter = void(SceneNode *);

class Integrator : public SceneNode {
public:
    class Instance {
        Pipeline &_pipeline;
        const Integrator *_integrator;

    public:
        Instance(Pipeline &pipeline, CommandBuffer &, const Integrator *integrator) noexcept
            : _pipeline{pipeline}, _integrator{integrator} {}
        virtual ~Instance() noexcept = default;
        template<typename T = Integrator>
        [[nodiscard]] auto node() const noexcept { return static_cast<const T *>(_integrator); }
        [[nodiscard]] auto &pipeline() noexcept { return _pipeline; }
        virtual void render(Stream &stream) noexcept = 0;
    };
    Integrator(Scene *scene, const SceneNodeDesc *desc) noexcept : SceneNode{reinterpret_cast<const Scene *>(scene), desc, SceneNodeTag::INTEGRATOR} {}
    [[nodiscard]] virtual std::unique_ptr<Instance> build(Pipeline &pipeline, CommandBuffer &command_buffer) const noexcept = 0;
};

// The scene graph: node descriptions + the flattened tables every plugin reads.
class Scene {
    std::filesystem::path _runtime_directory;
    std::unique_ptr<lr::SceneDesc> _desc;
    std::unique_ptr<lr::SceneData> _data;
    Integrator *_integrator{nullptr};
    NodeDeleter *_integrator_deleter{nullptr};
    void *_integrator_module{nullptr};

public:
    Scene(std::filesystem::path runtime_directory, std::unique_ptr<lr::SceneDesc> desc, std::unique_ptr<lr::SceneData> data) noexcept
        : _runtime_directory{std::move(runtime_directory)}, _desc{std::move(desc)}, _data{std::move(data)} {}
    ~Scene() noexcept;
    [[nodiscard]] const auto &runtime_directory() const noexcept { return _runtime_directory; }
    [[nodiscard]] const lr::SceneDesc &desc() const noexcept { return *_desc; }
    [[nodiscard]] lr::SceneData &data() noexcept { return *_data; }
    [[nodiscard]] const lr::SceneData &data() const noexcept { return *_data; }
    [[nodiscard]] const Integrator *integrator() const noexcept { return _integrator; }
    // Scene::load_integrator -> load_node: dlopen("luisa-render-integrator-<impl>") + create()
    Integrator *load_integrator(const SceneNodeDesc *desc);
    static std::unique_ptr<Scene> create(const std::filesystem::path &runtime_directory, std::unique_ptr<lr::SceneDesc> desc);
};

class Pipeline {
    Device &_device;
    Scene &_scene;
    std::unique_ptr<Integrator::Instance> _integrator;

public:
    Pipeline(Device &device, Scene &scene) noexcept : _device{device}, _scene{scene} {}
    [[nodiscard]] auto &device() noexcept { return _device; }
    [[nodiscard]] auto &scene() noexcept { return _scene; }
    [[nodiscard]] bool has_lighting() const noexcept;
    static std::unique_ptr<Pipeline> create(Device &device, Stream &stream, Scene &scene);
    void render(Stream &stream) noexcept { _integrator->render(stream); }
};

}// namespace luisa::render

#define LUISA_RENDER_MAKE_SCENE_NODE_PLUGIN(cls)                                                                            \
    extern "C" __attribute__((visibility("default"))) luisa::render::SceneNode *create(                                    \
        luisa::render::Scene *scene, const luisa::render::SceneNodeDesc *desc) noexcept { return new cls{scene, desc}; }   \
    extern "C" __attribute__((visibility("default"))) void destroy(luisa::render::SceneNode *node) noexcept { delete node; }
