// scene.h — scene graph loading and flattening into the POD tables of include/lr_scene.h.
//
// Replaces, for the megapath hot path, the reference's L4/L3 host build:
//   Scene::create                    src/base/scene.cpp:201-233
//   Pipeline::create                 src/base/pipeline.cpp:44-99
//   Geometry::build/_process_shape   src/base/geometry.cpp:12-163
// The reference JIT-compiles each scene's materials into code; here every node becomes a
// record in a table that one hand-written kernel interprets.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/lr_scene.h"
#include "sdl.h"
#include "lr_math.h"

namespace lr {

// A compiled Transform node (src/transforms/*.cpp): evaluable at any time after the scene description is gone.
struct XformNode {
    enum Kind : uint32_t { STATIC, STACK, LERP } kind{STATIC};
    float4x4 m{float4x4::identity()};// STATIC (and the value of a fully static STACK)
    std::vector<uint32_t> children;  // STACK: m = t_i(time) * m (stack.cpp:22-36); LERP: key transforms sorted by time
    std::vector<float> times;        // LERP time points (lerp.cpp:29-66)
    bool is_static{true};
};

// Camera::ShutterSample (src/base/camera.h:81-89): `spp` samples rendered with the scene at `time`, radiance x `weight`
struct ShutterSample {
    float time{0.f}, weight{1.f};
    uint32_t spp{0u};
};

struct CameraRecord {
    lr_camera camera{};
    lr_filter filter{};
    lr_film film{};
    std::string file;// output image path (src/base/camera.cpp:138-147)
    int32_t xform{-1};// compiled transform when it is animated
    float shutter_span[2]{0.f, 0.f};
    std::vector<ShutterSample> shutter_samples;// Camera::shutter_samples(), camera.cpp:163-203
};

// an instance below at least one animated transform: object_to_world(time) = M_root(time) * ... * M_leaf(time)
// (TransformTree::Node::matrix, src/base/transform.cpp:17-23)
struct DynamicInstance {
    uint32_t instance{0u};
    std::vector<uint32_t> chain;// XformNode ids, root first
};

struct SceneData {
    std::vector<lr_vertex> vertices;
    std::vector<lr_triangle> triangles;
    std::vector<lr_alias_entry> tri_alias;
    std::vector<float> tri_pdf;
    std::vector<lr_mesh> meshes;
    std::vector<lr_instance> instances;
    std::vector<lr_light_handle> light_instances;
    std::vector<lr_surface> surfaces;
    std::vector<lr_light> lights;
    std::vector<lr_medium> media;
    std::vector<lr_texture> textures;
    std::vector<float> texels;// float4 units
    lr_environment environment{};
    std::vector<lr_alias_entry> env_alias;
    std::vector<float> env_pdf;
    // the records below a Combined environment (combined.cpp: leaves with their own importance tables, nested Combined nodes;
    // children before parents)
    std::vector<lr_environment> env_children;
    std::vector<std::vector<lr_alias_entry>> env_child_alias;// [env_children.size()]
    std::vector<std::vector<float>> env_child_pdf;
    mutable std::vector<lr_environment> env_children_view;// children with table pointers patched, for view()
    std::vector<CameraRecord> cameras;
    lr_sampler sampler{};
    std::vector<uint32_t> sobol_matrices;      // [1024][52] (Sobol samplers only)
    std::vector<uint64_t> vdc_sobol, vdc_sobol_inv;// all rows [25|26][52]; view() points at the row of the camera's scale
    lr_integrator integrator{};
    std::string integrator_impl;
    bool any_non_opaque{false};
    // animation (src/transforms/lerp.cpp, Geometry::update geometry.cpp:194-216, Pipeline::update pipeline.cpp:101-113)
    std::vector<XformNode> xforms;
    std::vector<DynamicInstance> dynamic_instances;
    int32_t environment_xform{-1};
    float time{0.f};// the time the tables currently hold (initially min over cameras of shutter_span.x, pipeline.cpp:50-56)
    // wide BVH (accel.cpp)
    std::vector<lr_bvh4_node> bvh_nodes;
    std::vector<lr_bvh_triangle> bvh_triangles;
    float world_min[3]{}, world_max[3]{};

    [[nodiscard]] bool has_lighting() const {
        return !lights.empty() || environment.kind != LR_ENV_NONE;
    }
    // POD view for camera `index`; valid while *this is alive and unmodified
    [[nodiscard]] lr_scene view(size_t camera_index = 0u) const;
};

// std::pair<alias table, pdf>, restating create_alias_table (src/util/sampling.cpp:38-87)
void create_alias_table(const float *values, size_t n, std::vector<lr_alias_entry> &table, std::vector<float> &pdf);

// Shape::Handle::encode (src/base/shape.cpp:46-70)
lr_uint4 encode_instance_handle(uint32_t buffer_base, uint32_t flags, uint32_t surface_tag, uint32_t light_tag,
                                uint32_t medium_tag, uint32_t tri_count, float shadow_terminator,
                                float intersection_offset);

std::unique_ptr<SceneData> build_scene(const SceneDesc &desc);

// environment.cpp: importance tables of an image-based Spherical environment (spherical.cpp:144-235)
void build_environment_tables(const SceneData &scene, lr_environment &env, std::vector<lr_alias_entry> &alias, std::vector<float> &pdf);

// accel.cpp: flatten instances to world space and build the 4-wide BVH for the HIP kernel
void build_accel(SceneData &scene);
// accel.cpp: re-bake the triangles of moved instances and refit the boxes of the existing BVH (same topology)
void refit_accel(SceneData &scene);

// scene.cpp: evaluate a compiled transform (Transform::matrix(time))
float4x4 evaluate_xform(const SceneData &scene, uint32_t id, float time);
// scene.cpp: Pipeline::update (pipeline.cpp:101-113) + Geometry::update (geometry.cpp:194-216): re-evaluate every animated
// transform at `time` (instances, cameras, environment), refit the BVH if it is built; returns whether anything moved
bool set_scene_time(SceneData &scene, float time);

// mesh_io.cpp: OBJ loader standing in for assimp (src/shapes/mesh.cpp:46-69 flag semantics)
struct LoadedMesh {
    std::vector<lr_vertex> vertices;
    std::vector<lr_triangle> triangles;
    uint32_t properties{0u};
};
// subdivision > 0: the polygons are kept (no aiProcess_Triangulate, mesh.cpp:69) and go through that many levels of Catmull-Clark
LoadedMesh load_obj_mesh(const std::string &path, bool flip_uv, bool drop_normal, bool drop_uv, uint32_t subdivision = 0u);
// catmull_clark.cpp: the Mesh shape's `subdivision` (assimp's Subdivider in the reference; restated, parity unpinned)
struct PolygonMesh {
    std::vector<lr_vertex> vertices;
    std::vector<uint32_t> indices;     // corners of all faces, face after face
    std::vector<uint32_t> face_offsets;// face f = indices[face_offsets[f] .. face_offsets[f + 1])
};
PolygonMesh catmull_clark_level(const PolygonMesh &in);
LoadedMesh catmull_clark_subdivide(const PolygonMesh &base, uint32_t levels, uint32_t properties);
// subdiv.cpp: Loop subdivision (src/util/loop_subdiv.cpp) and the icosphere of the Sphere shape (src/shapes/sphere.cpp)
LoadedMesh loop_subdivide(const std::vector<lr_vertex> &vertices, const std::vector<lr_triangle> &triangles, uint32_t levels);
LoadedMesh make_sphere_mesh(uint32_t subdivision);

// image_io.cpp
void save_image(const std::string &path, const float *rgba, uint32_t width, uint32_t height);// src/util/imageio.cpp:694-726
struct LoadedImage {
    uint32_t width{0}, height{0}, channels{0};
    bool is_hdr{false};
    std::vector<float> pixels;// float4 per pixel, row 0 = top
};
LoadedImage load_image(const std::string &path);
// image_codecs.cpp: the LDR formats the reference reads through stb_image (imageio.cpp:486-538)
LoadedImage read_jpeg(const std::string &path);
LoadedImage read_bmp(const std::string &path);
LoadedImage read_tga(const std::string &path);

}// namespace lr
