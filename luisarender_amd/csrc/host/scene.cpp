// scene.cpp — scene graph -> flattened tables.  See scene.h for the reference map.
#include "scene.h"

#include <dlfcn.h>

#include <array>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <limits>
#include <numeric>
#include <random>
#include <algorithm>
#include <unordered_map>

namespace lr {

namespace fs = std::filesystem;

// ------------------------------------------------------------------ helpers

void create_alias_table(const float *values, size_t n, std::vector<lr_alias_entry> &table, std::vector<float> &pdf) {
    // src/util/sampling.cpp:38-87 — double-precision sum, float table, pairing from the back
    auto sum = 0.0;
    for (size_t i = 0; i < n; i++) { sum += std::abs(values[i]); }
    pdf.assign(n, 0.f);
    if (sum == 0.) {
        std::fill(pdf.begin(), pdf.end(), static_cast<float>(1.0 / static_cast<double>(n)));
    } else {
        auto inv_sum = 1.0 / sum;
        for (size_t i = 0; i < n; i++) { pdf[i] = static_cast<float>(std::abs(values[i]) * inv_sum); }
    }
    auto ratio = static_cast<double>(n) / sum;
    std::vector<uint32_t> over, under;
    over.reserve(n), under.reserve(n);
    table.assign(n, lr_alias_entry{});
    for (size_t i = 0; i < n; i++) {
        auto p = static_cast<float>(values[i] * ratio);
        table[i] = {p, static_cast<uint32_t>(i)};
        (p > 1.0f ? over : under).emplace_back(static_cast<uint32_t>(i));
    }
    while (!over.empty() && !under.empty()) {
        auto o = over.back();
        auto u = under.back();
        over.pop_back();
        under.pop_back();
        table[o].prob -= 1.0f - table[u].prob;
        table[u].alias = o;
        if (table[o].prob > 1.0f) {
            over.push_back(o);
        } else if (table[o].prob < 1.0f) {
            under.push_back(o);
        }
    }
    for (auto i : over) { table[i] = {1.0f, i}; }
    for (auto i : under) { table[i] = {1.0f, i}; }
}

lr_uint4 encode_instance_handle(uint32_t buffer_base, uint32_t flags, uint32_t surface_tag, uint32_t light_tag,
                                uint32_t medium_tag, uint32_t tri_count, float shadow_terminator,
                                float intersection_offset) {
    // src/base/shape.cpp:46-70, limits src/base/shape.h:124-137
    if (buffer_base > (1u << 22u) - 1u) { throw Error{"Invalid geometry buffer base."}; }
    if (flags > 1023u) { throw Error{"Invalid property flags."}; }
    if (surface_tag > 4095u) { throw Error{"Invalid surface tag (more than 4095 surfaces)."}; }
    if (light_tag > 4095u) { throw Error{"Invalid light tag (more than 4095 lights)."}; }
    if (medium_tag > 255u) { throw Error{"Invalid medium tag."}; }
    auto fixed = [](float x) {
        x = std::clamp(x, 0.f, 1.f);
        constexpr auto scale = 1.f / 65536.f;
        return static_cast<uint32_t>(std::clamp(std::round(x / scale), 0.f, 65535.f));
    };
    lr_uint4 h{};
    h.x = (buffer_base << 10u) | flags;
    h.y = (surface_tag << 12u) | light_tag | (medium_tag << 24u);
    h.z = tri_count;
    h.w = (fixed(shadow_terminator) << 16u) | fixed(intersection_offset);
    return h;
}

lr_scene SceneData::view(size_t camera_index) const {
    lr_scene s{};
    s.vertices = vertices.data(), s.vertex_count = vertices.size();
    s.triangles = triangles.data(), s.triangle_count = triangles.size();
    s.tri_alias = tri_alias.data();
    s.tri_pdf = tri_pdf.data();
    s.meshes = meshes.data(), s.mesh_count = static_cast<uint32_t>(meshes.size());
    s.instances = instances.data(), s.instance_count = static_cast<uint32_t>(instances.size());
    s.light_instances = light_instances.data(), s.light_instance_count = static_cast<uint32_t>(light_instances.size());
    s.surfaces = surfaces.data(), s.surface_count = static_cast<uint32_t>(surfaces.size());
    s.lights = lights.data(), s.light_count = static_cast<uint32_t>(lights.size());
    s.textures = textures.data(), s.texture_count = static_cast<uint32_t>(textures.size());
    s.texels = texels.data(), s.texel_count = texels.size() / 4u;
    s.environment = environment;
    s.environment.alias = env_alias.empty() ? nullptr : env_alias.data();
    s.environment.pdf = env_pdf.empty() ? nullptr : env_pdf.data();
    env_children_view = env_children;
    for (size_t i = 0; i < env_children_view.size(); i++) {
        env_children_view[i].alias = env_child_alias[i].empty() ? nullptr : env_child_alias[i].data();
        env_children_view[i].pdf = env_child_pdf[i].empty() ? nullptr : env_child_pdf[i].data();
    }
    s.environment_children = env_children_view.empty() ? nullptr : env_children_view.data();
    s.environment_child_count = static_cast<uint32_t>(env_children_view.size());
    if (camera_index >= cameras.size()) { throw Error{"Camera index out of range."}; }
    s.camera = cameras[camera_index].camera;
    s.filter = cameras[camera_index].filter;
    s.film = cameras[camera_index].film;
    s.sampler = sampler;
    s.sampler.spp = s.camera.spp;// sampler()->reset(.., resolution, pixel_count, spp), integrator.cpp:59
    s.sampler.scale = next_pow2(std::max(s.camera.width, s.camera.height));
    if (sampler.tile_size[0] != 0u) {// TileSharedSamplerInstance::reset (tile_shared.cpp:44-50): the base sees the tile grid as its resolution
        const auto tw = std::min(s.camera.width, sampler.tile_size[0]), th = std::min(s.camera.height, sampler.tile_size[1]);
        s.sampler.tile_size[0] = tw, s.sampler.tile_size[1] = th;
        s.sampler.scale = next_pow2(std::max((s.camera.width + tw - 1u) / tw, (s.camera.height + th - 1u) / th));
    }
    if (sampler.kind == LR_SAMPLER_SOBOL || sampler.kind == LR_SAMPLER_PADDED_SOBOL) {
        s.sampler.sobol_matrices = sobol_matrices.data();
        if (sampler.kind == LR_SAMPLER_SOBOL) {
            if (s.sampler.scale > 0xffffu) { throw Error{"Sobol sampler scale is too large."}; }// sobol.cpp:120
            auto m = 0u;
            while ((1u << m) < s.sampler.scale) { m++; }
            if (m >= 1u) {// m == 0 (1x1 film) never reads the tables (sobol.cpp:68)
                s.sampler.vdc_sobol = vdc_sobol.data() + static_cast<size_t>(m - 1u) * LR_SOBOL_MATRIX_SIZE;
                s.sampler.vdc_sobol_inv = vdc_sobol_inv.data() + static_cast<size_t>(m - 1u) * LR_SOBOL_MATRIX_SIZE;
            }
        }
    }
    s.integrator = integrator;
    s.media = media.empty() ? nullptr : media.data(), s.medium_count = static_cast<uint32_t>(media.size());
    s.accel.nodes = bvh_nodes.empty() ? nullptr : bvh_nodes.data();
    s.accel.node_count = static_cast<uint32_t>(bvh_nodes.size());
    s.accel.triangles = bvh_triangles.empty() ? nullptr : bvh_triangles.data();
    s.accel.triangle_count = static_cast<uint32_t>(bvh_triangles.size());
    for (auto i = 0; i < 3; i++) { s.accel.world_min[i] = world_min[i], s.accel.world_max[i] = world_max[i]; }
    s.any_non_opaque = any_non_opaque ? 1u : 0u;
    return s;
}

namespace {

float3 to_float3(const std::vector<double> &v) {
    return {static_cast<float>(v[0]), static_cast<float>(v[1]), static_cast<float>(v[2])};
}

float3 float3_or(const NodeDesc *d, const std::string &name, float3 dv) {
    auto v = d->vector_opt(name, 3u);
    return v ? to_float3(*v) : dv;
}

void store_matrix(float *dst, const float4x4 &m) {
    for (auto c = 0; c < 4; c++) {
        for (auto r = 0; r < 4; r++) { dst[c * 4 + r] = m[c][r]; }
    }
}

uint64_t fnv1a(const void *data, size_t size, uint64_t h = 0xcbf29ce484222325ull) {
    auto p = static_cast<const uint8_t *>(data);
    for (size_t i = 0; i < size; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

// 602.785 / 539.285 / 445.772 nm samples of the reference's built-in conductor tables
// (src/surfaces/metal_ior.inl.h through SPD::sample, src/base/spd.cpp:91-99: 5 nm LUT,
// linear interpolation); generated by tools/extract_metal_ior.py.
struct MetalIOR {
    const char *name;
    float n[3], k[3];
};
#include "metal_ior_rgb.inl.h"

const MetalIOR *find_metal(std::string name) {
    for (auto &c : name) { c = static_cast<char>(std::tolower(c)); }
    static const std::unordered_map<std::string, const char *> aliases{
        {"ag", "Ag"}, {"silver", "Ag"}, {"al", "Al"}, {"aluminium", "Al"}, {"au", "Au"}, {"gold", "Au"},
        {"cu", "Cu"}, {"copper", "Cu"}, {"cuzn", "CuZn"}, {"cu-zn", "CuZn"}, {"brass", "CuZn"},
        {"fe", "Fe"}, {"iron", "Fe"}, {"ti", "Ti"}, {"titanium", "Ti"}, {"v", "V"}, {"vanadium", "V"},
        {"vn", "VN"}, {"li", "Li"}, {"lithium", "Li"}, {"cr", "Cr"}, {"chromium", "Cr"}};
    auto it = aliases.find(name);
    if (it == aliases.end()) { return nullptr; }
    for (auto &m : metal_ior_rgb) {
        if (std::strcmp(m.name, it->second) == 0) { return &m; }
    }
    return nullptr;
}

// ------------------------------------------------------------------ builder

class Builder {
    const SceneDesc &_desc;
    SceneData &_out;
    std::unordered_map<const NodeDesc *, int32_t> _texture_ids;
    std::unordered_map<const NodeDesc *, uint32_t> _surface_tags;
    std::unordered_map<const NodeDesc *, uint32_t> _light_tags;
    std::unordered_map<const NodeDesc *, uint32_t> _shape_meshes;// shape node -> mesh index
    std::unordered_map<const NodeDesc *, uint32_t> _shape_props;
    std::unordered_map<uint64_t, uint32_t> _mesh_cache;          // content hash -> mesh index
    std::unordered_map<const NodeDesc *, bool> _light_null, _surface_null;
    float _scene_shadow_terminator{0.f};
    float _scene_intersection_offset{0.f};
    std::vector<float4x4> _transform_stack;// TransformTree (src/base/transform.cpp:25-59)
    struct ChainEntry { const NodeDesc *node; bool animated; };
    std::vector<ChainEntry> _transform_chain;// the non-identity transform nodes above the current shape, root first

    static void _check_tag(const NodeDesc *d, Tag tag) {
        if (d->tag() != tag && d->tag() != Tag::INTERNAL) {
            throw Error{std::string{"Invalid tag of scene description node '"} + d->identifier() + "' (expected " +
                        tag_description(tag) + ", got " + tag_description(d->tag()) + "). [" + d->location() + "]"};
        }
        if (!d->is_defined() && d->tag() != Tag::INTERNAL) {
            throw Error{"Undefined scene description node '" + d->identifier() + "'."};
        }
    }

public:
    Builder(const SceneDesc &desc, SceneData &out) : _desc{desc}, _out{out} {}

    // ---------------- transforms (src/transforms/*.cpp)
    float4x4 transform_matrix(const NodeDesc *d) {
        if (d == nullptr) { return float4x4::identity(); }
        _check_tag(d, Tag::TRANSFORM);
        auto &impl = d->impl_type();
        if (impl == "identity") { return float4x4::identity(); }
        if (impl == "matrix") {// matrix.cpp:15-42 (row-major text, transposed on load)
            auto m = d->float_list_or_empty("m");
            auto out = float4x4::identity();
            if (m.size() == 16u) {
                if (!(m[12] == 0.f && m[13] == 0.f && m[14] == 0.f && m[15] == 1.f)) {
                    log_warning("Expected affine transform matrices; the last row is fixed. [" + d->location() + "]");
                    m[12] = 0.f, m[13] = 0.f, m[14] = 0.f, m[15] = 1.f;
                }
                for (auto row = 0; row < 4; row++) {
                    for (auto col = 0; col < 4; col++) { out[col][row] = m[static_cast<size_t>(row * 4 + col)]; }
                }
            } else if (!m.empty()) {
                throw Error{"Invalid matrix entries. [" + d->location() + "]"};
            }
            return out;
        }
        if (impl == "srt") {// srt.cpp:14-24
            auto s3 = d->vector_opt("scale", 3u);
            auto scale = s3 ? to_float3(*s3) : make_float3(d->float_or("scale", 1.f));
            auto r = d->vector_opt("rotate", 4u);
            float4 rot = r ? float4{static_cast<float>((*r)[0]), static_cast<float>((*r)[1]),
                                    static_cast<float>((*r)[2]), static_cast<float>((*r)[3])} :
                             float4{0.f, 0.f, 1.f, 0.f};
            auto t = float3_or(d, "translate", make_float3(0.f));
            return translation(t) * rotation(normalize(float3{rot.x, rot.y, rot.z}), radians(rot.w)) * scaling(scale);
        }
        if (impl == "view") {// view.cpp:18-42
            auto origin = d->vector_opt("origin", 3u);
            auto o = origin ? to_float3(*origin) : float3_or(d, "position", make_float3(0.f));
            auto front = float3_or(d, "front", {0.f, 0.f, -1.f});
            auto up = float3_or(d, "up", {0.f, 1.f, 0.f});
            auto w = normalize(-front);
            auto u = normalize(cross(up, w));
            auto v = normalize(cross(w, u));
            return {{make_float4(u, 0.f), make_float4(v, 0.f), make_float4(w, 0.f), make_float4(o, 1.f)}};
        }
        if (impl == "stack") {// stack.cpp:22-36: m = t_i * m
            auto m = float4x4::identity();
            for (auto c : d->node_list_or_empty("transforms")) { m = transform_matrix(c) * m; }
            return m;
        }
        if (impl == "lerp") { return evaluate_xform(_out, compile_transform(d), 0.f); }
        throw Error{"Unknown transform implementation '" + impl + "'. [" + d->location() + "]"};
    }

    // Transform::is_static (lerp.cpp:69, stack.cpp:31)
    bool transform_is_animated(const NodeDesc *d) {
        if (d == nullptr) { return false; }
        auto &impl = d->impl_type();
        if (impl == "lerp") { return true; }
        if (impl == "stack") {
            for (auto c : d->node_list_or_empty("transforms")) {
                if (transform_is_animated(c)) { return true; }
            }
        }
        return false;
    }

    uint32_t static_xform(const float4x4 &m) {
        XformNode n;
        n.m = m;
        _out.xforms.emplace_back(std::move(n));
        return static_cast<uint32_t>(_out.xforms.size() - 1u);
    }

    // compile a transform node into SceneData::xforms so that it can be evaluated at any time later on
    uint32_t compile_transform(const NodeDesc *d) {
        if (!transform_is_animated(d)) { return static_xform(transform_matrix(d)); }
        _check_tag(d, Tag::TRANSFORM);
        XformNode n;
        n.is_static = false;
        if (d->impl_type() == "stack") {// stack.cpp:22-47
            n.kind = XformNode::STACK;
            for (auto c : d->node_list_or_empty("transforms")) { n.children.push_back(compile_transform(c)); }
        } else {// lerp.cpp:26-66
            n.kind = XformNode::LERP;
            auto nodes = d->node_list_required("transforms");
            auto times = d->float_list_or_empty("time_points");
            if (nodes.size() != times.size()) { throw Error{"Number of transforms and number of time points mismatch. [" + d->location() + "]"}; }
            if (nodes.empty()) { throw Error{"Empty transform list. [" + d->location() + "]"}; }
            std::vector<uint32_t> indices(times.size());
            std::iota(indices.begin(), indices.end(), 0u);
            std::sort(indices.begin(), indices.end(), [&](auto a, auto b) { return times[a] < times[b]; });
            if (auto it = std::unique(indices.begin(), indices.end(), [&](auto a, auto b) { return times[a] == times[b]; }); it != indices.end()) {
                log_warning("Duplicate time points (count = " + std::to_string(std::distance(it, indices.end())) + ") in LerpTransform will be removed. [" + d->location() + "]");
                indices.erase(it, indices.end());
            }
            for (auto i : indices) {
                n.times.push_back(times[i]);
                n.children.push_back(compile_transform(nodes[i]));
            }
        }
        _out.xforms.emplace_back(std::move(n));
        return static_cast<uint32_t>(_out.xforms.size() - 1u);
    }

    // ---------------- textures
    // SwizzleTexture over texture `base` (src/textures/swizzle.cpp), materialised: returns the id of a plain texture
    int32_t swizzled_texture(int32_t base, const std::vector<uint32_t> &channels, const NodeDesc *d) {
        auto n = channels.size();
        auto b = _out.textures[static_cast<size_t>(base)];// (copy: _out.textures grows below)
        auto pick = [&](const float in[4], float out[4], float pad_w) {// the value SwizzleTextureInstance::evaluate returns
            for (size_t i = 0; i < 4u; i++) { out[i] = i < n ? in[channels[i]] : (n == 1u ? in[channels[0]] : (i == 3u ? pad_w : 0.f)); }
        };
        lr_texture t = b;
        t.channels = static_cast<uint32_t>(n);
        if (b.kind == LR_TEX_CONSTANT) {
            float out[4];
            pick(b.v, out, 1.f);
            for (size_t i = 0; i < 4u; i++) { t.v[i] = i < n ? out[i] : 0.f; }// (evaluate_static leaves the unused channels 0)
        } else if (b.kind == LR_TEX_IMAGE) {
            // per-channel decode parameters move with their channel; the padded 0 / 1 pass through every encoding unchanged
            float scale[4];
            pick(b.scale, scale, 1.f);
            for (size_t i = n; i < 4u && n != 1u; i++) { scale[i] = 1.f; }
            if (b.encoding == LR_TEX_ENC_GAMMA) {// gamma[min(channel, 2)]: the alpha slot shares the blue exponent
                float g[4];
                for (size_t i = 0; i < 4u; i++) { g[i] = b.gamma[std::min<size_t>(i < n ? channels[i] : (n == 1u ? channels[0] : i), 2u)]; }
                if (n == 4u && g[3] != g[2]) { throw Error{"Swizzle of a gamma-encoded image whose alpha would need its own exponent is not supported. [" + d->location() + "]"}; }
                t.gamma[0] = g[0], t.gamma[1] = g[1], t.gamma[2] = g[2];
            }
            std::copy_n(scale, 4, t.scale);
            auto texels = static_cast<size_t>(b.width) * b.height;
            t.texel_offset = _out.texels.size() / 4u;
            _out.texels.resize(_out.texels.size() + texels * 4u);
            for (size_t k = 0; k < texels; k++) {
                float in[4], out[4];
                std::copy_n(_out.texels.data() + (b.texel_offset + k) * 4u, 4, in);
                pick(in, out, 1.f);
                std::copy_n(out, 4, _out.texels.data() + (t.texel_offset + k) * 4u);
            }
        } else {// checkerboard: swizzle the children (a missing child is the default on = 1 / off = (0, 0, 0, 1), checkerboard.cpp)
            for (auto k = 0; k < 2; k++) {
                auto child = b.child[k];
                if (child < 0) {
                    lr_texture c{};
                    c.kind = LR_TEX_CONSTANT, c.channels = 4u;
                    c.v[0] = c.v[1] = c.v[2] = k == 0 ? 1.f : 0.f, c.v[3] = 1.f;
                    c.child[0] = c.child[1] = -1;
                    child = static_cast<int32_t>(_out.textures.size());
                    _out.textures.emplace_back(c);
                }
                t.child[k] = swizzled_texture(child, channels, d);
            }
        }
        auto id = static_cast<int32_t>(_out.textures.size());
        _out.textures.emplace_back(t);
        return id;
    }

    int32_t load_texture(const NodeDesc *d) {
        if (d == nullptr) { return -1; }
        _check_tag(d, Tag::TEXTURE);
        if (auto it = _texture_ids.find(d); it != _texture_ids.end()) { return it->second; }
        lr_texture t{};
        t.child[0] = t.child[1] = -1;
        auto &impl = d->impl_type();
        if (impl == "constant") {// constant.cpp:21-42
            auto scale = d->float_or("scale", 1.f);
            auto v = d->float_list_or_empty("v");
            if (v.empty()) {
                log_warning("No value for ConstantTexture. Fallback to single-channel zero. [" + d->location() + "]");
                v.emplace_back(0.f);
            } else if (v.size() > 4u) {
                log_warning("Too many values for ConstantTexture; extra values are discarded. [" + d->location() + "]");
                v.resize(4u);
            }
            t.kind = LR_TEX_CONSTANT;
            t.channels = static_cast<uint32_t>(v.size());
            for (size_t i = 0; i < v.size(); i++) { t.v[i] = scale * v[i]; }
        } else if (impl == "image") {// image.cpp:47-112
            t.kind = LR_TEX_IMAGE;
            auto filter = d->string_or("filter", "bilinear");
            auto address = d->string_or("address", "repeat");
            for (auto &c : filter) { c = static_cast<char>(std::tolower(c)); }
            for (auto &c : address) { c = static_cast<char>(std::tolower(c)); }
            if (address == "zero") { t.address = LR_TEX_ADDR_ZERO; }
            else if (address == "edge") { t.address = LR_TEX_ADDR_EDGE; }
            else if (address == "mirror") { t.address = LR_TEX_ADDR_MIRROR; }
            else if (address == "repeat") { t.address = LR_TEX_ADDR_REPEAT; }
            else { throw Error{"Invalid texture address mode '" + address + "'. [" + d->location() + "]"}; }
            if (filter == "point") { t.filter = LR_TEX_FILTER_POINT; }
            else if (filter == "bilinear" || filter == "trilinear" || filter == "anisotropic" || filter == "aniso") {
                t.filter = LR_TEX_FILTER_BILINEAR;// the reference samples LOD 0 only (image.cpp:166 "TODO: LOD")
            } else { throw Error{"Invalid texture filter mode '" + filter + "'. [" + d->location() + "]"}; }
            auto s2 = d->vector_opt("uv_scale", 2u);
            auto o2 = d->vector_opt("uv_offset", 2u);
            auto s1 = s2 ? 1.f : d->float_or("uv_scale", 1.f), o1 = o2 ? 0.f : d->float_or("uv_offset", 0.f);
            t.uv_scale[0] = s2 ? static_cast<float>((*s2)[0]) : s1, t.uv_scale[1] = s2 ? static_cast<float>((*s2)[1]) : s1;
            t.uv_offset[0] = o2 ? static_cast<float>((*o2)[0]) : o1, t.uv_offset[1] = o2 ? static_cast<float>((*o2)[1]) : o1;
            auto path = d->path_or("file");
            if (path.empty()) { throw Error{"No valid values given for property 'file'. [" + d->location() + "]"}; }
            auto ext = fs::path{path}.extension().string();
            for (auto &c : ext) { c = static_cast<char>(std::tolower(c)); }
            auto encoding = d->string_or("encoding", (ext == ".exr" || ext == ".hdr") ? "linear" : "sRGB");
            for (auto &c : encoding) { c = static_cast<char>(std::tolower(c)); }
            t.gamma[0] = t.gamma[1] = t.gamma[2] = 1.f;
            if (encoding == "srgb") { t.encoding = LR_TEX_ENC_SRGB; }
            else if (encoding == "gamma") {
                t.encoding = LR_TEX_ENC_GAMMA;
                t.gamma[0] = t.gamma[1] = t.gamma[2] = d->float_or("gamma", 1.f);
            } else {
                if (encoding != "linear") { log_warning("Unknown texture encoding '" + encoding + "'; using linear."); }
                t.encoding = LR_TEX_ENC_LINEAR;
            }
            auto scale = d->float_or("scale", 1.f);
            for (auto &s : t.scale) { s = scale; }
            auto image = load_image(path);
            t.width = image.width, t.height = image.height, t.channels = image.channels;
            t.texel_offset = _out.texels.size() / 4u;
            _out.texels.insert(_out.texels.end(), image.pixels.begin(), image.pixels.end());
        } else if (impl == "checkerboard") {
            t.kind = LR_TEX_CHECKERBOARD;
            t.child[0] = load_texture(d->node_or_null("on"));
            t.child[1] = load_texture(d->node_or_null("off"));
            t.checker_scale = d->float_or("scale", 1.f);
            for (auto c : t.child) {// (the device evaluates one level of nesting: constant or image children)
                if (c >= 0 && _out.textures[static_cast<size_t>(c)].kind != LR_TEX_CONSTANT && _out.textures[static_cast<size_t>(c)].kind != LR_TEX_IMAGE) {
                    throw Error{"Checkerboard children must be Constant or Image textures. [" + d->location() + "]"};
                }
            }
            auto child_channels = [&](int32_t c) { return c < 0 ? 4u : _out.textures[static_cast<size_t>(c)].channels; };
            t.channels = std::min(child_channels(t.child[0]), child_channels(t.child[1]));// checkerboard.cpp:38-48
        } else if (impl == "swizzle") {// swizzle.cpp:18-53
            auto base = load_texture(d->node("base"));
            std::vector<uint32_t> channels;
            if (auto s = d->string_opt("swizzle")) {
                for (auto c : *s) {
                    switch (c) {
                        case 'r': case 'x': channels.push_back(0u); break;
                        case 'g': case 'y': channels.push_back(1u); break;
                        case 'b': case 'z': channels.push_back(2u); break;
                        case 'a': case 'w': channels.push_back(3u); break;
                        default: throw Error{std::string{"Invalid swizzle channel '"} + c + "'. [" + d->location() + "]"};
                    }
                }
            } else if (d->has_property("swizzle")) {
                channels = d->uint_list("swizzle");
            } else {
                channels = {0u, 1u, 2u, 3u};
            }
            if (channels.size() > 4u) {
                log_warning("Too many swizzle channels (count = " + std::to_string(channels.size()) + ") for SwizzleTexture. Additional channels will be discarded. [" + d->location() + "]");
                channels.resize(4u);
            }
            if (channels.empty()) { throw Error{"Empty swizzle. [" + d->location() + "]"}; }
            for (auto c : channels) {
                if (c >= 4u) { throw Error{"Swizzle channel '" + std::to_string(c) + "' out of range. [" + d->location() + "]"}; }
            }
            // The swizzle is applied HERE, once, to the base texture's data — a swizzled constant is a constant
            // (evaluate_static, :60-67), a swizzled image is the image with its texel channels (and per-channel scale)
            // permuted, a swizzled checkerboard is the checkerboard of its swizzled children — so the device interprets no
            // Swizzle node per lookup (that cost the LEAN kernel 2.7 % on a scene without a single texture: registers).
            // Per lookup the reference computes (SwizzleTextureInstance::evaluate, :100-113): 1 channel -> (s0, s0, s0, s0),
            // 2 -> (s0, s1, 0, 1), 3 -> (s0, s1, s2, 1), 4 -> all four; permuting before the bilinear filter and the
            // per-channel decode gives the same values because both act on each channel separately.
            auto id = swizzled_texture(base, channels, d);
            _texture_ids.emplace(d, id);
            return id;
        } else {
            throw Error{"Unsupported texture implementation '" + impl + "'. [" + d->location() + "]"};
        }
        auto id = static_cast<int32_t>(_out.textures.size());
        _out.textures.emplace_back(t);
        _texture_ids.emplace(d, id);
        return id;
    }

    bool texture_is_black(int32_t id) const {// Texture::is_black
        if (id < 0) { return false; }
        auto &t = _out.textures[static_cast<size_t>(id)];
        if (t.kind == LR_TEX_CONSTANT) { return t.v[0] == 0.f && t.v[1] == 0.f && t.v[2] == 0.f && t.v[3] == 0.f; }
        if (t.kind == LR_TEX_IMAGE) { return t.scale[0] == 0.f; }
        return false;
    }

    // ---------------- surfaces
    static bool surface_is_null(const NodeDesc *d) { return d == nullptr || d->impl_type() == "null"; }

    uint32_t register_surface(const NodeDesc *d) {// Pipeline::register_surface, pipeline.cpp:20-26
        _check_tag(d, Tag::SURFACE);
        if (auto it = _surface_tags.find(d); it != _surface_tags.end()) { return it->second; }
        lr_surface s{};
        for (auto &t : s.tex) { t = -1; }
        s.alpha_tex = s.normal_tex = -1;
        s.normal_strength = 1.f;
        auto &impl = d->impl_type();
        auto tex = [&](const char *name) { return load_texture(d->node_or_null(name)); };
        auto remap = d->bool_or("remap_roughness", true);
        if (remap) { s.flags |= LR_SURFACE_FLAG_REMAP_ROUGHNESS; }
        auto wrappers = true;
        auto children = std::array<uint32_t, 2>{0u, 0u};
        if (impl == "matte") {// matte.cpp:25-26
            s.kind = LR_SURFACE_MATTE;
            s.tex[0] = tex("Kd"), s.tex[1] = tex("sigma");
            if (texture_is_black(s.tex[1])) { s.tex[1] = -1; }// `_sigma && !_sigma->node()->is_black()`, matte.cpp:125
        } else if (impl == "mirror") {// mirror.cpp:23-28
            s.kind = LR_SURFACE_MIRROR;
            auto color = d->node_or_null("color");
            if (color == nullptr) { color = d->node_or_null("Kd"); }
            s.tex[0] = load_texture(color), s.tex[1] = tex("roughness");
        } else if (impl == "glass") {// glass.cpp:57-81
            s.kind = LR_SURFACE_GLASS;
            s.tex[0] = tex("Kr"), s.tex[1] = tex("Kt"), s.tex[2] = tex("roughness");
            if (auto name = d->is_string_property("eta") ? d->string_or("eta") : std::string{}; !name.empty()) {
                static const std::unordered_map<std::string, std::array<float, 3>> builtin{
                    {"bk7", {1.5140814565098806f, 1.5165571794092296f, 1.5223224896834853f}},
                    {"baf10", {1.665552211440938f, 1.6698355055693541f, 1.680044942398477f}},
                    {"fk51a", {1.4846524304153899f, 1.486399794903804f, 1.4904761200647965f}},
                    {"lasf9", {1.8422161861952726f, 1.8499302872507852f, 1.8690214187351977f}},
                    {"sf5", {1.6663001504164476f, 1.6723956342450994f, 1.6875677127863755f}},
                    {"sf10", {1.720557014155419f, 1.7279815121138318f, 1.7465722778961204f}},
                    {"sf11", {1.7754589288508518f, 1.7842240428294434f, 1.8065917880168352f}},
                    {"diamond", {2.410486117067883f, 2.4164392529553234f, 2.431466411471524f}},
                    {"ice", {1.3077084260466776f, 1.3095827995357034f, 1.3137441348487024f}},
                    {"quartz", {1.4562471554155727f, 1.4582990183632742f, 1.4630571022260817f}},
                    {"salt", {1.5404463273409252f, 1.5441711411436845f, 1.5531314007749342f}},
                    {"sapphire", {1.764706495252994f, 1.7680107911479397f, 1.7755615929437936f}}};
                for (auto &c : name) { c = static_cast<char>(std::tolower(c)); }
                if (auto it = builtin.find(name); it != builtin.end()) {
                    lr_texture t{};
                    t.kind = LR_TEX_CONSTANT, t.channels = 3u;
                    t.v[0] = it->second[0], t.v[1] = it->second[1], t.v[2] = it->second[2];
                    t.child[0] = t.child[1] = -1;
                    s.tex[3] = static_cast<int32_t>(_out.textures.size());
                    _out.textures.emplace_back(t);
                } else {
                    log_warning("Unknown built-in glass '" + name + "'. Fallback to constant IOR = 1.5.");
                }
            } else {
                s.tex[3] = tex("eta");
                if (s.tex[3] >= 0) {
                    auto ch = _out.textures[static_cast<size_t>(s.tex[3])].channels;
                    if (ch == 2u || ch == 4u) { throw Error{"Invalid channel count for GlassSurface::eta. [" + d->location() + "]"}; }
                }
            }
            wrappers = false;// glass has only the normal-map wrapper (glass.cpp:281-282)
            s.normal_tex = tex("normal_map");
            s.normal_strength = d->float_or("normal_map_strength", 1.f);
        } else if (impl == "plastic" || impl == "substrate") {// plastic.cpp:53-60
            s.kind = LR_SURFACE_PLASTIC;
            s.tex[0] = tex("Kd"), s.tex[1] = tex("roughness"), s.tex[2] = tex("sigma_a");
            s.tex[3] = tex("eta"), s.tex[4] = tex("thickness");
        } else if (impl == "metal") {// metal.cpp:54-152
            s.kind = LR_SURFACE_METAL;
            s.tex[0] = tex("Kd"), s.tex[1] = tex("roughness");
            const MetalIOR *ior = nullptr;
            if (auto name = d->is_string_property("eta") ? d->string_or("eta") : std::string{}; !name.empty()) {
                ior = find_metal(name);
                if (ior == nullptr) {
                    log_warning("Unknown metal '" + name + "'. Fallback to Aluminium. [" + d->location() + "]");
                    ior = find_metal("al");
                }
                for (auto i = 0; i < 3; i++) { s.f[i] = ior->n[i], s.f[3 + i] = ior->k[i]; }
            } else {
                auto eta = d->float_list("eta");
                if (eta.size() % 3u != 0u || eta.size() < 6u) { throw Error{"Invalid eta list size. [" + d->location() + "]"}; }
                auto count = eta.size() / 3u;
                std::vector<float> lambda(count), n(count), k(count);
                for (size_t i = 0; i < count; i++) { lambda[i] = eta[i * 3u], n[i] = eta[i * 3u + 1u], k[i] = eta[i * 3u + 2u]; }
                if (!std::is_sorted(lambda.begin(), lambda.end())) { throw Error{"Unsorted wavelengths in eta list. [" + d->location() + "]"}; }
                if (lambda.front() > 360.f || lambda.back() < 830.f) { throw Error{"Invalid wavelength range in eta list. [" + d->location() + "]"}; }
                // 5 nm LUT (metal.cpp:131-147) then SPD::sample at the RGB peak wavelengths
                auto lut = [&](uint32_t i, const std::vector<float> &y) {
                    auto wavelength = static_cast<float>(i * 5u + 360u);
                    auto lb = std::lower_bound(lambda.begin(), lambda.end(), wavelength);
                    auto index = std::clamp(static_cast<size_t>(std::distance(lambda.begin(), lb)), size_t{1}, lambda.size() - 1u);
                    auto t = (wavelength - lambda[index - 1u]) / (lambda[index] - lambda[index - 1u]);
                    return y[index - 1u] + t * (y[index] - y[index - 1u]);
                };
                constexpr std::array<float, 3> peaks{602.785f, 539.285f, 445.772f};
                for (auto c = 0; c < 3; c++) {
                    auto t = (std::clamp(peaks[static_cast<size_t>(c)], 360.f, 830.f) - 360.f) / 5.f;
                    auto i = static_cast<uint32_t>(std::min(t, 93.f));
                    auto fr = t - std::floor(t);
                    s.f[c] = lut(i, n) + fr * (lut(i + 1u, n) - lut(i, n));
                    s.f[3 + c] = lut(i, k) + fr * (lut(i + 1u, k) - lut(i, k));
                }
            }
        } else if (impl == "disney") {// disney.cpp:36-58
            s.kind = LR_SURFACE_DISNEY;
            auto color = d->node_or_null("color");
            if (color == nullptr) { color = d->node_or_null("Kd"); }
            s.tex[0] = load_texture(color);
            static constexpr std::array<const char *, 12> names{
                "metallic", "eta", "roughness", "specular_tint", "anisotropic", "sheen", "sheen_tint",
                "clearcoat", "clearcoat_gloss", "specular_trans", "flatness", "diffuse_trans"};
            for (size_t i = 0; i < names.size(); i++) { s.tex[i + 1u] = tex(names[i]); }
            if (d->bool_or("thin", false)) { s.flags |= LR_SURFACE_FLAG_THIN; }
            // used lobes (disney.cpp:969-991) from the static blackness of the parameter textures
            auto present = [&](int slot) { return s.tex[slot] >= 0 && !texture_is_black(s.tex[slot]); };
            auto lobes = 0u;
            if (s.tex[0] < 0 || !texture_is_black(s.tex[0])) {
                lobes |= 1u | 2u;// diffuse, retro
                if (present(6)) { lobes |= 8u; }// sheen
                if (present(11)) { lobes |= 4u; }// fake subsurface (flatness)
            }
            lobes |= 32u;// specular
            if (present(8)) { lobes |= 16u; }  // clearcoat
            if (present(10)) { lobes |= 128u; }// specular transmission
            auto thin = (s.flags & LR_SURFACE_FLAG_THIN) != 0u;
            if (thin && present(12)) { lobes |= 64u; }// diffuse transmission (texture only built for thin, :1011)
            if (!thin) { s.tex[12] = -1; }
            s.u[2] = lobes;
            s.u[1] = (!thin && present(10)) ? 1u : 0u;// Surface::is_transmissive (disney.cpp:63-77)
        } else if (impl == "mix") {// mix.cpp:23-32
            s.kind = LR_SURFACE_MIX;
            auto a = d->node("a"), b = d->node("b");
            if (surface_is_null(a) || surface_is_null(b)) {
                throw Error{"Mix surface with a null child is not supported. [" + d->location() + "]"};
            }
            children = {register_surface(a), register_surface(b)};
            // nested Mix surfaces: the kernels walk up to LR_MIX_MAX_DEPTH levels below the root (dev_heavy.h: kMixMaxDepth); u[2] = depth.
            // A Layered child is a leaf of the tree (its own interfaces may be Mix trees again, counted from zero); it sends the
            // Mix through the general interpreter (u[2] != 0), whose leaves know the Layered closure.
            auto depth_of = [&](uint32_t c) {
                auto &cs = _out.surfaces[c];
                return cs.kind == LR_SURFACE_MIX ? 1u + cs.u[2] : (cs.kind == LR_SURFACE_LAYERED ? 1u : 0u);
            };
            s.u[2] = std::max(depth_of(children[0]), depth_of(children[1]));
            if (s.u[2] > static_cast<uint32_t>(LR_MIX_MAX_DEPTH)) {
                throw Error{"Mix surfaces nested more than " + std::to_string(LR_MIX_MAX_DEPTH) + " levels deep are not supported by the megakernel. [" + d->location() + "]"};
            }
            s.u[0] = children[0], s.u[1] = children[1];
            s.tex[0] = tex("ratio");
            wrappers = false;// NormalMapWrapper<MixSurface> only (mix.cpp:214-215)
            s.normal_tex = tex("normal_map");
            s.normal_strength = d->float_or("normal_map_strength", 1.f);
        } else if (impl == "layered") {// layered.cpp:107-126
            s.kind = LR_SURFACE_LAYERED;
            auto top = d->node("top"), bottom = d->node("bottom");
            if (surface_is_null(top) || surface_is_null(bottom)) { throw Error{"Creating closure for null LayeredSurface. [" + d->location() + "]"}; }
            children = {register_surface(top), register_surface(bottom)};
            for (auto c : children) {// interfaces: basic closures, Disney, Mix trees, and -- one level -- Layered surfaces again (dev_layered.h)
                if (surface_layered_levels(c) >= static_cast<uint32_t>(LR_LAYERED_MAX_LEVELS)) {
                    throw Error{"Layered surfaces nested more than " + std::to_string(LR_LAYERED_MAX_LEVELS) + " levels deep are not supported by the megakernel. [" + d->location() + "]"};
                }
            }
            s.u[0] = children[0], s.u[1] = children[1];
            s.tex[0] = tex("thickness"), s.tex[1] = tex("g"), s.tex[2] = tex("albedo");
            s.u[2] = d->uint_or("max_depth", 10u), s.u[3] = d->uint_or("samples", 1u);
            // TwoSidedWrapper (surface.h:300-309) populates the closure with the flipped frame and then again with the real
            // one, i.e. it has no effect on the closure; only the flag is recorded
            if (d->bool_or("two_sided", false)) { s.flags |= LR_SURFACE_FLAG_TWO_SIDED; }
        } else {
            throw Error{"Unknown surface implementation '" + impl + "'. [" + d->location() + "]"};
        }
        if (wrappers) {// NormalMapWrapper<OpacitySurfaceWrapper<...>>, surface.h:196-203,262-267
            auto alpha = d->node_or_null("alpha");
            if (alpha == nullptr) { alpha = d->node_or_null("opacity"); }
            s.alpha_tex = load_texture(alpha);
            s.normal_tex = tex("normal_map");
            s.normal_strength = d->float_or("normal_map_strength", 1.f);
        }
        auto tag = static_cast<uint32_t>(_out.surfaces.size());
        _out.surfaces.emplace_back(s);
        _surface_tags.emplace(d, tag);
        return tag;
    }

    uint32_t surface_layered_levels(uint32_t tag) const {// Layered surfaces on the deepest path below (and including) this one
        auto &s = _out.surfaces[tag];
        if (s.kind != LR_SURFACE_LAYERED && s.kind != LR_SURFACE_MIX) { return 0u; }
        return (s.kind == LR_SURFACE_LAYERED ? 1u : 0u) + std::max(surface_layered_levels(s.u[0]), surface_layered_levels(s.u[1]));
    }

    bool surface_maybe_non_opaque(uint32_t tag) const {// OpacitySurfaceWrapper::maybe_non_opaque
        auto &s = _out.surfaces[tag];
        if (s.kind == LR_SURFACE_MIX) { return surface_maybe_non_opaque(s.u[0]) || surface_maybe_non_opaque(s.u[1]); }// mix.cpp:59-61
        if (s.kind == LR_SURFACE_LAYERED && (surface_maybe_non_opaque(s.u[0]) || surface_maybe_non_opaque(s.u[1]))) { return true; }// layered.cpp:174-176
        if (s.alpha_tex < 0) { return false; }
        auto &t = _out.textures[static_cast<size_t>(s.alpha_tex)];
        // constant alpha >= 1 is treated as opaque (surface.h:204-216)
        if (t.kind == LR_TEX_CONSTANT && t.v[0] >= 1.f) { return false; }
        return true;
    }

    // ---------------- lights
    // Pipeline::register_medium (pipeline.cpp:36-42) + the Medium / PhaseFunction node constructors
    std::unordered_map<const NodeDesc *, uint32_t> _medium_tags;
    uint32_t register_medium(const NodeDesc *d) {
        _check_tag(d, Tag::MEDIUM);
        if (auto it = _medium_tags.find(d); it != _medium_tags.end()) { return it->second; }
        lr_medium m{};
        m.priority = d->uint_or("priority", 0u);// medium.cpp:13
        m.eta = 1.f;
        auto constant3 = [&](const char *name, float out[3], bool required) {
            auto t = d->node_or_null(name);
            if (t == nullptr) {
                if (required) { throw Error{std::string{name} + " must be specified as constant. [" + d->location() + "]"}; }
                return;
            }
            _check_tag(t, Tag::TEXTURE);
            if (t->impl_type() != "constant") { throw Error{std::string{name} + " must be specified as constant. [" + t->location() + "]"}; }
            auto &tex = _out.textures[static_cast<size_t>(load_texture(t))];
            for (auto c = 0; c < 3; c++) { out[c] = tex.channels == 1u ? tex.v[0] : tex.v[c]; }
        };
        if (d->impl_type() == "homogeneous") {// homogeneous.cpp:189-201
            m.kind = LR_MEDIUM_HOMOGENEOUS;
            m.eta = d->float_or("eta", 1.f);
            constant3("sigma_a", m.sigma_a, true);
            constant3("sigma_s", m.sigma_s, true);
            constant3("Le", m.le, false);
            auto pf = d->node_or_null("phasefunction");
            if (pf == nullptr) { throw Error{"Phase function must be specified. [" + d->location() + "]"}; }
            _check_tag(pf, Tag::PHASE_FUNCTION);
            if (pf->impl_type() != "henyeygreenstein") { throw Error{"Unknown phase function '" + pf->impl_type() + "'. [" + pf->location() + "]"}; }
            m.g = std::clamp(pf->float_or("g", 0.f), -1.f, 1.f);// henyey_greenstein.cpp:64
        } else if (d->impl_type() == "vacuum") {// vacuum.cpp:66-70
            m.kind = LR_MEDIUM_VACUUM;
            m.priority = LR_MEDIUM_VACUUM_PRIORITY;
        } else {
            throw Error{"Medium '" + d->impl_type() + "' is not implemented (Homogeneous, Vacuum, Null are). [" + d->location() + "]"};
        }
        if (_out.media.size() >= 255u) { throw Error{"Too many media (8-bit medium tag, shape.h:131)."}; }
        auto tag = static_cast<uint32_t>(_out.media.size());
        _out.media.emplace_back(m);
        _medium_tags.emplace(d, tag);
        return tag;
    }

    bool light_is_null(const NodeDesc *d) {
        if (d == nullptr || d->impl_type() == "null") { return true; }
        _check_tag(d, Tag::LIGHT);
        if (auto it = _light_null.find(d); it != _light_null.end()) { return it->second; }
        auto null = false;
        if (d->impl_type() == "diffuse") {// diffuse.cpp:20-30
            auto scale = std::max(d->float_or("scale", 1.f), 0.f);
            auto emission = d->node_or_null("emission");
            auto id = load_texture(emission ? emission : NodeDesc::shared_default(Tag::TEXTURE, "Constant"));
            null = scale == 0.f || texture_is_black(id);
        } else {
            throw Error{"Unknown light implementation '" + d->impl_type() + "'. [" + d->location() + "]"};
        }
        _light_null.emplace(d, null);
        return null;
    }

    uint32_t register_light(const NodeDesc *d) {// Pipeline::register_light, pipeline.cpp:28-34
        if (auto it = _light_tags.find(d); it != _light_tags.end()) { return it->second; }
        lr_light l{};
        l.kind = LR_LIGHT_DIFFUSE;
        auto emission = d->node_or_null("emission");
        l.emission_tex = load_texture(emission ? emission : NodeDesc::shared_default(Tag::TEXTURE, "Constant"));
        l.scale = std::max(d->float_or("scale", 1.f), 0.f);
        l.two_sided = d->bool_or("two_sided", false) ? 1u : 0u;
        auto tag = static_cast<uint32_t>(_out.lights.size());
        _out.lights.emplace_back(l);
        _light_tags.emplace(d, tag);
        return tag;
    }

    // ---------------- shapes
    struct ShapeInfo {
        bool is_mesh;
        bool visible;
        float shadow_terminator;
        float intersection_offset;
    };

    ShapeInfo shape_info(const NodeDesc *d) const {
        auto &impl = d->impl_type();
        ShapeInfo info{};
        info.is_mesh = impl == "mesh" || impl == "inlinemesh" || impl == "sphere" || impl == "loopsubdiv";
        if (!info.is_mesh && impl != "group" && impl != "instance") {
            throw Error{"Unsupported shape implementation '" + impl + "'. [" + d->location() + "]"};
        }
        info.visible = d->bool_or("visible", true);// VisibilityShapeWrapper, shape.h:104-115
        if (info.is_mesh) {                         // shape.h:66-102
            info.shadow_terminator = std::clamp(d->float_or("shadow_terminator", _scene_shadow_terminator), 0.f, 1.f);
            info.intersection_offset = std::clamp(d->float_or("intersection_offset", _scene_intersection_offset), 0.f, 1.f);
        }
        return info;
    }

    // Shape::mesh() of the mesh-like shapes (vertices, triangles, vertex property flags)
    LoadedMesh mesh_data(const NodeDesc *d) {
        LoadedMesh mesh;
        auto &impl = d->impl_type();
        if (impl == "inlinemesh") {// inline_mesh.cpp:21-58
            auto indices = d->uint_list("indices");
            auto positions = d->float_list("positions");
            auto normals = d->float_list_or_empty("normals");
            auto uvs = d->float_list_or_empty("uvs");
            if (indices.size() % 3u != 0u || positions.size() % 3u != 0u || normals.size() % 3u != 0u ||
                uvs.size() % 2u != 0u || (!normals.empty() && normals.size() != positions.size()) ||
                (!uvs.empty() && uvs.size() / 2u != positions.size() / 3u)) {
                throw Error{"Invalid vertex or triangle count. [" + d->location() + "]"};
            }
            mesh.properties = (!uvs.empty() ? uint32_t{LR_SHAPE_HAS_VERTEX_UV} : 0u) | (!normals.empty() ? uint32_t{LR_SHAPE_HAS_VERTEX_NORMAL} : 0u);
            auto vertex_count = positions.size() / 3u;
            mesh.triangles.resize(indices.size() / 3u);
            for (size_t i = 0; i < mesh.triangles.size(); i++) {
                mesh.triangles[i] = {indices[i * 3u], indices[i * 3u + 1u], indices[i * 3u + 2u]};
                if (indices[i * 3u] >= vertex_count || indices[i * 3u + 1u] >= vertex_count || indices[i * 3u + 2u] >= vertex_count) {
                    throw Error{"Triangle index out of range. [" + d->location() + "]"};
                }
            }
            mesh.vertices.resize(vertex_count);
            for (size_t i = 0; i < vertex_count; i++) {
                lr_vertex v{};
                v.px = positions[i * 3u], v.py = positions[i * 3u + 1u], v.pz = positions[i * 3u + 2u];
                if (normals.empty()) { v.nx = 0.f, v.ny = 0.f, v.nz = 1.f; }
                else { v.nx = normals[i * 3u], v.ny = normals[i * 3u + 1u], v.nz = normals[i * 3u + 2u]; }
                if (!uvs.empty()) { v.u = uvs[i * 2u], v.v = uvs[i * 2u + 1u]; }
                mesh.vertices[i] = v;
            }
        } else if (impl == "mesh") {// mesh.cpp:151-157
            auto path = d->path_or("file");
            if (path.empty()) { throw Error{"No valid values given for property 'file'. [" + d->location() + "]"}; }
            auto subdivision = d->uint_or("subdivision", 0u);// mesh.cpp:155: Catmull-Clark levels
            if (subdivision > 8u) { throw Error{"Mesh subdivision level " + std::to_string(subdivision) + " is out of range (at most 8: 4^8 quads per face). [" + d->location() + "]"}; }
            mesh = load_obj_mesh(path, d->bool_or("flip_uv", false), d->bool_or("drop_normal", false), d->bool_or("drop_uv", false), subdivision);
        } else if (impl == "sphere") {// sphere.cpp:113-117
            mesh = make_sphere_mesh(std::min(d->uint_or("subdivision", 0u), 8u));
        } else if (impl == "loopsubdiv") {// loop_subdiv.cpp:24-58
            auto base = d->node_or_null("mesh");
            if (base == nullptr) { base = d->node_or_null("shape"); }
            if (base == nullptr) { base = d->node("base"); }
            _check_tag(base, Tag::SHAPE);
            if (!shape_info(base).is_mesh) { throw Error{"LoopSubdiv only supports mesh shapes. [" + d->location() + "]"}; }
            auto level = std::min(d->uint_or("level", 1u), 10u);
            mesh = mesh_data(base);
            if (level == 0u) {
                log_warning("LoopSubdiv level is 0, which is equivalent to no subdivision. [" + d->location() + "]");
            } else {
                mesh = loop_subdivide(mesh.vertices, mesh.triangles, level);// (normals only: "TODO: preserve uv mapping", :17)
            }
        } else {
            throw Error{"Unknown mesh shape '" + impl + "'. [" + d->location() + "]"};
        }
        if (mesh.vertices.empty() || mesh.triangles.empty()) { throw Error{"Empty mesh. [" + d->location() + "]"}; }
        return mesh;
    }

    uint32_t load_mesh(const NodeDesc *d, uint32_t &properties) {
        if (auto it = _shape_meshes.find(d); it != _shape_meshes.end()) {
            properties = _shape_props.at(d);
            return it->second;
        }
        auto mesh = mesh_data(d);
        // dedup by content (geometry.cpp:53-57)
        auto hash = fnv1a(mesh.vertices.data(), mesh.vertices.size() * sizeof(lr_vertex));
        hash = fnv1a(mesh.triangles.data(), mesh.triangles.size() * sizeof(lr_triangle), hash);
        uint32_t index;
        if (auto it = _mesh_cache.find(hash); it != _mesh_cache.end()) {
            index = it->second;
        } else {
            index = static_cast<uint32_t>(_out.meshes.size());
            lr_mesh m{};
            m.vertex_offset = static_cast<uint32_t>(_out.vertices.size());
            m.vertex_count = static_cast<uint32_t>(mesh.vertices.size());
            m.triangle_offset = static_cast<uint32_t>(_out.triangles.size());
            m.triangle_count = static_cast<uint32_t>(mesh.triangles.size());
            _out.vertices.insert(_out.vertices.end(), mesh.vertices.begin(), mesh.vertices.end());
            _out.triangles.insert(_out.triangles.end(), mesh.triangles.begin(), mesh.triangles.end());
            // per-mesh area alias table (geometry.cpp:69-79)
            std::vector<float> areas(mesh.triangles.size());
            for (size_t i = 0; i < mesh.triangles.size(); i++) {
                auto t = mesh.triangles[i];
                auto &a = mesh.vertices[t.i0], &b = mesh.vertices[t.i1], &c = mesh.vertices[t.i2];
                float3 p0{a.px, a.py, a.pz}, p1{b.px, b.py, b.pz}, p2{c.px, c.py, c.pz};
                areas[i] = std::abs(length(cross(p1 - p0, p2 - p0)));
            }
            std::vector<lr_alias_entry> table;
            std::vector<float> pdf;
            create_alias_table(areas.data(), areas.size(), table, pdf);
            _out.tri_alias.insert(_out.tri_alias.end(), table.begin(), table.end());
            _out.tri_pdf.insert(_out.tri_pdf.end(), pdf.begin(), pdf.end());
            _out.meshes.emplace_back(m);
            _mesh_cache.emplace(hash, index);
        }
        _shape_meshes.emplace(d, index);
        _shape_props.emplace(d, mesh.properties);
        properties = mesh.properties;
        return index;
    }

    // Geometry::_process_shape, geometry.cpp:29-163
    void process_shape(const NodeDesc *d, const NodeDesc *overridden_surface, const NodeDesc *overridden_light,
                       bool overridden_visible, const NodeDesc *overridden_medium = nullptr) {
        _check_tag(d, Tag::SHAPE);
        auto info = shape_info(d);
        auto own_surface = d->node_or_null("surface");
        auto own_light = d->node_or_null("light");
        auto surface = overridden_surface == nullptr ? own_surface : overridden_surface;
        auto light = overridden_light == nullptr ? own_light : overridden_light;
        auto visible = overridden_visible && info.visible;
        auto own_medium = d->node_or_null("medium");
        auto medium = overridden_medium == nullptr ? own_medium : overridden_medium;// geometry.cpp:38
        auto local_node = d->node_or_null("transform");
        auto local = transform_matrix(local_node);// (animated transforms: value at time 0; set_scene_time finalises them)
        auto local_animated = transform_is_animated(local_node);
        if (info.is_mesh) {
            uint32_t vertex_props = 0u;
            auto mesh_index = load_mesh(d, vertex_props);
            auto &mesh = _out.meshes[mesh_index];
            auto instance_id = static_cast<uint32_t>(_out.instances.size());
            // TransformTree::leaf + Node::matrix: M = M_root * ... * M_leaf
            auto object_to_world = is_identity(local) && !local_animated ? _transform_stack.back() : _transform_stack.back() * local;
            if (local_animated || std::any_of(_transform_chain.begin(), _transform_chain.end(), [](auto &e) { return e.animated; })) {
                DynamicInstance dyn;
                dyn.instance = instance_id;
                for (auto &e : _transform_chain) { dyn.chain.push_back(compile_transform(e.node)); }
                if (local_animated || !is_identity(local)) { dyn.chain.push_back(compile_transform(local_node)); }
                _out.dynamic_instances.emplace_back(std::move(dyn));
            }
            for (uint32_t i = 0; i < mesh.vertex_count; i++) {
                auto &v = _out.vertices[mesh.vertex_offset + i];
                auto p = transform_point(object_to_world, {v.px, v.py, v.pz});
                for (auto a = 0; a < 3; a++) {
                    _out.world_min[a] = std::min(_out.world_min[a], p[a]);
                    _out.world_max[a] = std::max(_out.world_max[a], p[a]);
                }
            }
            auto properties = vertex_props;
            auto surface_tag = 0u, light_tag = 0u;
            if (!surface_is_null(surface)) {
                surface_tag = register_surface(surface);
                properties |= LR_SHAPE_HAS_SURFACE;
                if (surface_maybe_non_opaque(surface_tag)) {
                    properties |= LR_SHAPE_MAYBE_NON_OPAQUE;
                    _out.any_non_opaque = true;
                }
            }
            if (!light_is_null(light)) {
                light_tag = register_light(light);
                properties |= LR_SHAPE_HAS_LIGHT;
            }
            auto medium_tag = 0u;
            if (medium != nullptr && medium->impl_type() != "null") {// geometry.cpp:139-142
                medium_tag = register_medium(medium);
                properties |= LR_SHAPE_HAS_MEDIUM;
            }
            // u16 round trip of the wrapper factors (geometry.cpp:92-101,143-148)
            auto fixed16 = [](float x) { return static_cast<uint16_t>(std::clamp(std::round(x * 65535.f), 0.f, 65535.f)); };
            auto has_normal = (vertex_props & LR_SHAPE_HAS_VERTEX_NORMAL) != 0u;
            auto shadow_term = fixed16(has_normal ? info.shadow_terminator : 0.f);
            auto isect_offset = fixed16(info.intersection_offset);
            lr_instance inst{};
            inst.handle = encode_instance_handle(mesh_index, properties, surface_tag, light_tag, medium_tag, mesh.triangle_count,
                                                 static_cast<float>(shadow_term) / 65535.f,
                                                 static_cast<float>(isect_offset) / 65535.f);
            store_matrix(inst.object_to_world, object_to_world);
            inst.visible = visible ? 1u : 0u;
            _out.instances.emplace_back(inst);
            if (properties & LR_SHAPE_HAS_LIGHT) { _out.light_instances.push_back({instance_id, light_tag}); }
        } else {
            auto pushed = !is_identity(local) || local_animated;
            if (pushed) {
                _transform_stack.emplace_back(_transform_stack.back() * local);
                _transform_chain.push_back({local_node, local_animated});
            }
            auto children = d->impl_type() == "group" ? d->node_list_required("shapes") :
                                                        NodeDesc::node_list{d->node("shape")};
            for (auto child : children) { process_shape(child, surface, light, visible, medium); }
            if (pushed) { _transform_stack.pop_back(), _transform_chain.pop_back(); }
        }
    }

    // ---------------- shutter (Camera::Camera camera.cpp:22-131, Camera::shutter_weight :150-161, Camera::shutter_samples :163-203)
    // Kept quirks: the custom curve is appended BEHIND `n` zero-initialised points (`resize(n)` followed by a
    // back_inserter, :109-114); bucket times are measured from 0, not from shutter_span.x (:172-176).  Conscious
    // correction: the reference draws the bucket jitter and the spp shuffle from an unseeded std::random_device (:171), so
    // two runs of it never agree; here the engine is seeded (19980810 + camera index) and the same samples reach the
    // oracle and the device through lrhost_scene_shutter_sample.
    void build_shutter(const NodeDesc *d, CameraRecord &rec, uint32_t camera_index) {
        auto spp = rec.camera.spp;
        float span[2];
        if (auto s2 = d->vector_opt("shutter_span", 2u)) { span[0] = static_cast<float>((*s2)[0]), span[1] = static_cast<float>((*s2)[1]); }
        else { span[0] = span[1] = d->float_or("shutter_span", 0.f); }
        if (span[1] < span[0]) { throw Error{"Invalid time span: [" + std::to_string(span[0]) + ", " + std::to_string(span[1]) + "]. [" + d->location() + "]"}; }
        rec.shutter_span[0] = span[0], rec.shutter_span[1] = span[1];
        if (span[0] == span[1]) {
            rec.shutter_samples = {ShutterSample{span[0], 1.f, spp}};
            return;
        }
        auto shutter_samples = d->uint_or("shutter_samples", 0u);
        if (shutter_samples == 0u) { shutter_samples = std::min(spp, 256u); }
        else if (shutter_samples > spp) {
            log_warning("Too many shutter samples (" + std::to_string(shutter_samples) + "), clamping to samples per pixel (" + std::to_string(spp) + "). [" + d->location() + "]");
            shutter_samples = spp;
        }
        struct Point { float time, weight; };
        std::vector<Point> points;
        auto time_points = d->float_list_or_empty("shutter_time_points");
        auto weights = d->float_list_or_empty("shutter_weights");
        if (time_points.size() != weights.size()) { throw Error{"Number of shutter time points and number of shutter weights mismatch. [" + d->location() + "]"}; }
        if (std::any_of(weights.begin(), weights.end(), [](auto w) { return w < 0.f; })) { throw Error{"Found negative shutter weight. [" + d->location() + "]"}; }
        if (time_points.empty()) {
            points = {{span[0], 1.f}, {span[1], 1.f}};
        } else {
            std::vector<uint32_t> indices(time_points.size());
            std::iota(indices.begin(), indices.end(), 0u);
            if (auto it = std::remove_if(indices.begin(), indices.end(), [&](auto i) { return time_points[i] < span[0] || time_points[i] > span[1]; }); it != indices.end()) {
                log_warning("Out-of-shutter samples (count = " + std::to_string(std::distance(it, indices.end())) + ") are to be removed. [" + d->location() + "]");
                indices.erase(it, indices.end());
            }
            std::sort(indices.begin(), indices.end(), [&](auto a, auto b) { return time_points[a] < time_points[b]; });
            if (auto it = std::unique(indices.begin(), indices.end(), [&](auto a, auto b) { return time_points[a] == time_points[b]; }); it != indices.end()) {
                log_warning("Duplicate shutter samples (count = " + std::to_string(std::distance(it, indices.end())) + ") are to be removed. [" + d->location() + "]");
                indices.erase(it, indices.end());
            }
            points.resize(indices.size());// (sic, camera.cpp:110)
            for (auto i : indices) { points.push_back({time_points[i], weights[i]}); }
            if (!points.empty()) {
                if (points.front().time > span[0]) { points.insert(points.begin(), Point{span[0], points.front().weight}); }
                if (points.back().time < span[1]) { points.push_back({span[1], points.back().weight}); }
            }
        }
        auto shutter_weight = [&](float time) {// camera.cpp:150-161
            if (time < span[0] || time > span[1]) { return 0.f; }
            auto ub = std::upper_bound(points.cbegin(), points.cend(), time, [](auto lhs, auto rhs) { return lhs < rhs.time; });
            auto u = std::distance(points.cbegin(), ub);
            if (u <= 0 || static_cast<size_t>(u) >= points.size()) { return u <= 0 ? points.front().weight : points.back().weight; }// (the reference reads out of range here)
            auto p0 = points[static_cast<size_t>(u - 1)], p1 = points[static_cast<size_t>(u)];
            auto t = (time - p0.time) / (p1.time - p0.time);
            return p0.weight + t * (p1.weight - p0.weight);
        };
        auto duration = span[1] - span[0];
        auto inv_n = 1.f / static_cast<float>(shutter_samples);
        std::uniform_real_distribution<float> dist{};
        std::default_random_engine random{19980810u + camera_index};
        std::vector<ShutterSample> buckets(shutter_samples);
        for (auto b = 0u; b < shutter_samples; b++) {
            auto ts = static_cast<float>(b) * inv_n * duration;
            auto te = static_cast<float>(b + 1u) * inv_n * duration;
            auto a = dist(random);
            auto t = ts + a * (te - ts);
            buckets[b].time = t, buckets[b].weight = shutter_weight(t);
        }
        std::vector<uint32_t> order(shutter_samples);
        std::iota(order.begin(), order.end(), 0u);
        std::shuffle(order.begin(), order.end(), random);
        auto remainder = spp % shutter_samples, per_bucket = spp / shutter_samples;
        for (auto i = 0u; i < remainder; i++) { buckets[order[i]].spp = per_bucket + 1u; }
        for (auto i = remainder; i < shutter_samples; i++) { buckets[order[i]].spp = per_bucket; }
        auto sum_weights = 0.0;
        for (auto &b : buckets) { sum_weights += static_cast<double>(b.weight * static_cast<float>(b.spp)); }
        if (sum_weights == 0.0) {
            log_warning("Invalid shutter samples generated. Falling back to uniform shutter curve.");
            for (auto &b : buckets) { b.weight = 1.f; }
        } else {
            auto scale = static_cast<double>(spp) / sum_weights;
            for (auto &b : buckets) { b.weight = static_cast<float>(static_cast<double>(b.weight) * scale); }
        }
        rec.shutter_samples = std::move(buckets);
    }

    // ---------------- camera, film, filter
    static float filter_evaluate(const NodeDesc *d, const std::string &impl, float radius, float x) {
        constexpr auto pi = 3.14159265358979323846f;
        if (impl == "box") { return 1.f; }
        if (impl == "triangle") { return std::max(1.f - std::abs(x / radius), 0.f); }
        if (impl == "gaussian") {// gaussian.cpp:16-40
            auto sigma = d->float_or("sigma", 0.f);
            if (sigma <= 0.f) { sigma = radius / 3.f; }
            auto s = 2.f * sigma * sigma;
            auto G = [s, pi](float v) { return 1.f / std::sqrt(pi * s) * std::exp(-v * v / s); };
            return G(x) - G(radius);
        }
        if (impl == "mitchell") {// mitchell.cpp:21-38
            auto b = d->float_or("b", 1.f / 3.f), c = d->float_or("c", 1.f / 3.f);
            x = 2.f * std::abs(x / radius);
            if (x <= 1.f) {
                return ((12.f - 9.f * b - 6.f * c) * x * x * x + (-18.f + 12.f * b + 6.f * c) * x * x + (6.f - 2.f * b)) * (1.f / 6.f);
            }
            if (x <= 2.f) {
                return ((-b - 6.f * c) * x * x * x + (6.f * b + 30.f * c) * x * x + (-12.f * b - 48.f * c) * x + (8.f * b + 24.f * c)) * (1.f / 6.f);
            }
            return 0.f;
        }
        if (impl == "lanczossinc") {// lanczos_sinc.cpp:18-28
            auto tau = d->float_or("tau", 3.f);
            x = x / radius;
            auto sin_x_over_x = [](float v) { return 1.f + v * v == 1.f ? 1.f : std::sin(v) / v; };
            auto sinc = [&](float v) { return sin_x_over_x(pi * v); };
            if (std::abs(x) > 1.f) { return 0.f; }
            return sinc(x) * sinc(x / tau);
        }
        throw Error{"Unknown filter implementation '" + impl + "'."};
    }

    lr_filter build_filter(const NodeDesc *d) {// Filter::Instance::Instance, filter.cpp:24-47
        _check_tag(d, Tag::FILTER);
        lr_filter f{};
        f.radius = std::max(d->float_or("radius", 0.5f), 1e-3f);
        auto shift2 = d->vector_opt("shift", 2u);
        auto shift1 = shift2 ? 0.f : d->float_or("shift", 0.f);
        f.shift[0] = shift2 ? static_cast<float>((*shift2)[0]) : shift1;
        f.shift[1] = shift2 ? static_cast<float>((*shift2)[1]) : shift1;
        constexpr auto n = LR_FILTER_LUT_SIZE - 1;
        constexpr auto inv_n = 1.0f / static_cast<float>(n);
        std::array<float, n> abs_f{};
        auto &impl = d->impl_type();
        f.lut[0] = filter_evaluate(d, impl, f.radius, -f.radius);
        auto integral = 0.f;
        for (auto i = 0; i < n; i++) {
            auto x = static_cast<float>(i + 1) * inv_n * 2.f - 1.f;
            f.lut[i + 1] = filter_evaluate(d, impl, f.radius, x * f.radius);
            auto f_mid = 0.5f * (f.lut[i] + f.lut[i + 1]);
            integral += f_mid;
            abs_f[static_cast<size_t>(i)] = std::abs(f_mid);
        }
        auto inv_integral = 1.f / integral;
        for (auto &v : f.lut) { v *= inv_integral; }
        std::vector<lr_alias_entry> table;
        std::vector<float> pdf;
        create_alias_table(abs_f.data(), abs_f.size(), table, pdf);
        for (auto i = 0; i < n; i++) {
            f.pdf[i] = pdf[static_cast<size_t>(i)];
            f.alias_prob[i] = table[static_cast<size_t>(i)].prob;
            f.alias_index[i] = table[static_cast<size_t>(i)].alias;
        }
        return f;
    }

    CameraRecord build_camera(const NodeDesc *d) {
        _check_tag(d, Tag::CAMERA);
        CameraRecord rec{};
        auto &cam = rec.camera;
        // film (color.cpp:25-42)
        auto film = d->node("film");
        _check_tag(film, Tag::FILM);
        if (film->impl_type() != "color") {
            throw Error{"Film '" + film->impl_type() + "' is out of scope (only Color). [" + film->location() + "]"};
        }
        auto res2 = film->vector_opt("resolution", 2u);
        auto res1 = res2 ? 0u : film->uint_or("resolution", 1024u);
        cam.width = res2 ? static_cast<uint32_t>((*res2)[0]) : res1;
        cam.height = res2 ? static_cast<uint32_t>((*res2)[1]) : res1;
        auto exp3 = film->vector_opt("exposure", 3u);
        auto exp1 = exp3 ? 0.f : film->float_or("exposure", 0.f);
        for (auto i = 0; i < 3; i++) {
            rec.film.scale[i] = std::pow(2.f, exp3 ? static_cast<float>((*exp3)[static_cast<size_t>(i)]) : exp1);
        }
        rec.film.clamp = std::max(1.f, film->float_or("clamp", 256.f));
        // filter
        auto filter = d->node_or_null("filter");
        rec.filter = build_filter(filter ? filter : NodeDesc::shared_default(Tag::FILTER, "Box"));
        // transform (camera.cpp:16-50)
        auto xform = d->node_or_null("transform");
        auto c2w = float4x4::identity();
        if (xform != nullptr) {
            c2w = transform_matrix(xform);
            if (transform_is_animated(xform)) { rec.xform = static_cast<int32_t>(compile_transform(xform)); }
        } else {
            auto position = float3_or(d, "position", {0.f, 0.f, 0.f});
            auto front_opt = d->vector_opt("front", 3u);
            float3 front;
            if (front_opt) {
                front = to_float3(*front_opt);
            } else {
                auto look_at = float3_or(d, "look_at", position + float3{0.f, 0.f, -1.f});
                front = normalize(look_at - position);
            }
            auto up = float3_or(d, "up", {0.f, 1.f, 0.f});
            if (!(position == float3{0.f, 0.f, 0.f} && front == float3{0.f, 0.f, -1.f} && up == float3{0.f, 1.f, 0.f})) {
                auto w = normalize(-front);
                auto u = normalize(cross(up, w));
                auto v = normalize(cross(w, u));
                c2w = {{make_float4(u, 0.f), make_float4(v, 0.f), make_float4(w, 0.f), make_float4(position, 1.f)}};
            }
        }
        store_matrix(cam.camera_to_world, c2w);
        cam.spp = d->uint_or("spp", 1024u);
        build_shutter(d, rec, static_cast<uint32_t>(_out.cameras.size()));
        // clip planes (camera.h:116-157)
        auto clip2 = d->vector_opt("clip", 2u);
        if (!clip2) { clip2 = d->vector_opt("clip_plane", 2u); }
        if (clip2) {
            cam.clip_near = static_cast<float>((*clip2)[0]), cam.clip_far = static_cast<float>((*clip2)[1]);
        } else {
            auto near_plane = d->number_opt("clip");
            if (!near_plane) { near_plane = d->number_opt("clip_plane"); }
            cam.clip_near = near_plane ? static_cast<float>(*near_plane) : 0.f;
            cam.clip_far = 1e10f;
        }
        cam.clip_near = std::clamp(cam.clip_near, 0.f, 1e10f);
        cam.clip_far = std::clamp(cam.clip_far, 0.f, 1e10f);
        if (cam.clip_near > cam.clip_far) { std::swap(cam.clip_near, cam.clip_far); }
        auto &impl = d->impl_type();
        if (impl == "pinhole") {// pinhole.cpp:35-46
            cam.kind = LR_CAMERA_PINHOLE;
            auto fov = radians(std::clamp(d->float_or("fov", 35.f), 1e-3f, 180.f - 1e-3f));
            cam.tan_half_fov = std::tan(fov * 0.5f);
        } else if (impl == "thinlens") {// thin_lens.cpp:23-34,71-87
            cam.kind = LR_CAMERA_THIN_LENS;
            auto aperture = d->float_or("aperture", 2.f);
            auto focal_length = d->float_or("focal_length", 35.f);
            float focus_distance;
            if (auto fd = d->number_opt("focus_distance")) {
                focus_distance = static_cast<float>(*fd);
            } else {
                auto target = d->vector_opt("look_at", 3u), position = d->vector_opt("position", 3u);
                if (!target || !position) { throw Error{"ThinLens camera needs focus_distance or look_at/position. [" + d->location() + "]"}; }
                focus_distance = length(to_float3(*target) - to_float3(*position));
            }
            focus_distance = std::max(std::abs(focus_distance), 1e-4f);
            auto v = static_cast<double>(focus_distance);
            auto f = static_cast<double>(focal_length) * 1e-3;
            auto u = 1. / (1. / f - 1. / v);
            auto ratio = static_cast<float>(v / u);
            cam.focus_distance = focus_distance;
            cam.lens_radius = static_cast<float>(.5 * f / static_cast<double>(aperture));
            auto rx = static_cast<float>(cam.width), ry = static_cast<float>(cam.height);
            cam.projected_pixel_size = rx > ry ?
                std::min(static_cast<float>(ratio * .036 / rx), static_cast<float>(ratio * .024 / ry)) :
                std::min(static_cast<float>(ratio * .024 / rx), static_cast<float>(ratio * .036 / ry));
        } else if (impl == "ortho") {// ortho.cpp
            cam.kind = LR_CAMERA_ORTHO;
            cam.ortho_scale = std::pow(2.f, d->float_or("zoom", 0.f));
        } else {
            throw Error{"Unknown camera implementation '" + impl + "'. [" + d->location() + "]"};
        }
        // output file (camera.cpp:138-147)
        auto src = d->source_file();
        auto default_dir = src.empty() ? fs::current_path() : fs::path{src}.parent_path();
        rec.file = d->path_or("file", (default_dir / "render.exr").string());
        return rec;
    }

    void build_environment(const NodeDesc *d);
    lr_environment build_environment_node(const NodeDesc *d, std::vector<lr_alias_entry> &alias, std::vector<float> &pdf, bool is_root, uint32_t depth);
    lr_environment build_combined(const NodeDesc *d, std::vector<lr_alias_entry> &alias, std::vector<float> &pdf, uint32_t depth);

    // tables of src/util/sobolmatrices.cpp, re-derived by tools/gen_sobol_tables.py into data/sobol_tables.bin
    void load_sobol_tables() {
        Dl_info info{};
        std::string dir = ".";
        if (dladdr(reinterpret_cast<const void *>(&create_alias_table), &info) != 0 && info.dli_fname != nullptr) {
            dir = fs::path{info.dli_fname}.parent_path().string();
        }
        std::string path;
        if (auto env = std::getenv("LR_DATA_DIR")) { path = (fs::path{env} / "sobol_tables.bin").string(); }
        else { path = (fs::path{dir} / ".." / "data" / "sobol_tables.bin").string(); }
        std::ifstream f{path, std::ios::binary};
        if (!f) { throw Error{"Sobol sampler tables not found at '" + path + "' (run tools/gen_sobol_tables.py)."}; }
        char magic[4];
        uint32_t dims = 0, cols = 0, nvdc = 0, ninv = 0;
        f.read(magic, 4);
        f.read(reinterpret_cast<char *>(&dims), 4), f.read(reinterpret_cast<char *>(&cols), 4);
        f.read(reinterpret_cast<char *>(&nvdc), 4), f.read(reinterpret_cast<char *>(&ninv), 4);
        if (std::memcmp(magic, "LRSB", 4) != 0 || dims != LR_SOBOL_DIMENSIONS || cols != LR_SOBOL_MATRIX_SIZE || nvdc != 25u || ninv != 26u) {
            throw Error{"Invalid Sobol table file '" + path + "'."};
        }
        _out.sobol_matrices.resize(static_cast<size_t>(dims) * cols);
        _out.vdc_sobol.resize(static_cast<size_t>(nvdc) * cols);
        _out.vdc_sobol_inv.resize(static_cast<size_t>(ninv) * cols);
        f.read(reinterpret_cast<char *>(_out.sobol_matrices.data()), static_cast<std::streamsize>(_out.sobol_matrices.size() * 4u));
        f.read(reinterpret_cast<char *>(_out.vdc_sobol.data()), static_cast<std::streamsize>(_out.vdc_sobol.size() * 8u));
        f.read(reinterpret_cast<char *>(_out.vdc_sobol_inv.data()), static_cast<std::streamsize>(_out.vdc_sobol_inv.size() * 8u));
        if (!f) { throw Error{"Truncated Sobol table file '" + path + "'."}; }
    }

    void build() {
        auto root = _desc.root();
        if (!root->is_defined()) { throw Error{"Root node is not defined in the scene description."}; }
        for (auto i = 0; i < 3; i++) {
            _out.world_min[i] = std::numeric_limits<float>::max();
            _out.world_max[i] = -std::numeric_limits<float>::max();
        }
        _scene_shadow_terminator = root->float_or("shadow_terminator", 0.f);
        _scene_intersection_offset = root->float_or("intersection_offset", 0.f);
        if (auto spectrum = root->node_or_null("spectrum"); spectrum != nullptr && spectrum->impl_type() != "srgb") {
            throw Error{"Spectrum '" + spectrum->impl_type() + "' cannot be reproduced (srgb2spec table blob is missing "
                        "from the reference snapshot, SURVEY §0); use sRGB."};
        }
        // integrator (integrator.cpp:13-18, mega_path.cpp:21-25)
        auto integrator = root->node("integrator");
        _check_tag(integrator, Tag::INTEGRATOR);
        _out.integrator_impl = integrator->impl_type();
        if (integrator->impl_type() == "megapath") {
            _out.integrator.kind = LR_INTEGRATOR_MEGAPATH;
            _out.integrator.max_depth = std::max(integrator->uint_or("depth", 10u), 1u);
            _out.integrator.rr_depth = integrator->uint_or("rr_depth", 0u);
            _out.integrator.rr_threshold = std::max(integrator->float_or("rr_threshold", 0.95f), 0.05f);
        } else if (integrator->impl_type() == "direct") {// DirectLighting, direct.cpp:27-42
            _out.integrator.kind = LR_INTEGRATOR_DIRECT;
            auto is = integrator->string_or("importance_sampling", "both");
            for (auto &c : is) { c = static_cast<char>(std::tolower(c)); }
            if (is == "light") { _out.integrator.flags = LR_DIRECT_SAMPLE_LIGHTS; }
            else if (is == "material" || is == "surface" || is == "bsdf") { _out.integrator.flags = LR_DIRECT_SAMPLE_SURFACES; }
            else {
                if (is != "both" && is != "mis" && is != "multiple") {
                    log_warning("Unknown importance sampling method \"" + is + "\". Using \"both\" instead.");
                }
                _out.integrator.flags = LR_DIRECT_SAMPLE_LIGHTS | LR_DIRECT_SAMPLE_SURFACES;
            }
            _out.integrator.max_depth = 2u;// camera vertex + the BSDF-sampled one
            _out.integrator.rr_depth = ~0u, _out.integrator.rr_threshold = 0.95f;
        } else if (integrator->impl_type() == "normal") {// NormalVisualizer, normal.cpp:19-22
            _out.integrator.kind = LR_INTEGRATOR_NORMAL;
            _out.integrator.flags = (integrator->bool_or("remap", true) ? uint32_t{LR_NORMAL_REMAP} : 0u) |
                                    (integrator->bool_or("shading", true) ? uint32_t{LR_NORMAL_SHADING} : 0u);
            _out.integrator.max_depth = 1u;
            _out.integrator.rr_depth = ~0u, _out.integrator.rr_threshold = 0.95f;
        } else if (integrator->impl_type() == "megavptnaive") {// MegakernelVolumePathTracingNaive, mega_vpt_naive.cpp:29-32
            _out.integrator.kind = LR_INTEGRATOR_VPT_NAIVE;
            _out.integrator.max_depth = std::max(integrator->uint_or("depth", 20u), 1u);
            _out.integrator.rr_depth = integrator->uint_or("rr_depth", 0u);
            _out.integrator.rr_threshold = std::max(integrator->float_or("rr_threshold", 0.95f), 0.05f);
        } else {
            throw Error{"Integrator '" + integrator->impl_type() + "' is out of scope: this framework implements the MegaPath hot "
                        "path and its sibling megakernels Direct / Normal / MegaVPTNaive (SURVEY §2 row 21, §8 f3-f4)."};
        }
        _out.integrator.environment_medium_tag = LR_INVALID_ID;
        auto sampler = integrator->node_or_null("sampler");
        if (sampler == nullptr) { sampler = NodeDesc::shared_default(Tag::SAMPLER, "independent"); }
        _check_tag(sampler, Tag::SAMPLER);
        if (sampler->impl_type() == "tileshared") {// src/samplers/tile_shared.cpp:21-29: a wrapper around `base`
            std::vector<uint32_t> size{16u};// property_uint2_or_default("tile_size", uint or_default 16)
            if (!sampler->float_list_or_empty("tile_size").empty()) { size = sampler->uint_list("tile_size"); }
            if (size.empty() || size.size() > 2u || size[0] == 0u || size.back() == 0u) { throw Error{"Invalid tile_size of the TileShared sampler. [" + sampler->location() + "]"}; }
            _out.sampler.tile_size[0] = size[0], _out.sampler.tile_size[1] = size.back();
            _out.sampler.tile_jitter = sampler->bool_or("jitter", false) ? 1u : 0u;
            sampler = sampler->node("base");
            _check_tag(sampler, Tag::SAMPLER);
            if (sampler->impl_type() == "tileshared") { throw Error{"A TileShared sampler inside a TileShared sampler is not supported. [" + sampler->location() + "]"}; }
        }
        _out.sampler.seed = sampler->uint_or("seed", 19980810u);
        if (sampler->impl_type() == "independent") { _out.sampler.kind = LR_SAMPLER_INDEPENDENT; }
        else if (sampler->impl_type() == "sobol") { _out.sampler.kind = LR_SAMPLER_SOBOL; }
        else if (sampler->impl_type() == "paddedsobol") { _out.sampler.kind = LR_SAMPLER_PADDED_SOBOL; }
        else if (sampler->impl_type() == "pcg32") { _out.sampler.kind = LR_SAMPLER_PCG32; }
        else {
            throw Error{"Sampler '" + sampler->impl_type() + "' is not reproducible here (PMJ02BN tables missing, ZSobol "
                        "hash unpinned; SURVEY §2 row 8)."};
        }
        if (_out.sampler.kind == LR_SAMPLER_SOBOL || _out.sampler.kind == LR_SAMPLER_PADDED_SOBOL) { load_sobol_tables(); }
        auto light_sampler = integrator->node_or_null("light_sampler");
        auto env_weight = 0.5f;
        if (light_sampler != nullptr) {
            _check_tag(light_sampler, Tag::LIGHT_SAMPLER);
            if (light_sampler->impl_type() != "uniform") { throw Error{"Unknown light sampler '" + light_sampler->impl_type() + "'."}; }
            env_weight = light_sampler->float_or("environment_weight", 0.5f);
        }
        // environment (scene.cpp:216-217)
        build_environment(root->node_or_null("environment"));
        // cameras and shapes
        for (auto c : root->node_list_required("cameras")) { _out.cameras.emplace_back(build_camera(c)); }
        _transform_stack.assign(1u, float4x4::identity());
        for (auto s : root->node_list_required("shapes")) { process_shape(s, nullptr, nullptr, true); }
        if (_out.instances.empty()) { throw Error{"No shapes in the scene."}; }
        log_info("Geometry built with " + std::to_string([&] {
                     uint64_t n = 0;
                     for (auto &i : _out.instances) { n += i.handle.z; }
                     return n;
                 }()) + " triangles.");
        // Disney closures of one kind (disney / disney_trans / disney_thin) share ONE polymorphic closure object in
        // the reference, whose lobe mask is the union over all materials of that kind (enable_lobes, disney.cpp:856)
        uint32_t lobe_union[3] = {0u, 0u, 0u};
        auto disney_class = [](const lr_surface &s) { return (s.flags & LR_SURFACE_FLAG_THIN) ? 2u : (s.u[1] ? 1u : 0u); };
        for (auto &s : _out.surfaces) {
            if (s.kind == LR_SURFACE_DISNEY) { lobe_union[disney_class(s)] |= s.u[2]; }
        }
        for (auto &s : _out.surfaces) {
            if (s.kind == LR_SURFACE_DISNEY) { s.u[0] = lobe_union[disney_class(s)]; }
        }
        // the environment medium is registered after the geometry (pipeline.cpp:72-79): its tag follows the shapes' media
        if (auto env_medium = root->node_or_null("environment_medium"); env_medium != nullptr && env_medium->impl_type() != "null") {
            _out.integrator.environment_medium_tag = register_medium(env_medium);
        }
        // UniformLightSamplerInstance (uniform.cpp:31-48)
        _out.integrator.light_count = static_cast<uint32_t>(_out.lights.size());
        if (_out.environment.kind != LR_ENV_NONE) {
            _out.integrator.env_prob = _out.lights.empty() ? 1.f : std::clamp(env_weight, 0.01f, 0.99f);
        } else {
            _out.integrator.env_prob = 0.f;
        }
    }
};

// one Spherical / Directional record (kind LR_ENV_NONE when null or black)
// (the record of a Combined node is what CombinedInstance computes over its children: build_combined below)
lr_environment Builder::build_environment_node(const NodeDesc *d, std::vector<lr_alias_entry> &alias, std::vector<float> &pdf, bool is_root, uint32_t depth) {
    lr_environment env{};
    env.emission_tex = -1;
    alias.clear(), pdf.clear();
    if (d == nullptr || d->impl_type() == "null") { return env; }
    _check_tag(d, Tag::ENVIRONMENT);
    if (d->impl_type() == "combined") { return build_combined(d, alias, pdf, depth); }
    auto m = transform_matrix(d->node_or_null("transform"));
    if (transform_is_animated(d->node_or_null("transform"))) {
        if (!is_root) { throw Error{"Animated transforms on the children of a Combined environment are not supported. [" + d->location() + "]"}; }
        _out.environment_xform = static_cast<int32_t>(compile_transform(d->node_or_null("transform")));
    }
    // Environment::Instance::transform_to_world: 3x3 of the env transform (environment.cpp:17-19)
    for (auto c = 0; c < 3; c++) {
        for (auto r = 0; r < 3; r++) {
            env.env_to_world[c * 3 + r] = m[c][r];
            env.world_to_env[c * 3 + r] = m[r][c];// transpose (rotation)
        }
    }
    if (d->impl_type() == "spherical") {// spherical.cpp:17-40
        env.kind = LR_ENV_SPHERICAL;
        env.emission_tex = load_texture(d->node("emission"));
        env.scale = std::max(d->float_or("scale", 1.f), 0.f);
        env.compensate_mis = d->bool_or("compensate_mis", true) ? 1u : 0u;
        if (env.scale == 0.f || texture_is_black(env.emission_tex)) { env.kind = LR_ENV_NONE; }
        if (env.kind != LR_ENV_NONE) { build_environment_tables(_out, env, alias, pdf); }// no-op for constant emission
    } else if (d->impl_type() == "directional") {// directional.cpp:27-47
        env.kind = LR_ENV_DIRECTIONAL;
        env.emission_tex = load_texture(d->node("emission"));
        if (_out.textures[static_cast<size_t>(env.emission_tex)].kind != LR_TEX_CONSTANT) {
            log_warning("Directional environment emission is not constant. This may lead to unexpected results. [" + d->location() + "]");
        }
        auto scale = std::max(d->float_or("scale", 1.f), 0.f);
        env.visible = d->bool_or("visible", true) ? 1u : 0u;
        auto angle = std::clamp(d->float_or("angle", 1.f), 1e-3f, 360.f);
        auto cos_half_angle = std::cos(.5 * angle * 3.14159265358979323846 / 180.0);
        env.cos_half_angle = static_cast<float>(cos_half_angle);
        if (d->bool_or("normalize", true)) { scale = static_cast<float>(2. * scale / (1. - cos_half_angle)); }
        env.scale = scale;
        auto dv = d->vector_opt("direction", 3u);
        auto dir = dv ? normalize(float3{static_cast<float>((*dv)[0]), static_cast<float>((*dv)[1]), static_cast<float>((*dv)[2])}) : float3{0.f, 1.f, 0.f};
        env.direction[0] = dir.x, env.direction[1] = dir.y, env.direction[2] = dir.z;
        if (env.scale == 0.f || texture_is_black(env.emission_tex)) { env.kind = LR_ENV_NONE; }
    } else {
        throw Error{"Unknown environment implementation '" + d->impl_type() + "'. [" + d->location() + "]"};
    }
    return env;
}

// Combined (combined.cpp:17-124): the record of the node, its children appended to _out.env_children (children before parents, so
// that the indices of a record's children are smaller than its own).  `depth`: Combined nodes above this one.
lr_environment Builder::build_combined(const NodeDesc *d, std::vector<lr_alias_entry> &alias, std::vector<float> &pdf, uint32_t depth) {
    if (depth >= static_cast<uint32_t>(LR_ENV_MAX_COMBINED_DEPTH)) {
        throw Error{"Combined environments nested more than " + std::to_string(LR_ENV_MAX_COMBINED_DEPTH) + " deep are not supported. [" + d->location() + "]"};
    }
    // combined.cpp:23-36: a null or black child gets scale 0
    lr_environment child[2];
    std::vector<lr_alias_entry> child_alias[2];
    std::vector<float> child_pdf[2];
    child[0] = build_environment_node(d->node_or_null("a"), child_alias[0], child_pdf[0], false, depth + 1u);
    child[1] = build_environment_node(d->node_or_null("b"), child_alias[1], child_pdf[1], false, depth + 1u);
    float scales[2] = {std::max(d->float_or("scale_a", 1.f), 0.f), std::max(d->float_or("scale_b", 1.f), 0.f)};
    for (auto i = 0; i < 2; i++) {
        if (child[i].kind == LR_ENV_NONE) { scales[i] = 0.f; }
    }
    if (transform_is_animated(d->node_or_null("transform"))) { throw Error{"Animated transforms on a Combined environment are not supported. [" + d->location() + "]"}; }
    auto m = transform_matrix(d->node_or_null("transform"));
    float c2w[9], w2c[9];
    for (auto c = 0; c < 3; c++) {
        for (auto r = 0; r < 3; r++) { c2w[c * 3 + r] = m[c][r], w2c[c * 3 + r] = m[r][c]; }
    }
    lr_environment env{};
    env.emission_tex = -1;
    alias.clear(), pdf.clear();
    if (scales[0] == 0.f && scales[1] == 0.f) { return env; }// is_black, :36
    if (scales[0] == 0.f || scales[1] == 0.f) {
        // only one live child (combined.cpp:72-77,103-109): that child seen through the Combined node's transform with
        // its radiance scaled -- the same function as the child record with composed matrices and scale (a live child that
        // is a Combined node: its children's radiances scale linearly, so its scales take the factor)
        auto live = scales[0] == 0.f ? 1 : 0;
        env = child[live];
        if (env.kind == LR_ENV_COMBINED) { env.child_scale[0] *= scales[live], env.child_scale[1] *= scales[live]; }
        else { env.scale *= scales[live]; }
        auto mul = [](const float *a, const float *b, float *out) {// out = a * b, column-major 3x3
            for (auto c = 0; c < 3; c++) {
                for (auto r = 0; r < 3; r++) { out[c * 3 + r] = a[0 * 3 + r] * b[c * 3 + 0] + a[1 * 3 + r] * b[c * 3 + 1] + a[2 * 3 + r] * b[c * 3 + 2]; }
            }
        };
        float w[9], e[9];
        mul(child[live].world_to_env, w2c, w);// wi_local = W_child (W_comb wi)
        mul(c2w, child[live].env_to_world, e);// wi_world = E_comb (E_child w)
        std::copy_n(w, 9, env.world_to_env), std::copy_n(e, 9, env.env_to_world);
        alias = std::move(child_alias[live]), pdf = std::move(child_pdf[live]);
        return env;
    }
    env.kind = LR_ENV_COMBINED;
    std::copy_n(c2w, 9, env.env_to_world), std::copy_n(w2c, 9, env.world_to_env);
    for (auto i = 0; i < 2; i++) {
        env.child_scale[i] = scales[i];
        env.child[i] = static_cast<uint32_t>(_out.env_children.size());
        _out.env_children.push_back(child[i]);
        _out.env_child_alias.push_back(std::move(child_alias[i])), _out.env_child_pdf.push_back(std::move(child_pdf[i]));
    }
    return env;
}

void Builder::build_environment(const NodeDesc *d) {
    _out.env_children.clear(), _out.env_child_alias.clear(), _out.env_child_pdf.clear();
    _out.environment = build_environment_node(d, _out.env_alias, _out.env_pdf, true, 0u);
}

}// namespace

// ---------------- animation: Transform::matrix(time), Pipeline::update, Geometry::update
namespace {

// util/xform.cpp:12-117, restated on a plain column-major 3x3
struct M3 { float3 c[3]; };
M3 m3_of(const float4x4 &m) { return {{{m[0].x, m[0].y, m[0].z}, {m[1].x, m[1].y, m[1].z}, {m[2].x, m[2].y, m[2].z}}}; }
M3 m3_transpose(const M3 &m) { return {{{m.c[0].x, m.c[1].x, m.c[2].x}, {m.c[0].y, m.c[1].y, m.c[2].y}, {m.c[0].z, m.c[1].z, m.c[2].z}}}; }
M3 m3_mul(const M3 &a, const M3 &b) {
    M3 r;
    for (auto j = 0; j < 3; j++) { r.c[j] = a.c[0] * b.c[j].x + a.c[1] * b.c[j].y + a.c[2] * b.c[j].z; }
    return r;
}
M3 m3_inverse(const M3 &m) {// adjugate / determinant (luisa::inverse(float3x3), core/mathematics.h; absent from the snapshot)
    auto a = m.c[0], b = m.c[1], c = m.c[2];
    auto r0 = cross(b, c), r1 = cross(c, a), r2 = cross(a, b);
    auto inv_det = 1.f / dot(a, r0);
    return m3_transpose({{r0 * inv_det, r1 * inv_det, r2 * inv_det}});
}

struct Quat { float3 v; float w; };
struct Decomposed { float3 scaling; Quat q; float3 translation; };

Quat quaternion_of(const M3 &m) {// xform.cpp:45-74
    auto e = [&](int i, int j) { return m.c[i][j]; };
    if (auto trace = e(0, 0) + e(1, 1) + e(2, 2); trace > 0.f) {
        auto s = std::sqrt(trace + 1.f);
        auto w = 0.5f * s;
        s = 0.5f / s;
        return {float3{e(1, 2) - e(2, 1), e(2, 0) - e(0, 2), e(0, 1) - e(1, 0)} * s, w};
    }
    const int next[3] = {1, 2, 0};
    float3 v{};
    auto i = 0;
    if (e(1, 1) > e(0, 0)) { i = 1; }
    if (e(2, 2) > e(i, i)) { i = 2; }
    auto j = next[i], k = next[j];
    auto s = std::sqrt(std::max(e(i, i) - (e(j, j) + e(k, k)) + 1.f, 0.f));
    v[i] = s * 0.5f;
    if (s != 0.f) { s = 0.5f / s; }
    auto w = (e(j, k) - e(k, j)) * s;
    v[j] = (e(i, j) + e(j, i)) * s;
    v[k] = (e(i, k) + e(k, i)) * s;
    return {v, w};
}

Decomposed decompose(const float4x4 &m) {// xform.cpp:12-43 (polar decomposition by averaging R with its inverse transpose)
    float3 t{m[3].x, m[3].y, m[3].z};
    auto N = m3_of(m);
    auto R = N;
    for (auto it = 0; it < 100; it++) {
        auto R_it = m3_inverse(m3_transpose(R));
        M3 R_next, diff;
        for (auto c = 0; c < 3; c++) {
            R_next.c[c] = (R.c[c] + R_it.c[c]) * 0.5f;
            diff.c[c] = R.c[c] - R_next.c[c];
        }
        R = R_next;
        float3 n{std::abs(diff.c[0].x) + std::abs(diff.c[1].x) + std::abs(diff.c[2].x),
                 std::abs(diff.c[0].y) + std::abs(diff.c[1].y) + std::abs(diff.c[2].y),
                 std::abs(diff.c[0].z) + std::abs(diff.c[1].z) + std::abs(diff.c[2].z)};
        if (std::max({n.x, n.y, n.z}) <= 1e-4f) { break; }
    }
    auto S = m3_mul(m3_inverse(R), N);
    auto near_zero = [](float f) { return std::abs(f) <= 1e-4f; };
    if (!near_zero(S.c[0].y) || !near_zero(S.c[0].z) || !near_zero(S.c[1].x) || !near_zero(S.c[1].z) || !near_zero(S.c[2].x) || !near_zero(S.c[2].y)) {
        log_warning("Non-zero entries found in decomposed scaling matrix.");
    }
    return {{S.c[0].x, S.c[1].y, S.c[2].z}, quaternion_of(R), t};
}

float q_dot(Quat a, Quat b) { return dot(a.v, b.v) + a.w * b.w; }
float q_length(Quat a) { return std::sqrt(q_dot(a, a)); }
Quat q_axpby(Quat a, float x, Quat b, float y) { return {a.v * x + b.v * y, a.w * x + b.w * y}; }

Quat slerp(Quat q1, Quat q2, float t) {// xform.cpp:90-99
    auto safe_asin = [](float x) { return std::asin(std::clamp(x, -1.f, 1.f)); };
    auto sin_x_over_x = [](float x) { return 1.f + x * x == 1.f ? 1.f : std::sin(x) / x; };
    constexpr auto pi = 3.14159265358979323846f;
    auto theta = q_dot(q1, q2) < 0.f ? pi - 2.f * safe_asin(q_length(q_axpby(q1, 1.f, q2, 1.f)) * 0.5f) :
                                       2.f * safe_asin(q_length(q_axpby(q1, 1.f, q2, -1.f)) * 0.5f);
    auto sto = sin_x_over_x(theta);
    auto q = q_axpby(q1, (1.f - t) * sin_x_over_x((1.f - t) * theta) / sto, q2, t * sin_x_over_x(t * theta) / sto);
    auto l = q_length(q);
    return {q.v * (1.f / l), q.w / l};
}

float4x4 rotation_of(Quat q) {// xform.cpp:76-79
    auto l = std::sqrt(dot(q.v, q.v));
    // conscious correction: the reference normalises a zero axis here (a key pair without rotation, e.g. a pure translation,
    // gives q = (0, 0, 0, 1)) and turns the whole matrix into NaNs; a zero rotation is the identity
    if (l == 0.f) { return float4x4::identity(); }
    auto theta = 2.f * std::atan2(l, q.w);
    return rotation(q.v * (1.f / l), theta);
}

}// namespace

float4x4 evaluate_xform(const SceneData &scene, uint32_t id, float time) {
    auto &n = scene.xforms[id];
    if (n.kind == XformNode::STATIC) { return n.m; }
    if (n.kind == XformNode::STACK) {// stack.cpp:38-47
        auto m = float4x4::identity();
        for (auto c : n.children) { m = evaluate_xform(scene, c, time) * m; }
        return m;
    }
    // lerp.cpp:71-109
    if (time <= n.times.front()) { return evaluate_xform(scene, n.children.front(), n.times.front()); }
    if (time >= n.times.back()) { return evaluate_xform(scene, n.children.back(), n.times.back()); }
    auto upper = static_cast<size_t>(std::upper_bound(n.times.begin(), n.times.end(), time) - n.times.begin());
    // (the reference caches the two decompositions per interval at the time of the interval's first query; key transforms
    // that are themselves animated would make its result depend on the query order — here they are evaluated at `time`)
    auto t0 = decompose(evaluate_xform(scene, n.children[upper - 1u], time));
    auto t1 = decompose(evaluate_xform(scene, n.children[upper], time));
    auto t = (time - n.times[upper - 1u]) / (n.times[upper] - n.times[upper - 1u]);
    auto S = t0.scaling + (t1.scaling - t0.scaling) * t;
    auto R = slerp(t0.q, t1.q, t);
    auto T = t0.translation + (t1.translation - t0.translation) * t;
    return translation(T) * rotation_of(R) * scaling(S);
}

bool set_scene_time(SceneData &scene, float time) {
    scene.time = time;
    auto moved = false;
    for (auto &dyn : scene.dynamic_instances) {
        auto m = float4x4::identity();
        for (auto id : dyn.chain) { m = m * evaluate_xform(scene, id, time); }
        store_matrix(scene.instances[dyn.instance].object_to_world, m);
        moved = true;
    }
    for (auto &cam : scene.cameras) {
        if (cam.xform >= 0) {
            store_matrix(cam.camera.camera_to_world, evaluate_xform(scene, static_cast<uint32_t>(cam.xform), time));
            moved = true;
        }
    }
    if (scene.environment_xform >= 0) {
        auto m = evaluate_xform(scene, static_cast<uint32_t>(scene.environment_xform), time);
        for (auto c = 0; c < 3; c++) {
            for (auto r = 0; r < 3; r++) {
                scene.environment.env_to_world[c * 3 + r] = m[c][r];
                scene.environment.world_to_env[c * 3 + r] = m[r][c];
            }
        }
        moved = true;
    }
    if (!scene.dynamic_instances.empty() && !scene.bvh_nodes.empty()) { refit_accel(scene); }
    return moved;
}

std::unique_ptr<SceneData> build_scene(const SceneDesc &desc) {
    auto out = std::make_unique<SceneData>();
    Builder{desc, *out}.build();
    // Pipeline::create: the tables are built at the earliest shutter opening of any camera (pipeline.cpp:50-56,72)
    auto initial_time = std::numeric_limits<float>::max();
    for (auto &c : out->cameras) { initial_time = std::min(initial_time, c.shutter_span[0]); }
    set_scene_time(*out, out->cameras.empty() ? 0.f : initial_time);
    return out;
}

}// namespace lr
