// oracle_bvh.h — the oracle's canonical acceleration structure and ray casts.
// TEST INFRASTRUCTURE ONLY (see oracle_math.h).
//
// The reference leaves BVH build/traversal to LuisaCompute's Accel (Embree on its CPU
// backends; absent submodule — call sites src/base/geometry.cpp:16,26,66,130,221,250,265).
// The published interface is restated: a two-level structure (TLAS over instances with a 4x4
// object->world each, BLAS per unique mesh), closest-hit returning {inst, prim, bary(u, v)} with
// weights (1-u-v, u, v) on (v0, v1, v2) (src/base/geometry.h:16-28), and an any-hit query.
// Rays are transformed to object space un-normalised, so t is shared between spaces.
//
// Canonical form for the "algorithmic bytes" of SURVEY §8(d): binned-SAH BVH2 (16 bins, leaf
// <= 4 triangles); one node = two child boxes + two references = 64 B; one triangle test =
// 48 B.  Counters are gathered per ray.
#pragma once
#include <array>
#include <numeric>
#include <functional>
#include <vector>

#include "../include/lr_scene.h"
#include "oracle_math.h"

namespace oracle {

struct Ray {
    float3 o;
    float t_min;
    float3 d;
    float t_max;
};

struct Hit {
    uint32_t inst{LR_INVALID_ID};
    uint32_t prim{LR_INVALID_ID};
    float2 bary;
    float t{0.f};
    bool miss() const { return inst == LR_INVALID_ID; }
};

struct TraceCounters {
    uint64_t nodes{0}, tris{0};
};

struct Aabb {
    float3 lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float3 hi{-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
    void grow(float3 p) {
        lo = f3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
        hi = f3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
    }
    void grow(const Aabb &b) { grow(b.lo), grow(b.hi); }
    float half_area() const {
        auto e = hi - lo;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
    float3 center() const { return (lo + hi) * 0.5f; }
};

// 64-byte BVH2 node: both child boxes + two references (bit 31 = leaf: first | count << 27)
struct Bvh2Node {
    Aabb box[2];
    uint32_t child[2];
    uint32_t pad[2];
};
static_assert(sizeof(Bvh2Node) == 64u);

class Bvh2 {
public:
    std::vector<Bvh2Node> nodes;   // nodes[0] = root (if prim_count > leaf size)
    std::vector<uint32_t> prims;   // reordered primitive ids
    Aabb bounds;
    uint32_t root_leaf{0u};        // used when the whole set fits one leaf
    bool single_leaf{false};

    static constexpr uint32_t leaf_flag = 0x80000000u;
    static uint32_t make_leaf(uint32_t first, uint32_t count) { return leaf_flag | (count << 27u) | first; }
    static uint32_t leaf_first(uint32_t ref) { return ref & ((1u << 27u) - 1u); }
    static uint32_t leaf_count(uint32_t ref) { return (ref >> 27u) & 15u; }

    void build(const std::vector<Aabb> &boxes) {
        auto n = static_cast<uint32_t>(boxes.size());
        prims.resize(n);
        std::iota(prims.begin(), prims.end(), 0u);
        std::vector<float3> centers(n);
        bounds = Aabb{};
        for (uint32_t i = 0; i < n; i++) {
            centers[i] = boxes[i].center();
            bounds.grow(boxes[i]);
        }
        nodes.clear();
        if (n <= max_leaf) {
            single_leaf = true;
            root_leaf = make_leaf(0u, n);
            return;
        }
        nodes.reserve(n);
        nodes.emplace_back();
        _split(0u, 0u, n, boxes, centers);
    }

private:
    static constexpr uint32_t max_leaf = 4u;
    static constexpr uint32_t bins = 16u;

    // partitions [first, first + count) and fills node `index`
    void _split(uint32_t index, uint32_t first, uint32_t count, const std::vector<Aabb> &boxes,
                const std::vector<float3> &centers) {
        Aabb cbox;
        for (auto i = first; i < first + count; i++) { cbox.grow(centers[prims[i]]); }
        auto best_cost = std::numeric_limits<float>::max();
        auto best_axis = -1;
        auto best_bin = 0u;
        for (auto axis = 0; axis < 3; axis++) {
            auto extent = cbox.hi[axis] - cbox.lo[axis];
            if (!(extent > 0.f)) { continue; }
            std::array<Aabb, bins> bin_box{};
            std::array<uint32_t, bins> bin_n{};
            auto scale = static_cast<float>(bins) / extent;
            for (auto i = first; i < first + count; i++) {
                auto b = std::min(static_cast<uint32_t>((centers[prims[i]][axis] - cbox.lo[axis]) * scale), bins - 1u);
                bin_box[b].grow(boxes[prims[i]]);
                bin_n[b]++;
            }
            std::array<float, bins> r_area{};
            std::array<uint32_t, bins> r_n{};
            Aabb acc;
            auto n = 0u;
            for (auto b = bins - 1u; b > 0u; b--) {
                if (bin_n[b] != 0u) { acc.grow(bin_box[b]); }
                n += bin_n[b];
                r_area[b] = n ? acc.half_area() : 0.f;
                r_n[b] = n;
            }
            acc = Aabb{};
            n = 0u;
            for (auto b = 0u; b + 1u < bins; b++) {
                if (bin_n[b] != 0u) { acc.grow(bin_box[b]); }
                n += bin_n[b];
                if (n == 0u || r_n[b + 1u] == 0u) { continue; }
                auto cost = acc.half_area() * static_cast<float>(n) + r_area[b + 1u] * static_cast<float>(r_n[b + 1u]);
                if (cost < best_cost) { best_cost = cost, best_axis = axis, best_bin = b + 1u; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = first + count / 2u;
        } else {
            auto extent = cbox.hi[best_axis] - cbox.lo[best_axis];
            auto scale = static_cast<float>(bins) / extent;
            auto lo = cbox.lo[best_axis];
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](uint32_t p) {
                return std::min(static_cast<uint32_t>((centers[p][best_axis] - lo) * scale), bins - 1u) < best_bin;
            });
            mid = static_cast<uint32_t>(it - prims.begin());
            if (mid == first || mid == first + count) { mid = first + count / 2u; }
        }
        uint32_t range[2][2] = {{first, mid - first}, {mid, first + count - mid}};
        for (auto s = 0; s < 2; s++) {
            Aabb b;
            for (auto i = range[s][0]; i < range[s][0] + range[s][1]; i++) { b.grow(boxes[prims[i]]); }
            nodes[index].box[s] = b;
            if (range[s][1] <= max_leaf) {
                nodes[index].child[s] = make_leaf(range[s][0], range[s][1]);
            } else {
                auto c = static_cast<uint32_t>(nodes.size());
                nodes.emplace_back();
                nodes[index].child[s] = c;
                _split(c, range[s][0], range[s][1], boxes, centers);
            }
        }
    }
};

// slab test; `inv` may contain infinities (IEEE min/max drop NaNs like the GPU's v_min/v_max)
inline bool hit_box(const Aabb &b, float3 o, float3 inv, float t_min, float t_max, float &t_near) {
    auto t0 = (b.lo - o) * inv, t1 = (b.hi - o) * inv;
    auto tn = std::fmax(std::fmax(std::fmin(t0.x, t1.x), std::fmin(t0.y, t1.y)), std::fmax(std::fmin(t0.z, t1.z), t_min));
    auto tf = std::fmin(std::fmin(std::fmax(t0.x, t1.x), std::fmax(t0.y, t1.y)), std::fmin(std::fmax(t0.z, t1.z), t_max));
    t_near = tn;
    return tn <= tf * 1.0000004f;
}

// Moeller-Trumbore on (p0, e1 = p1 - p0, e2 = p2 - p0); accepts t in (t_min, t_max)
inline bool hit_triangle(float3 o, float3 d, float t_min, float t_max, float3 p0, float3 e1, float3 e2,
                         float &t, float &u, float &v) {
    auto pvec = cross(d, e2);
    auto det = dot(e1, pvec);
    if (det == 0.f) { return false; }
    auto inv_det = 1.0f / det;
    auto tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    auto qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    t = dot(e2, qvec) * inv_det;
    return u >= 0.f && v >= 0.f && u + v <= 1.f && t > t_min && t < t_max;
}

class Accel {
public:
    // Geometry::_alpha_skip hook (geometry.cpp:165-192): true = ignore this candidate hit
    std::function<bool(uint32_t inst, uint32_t prim, float u, float v)> alpha_skip;

private:
    const lr_scene &_scene;
    std::vector<Bvh2> _blas;                 // per mesh
    Bvh2 _tlas;
    struct InstanceXform {
        float w2o[12];                       // 3 rows x 4: object = W * (world, 1)
    };
    std::vector<InstanceXform> _xforms;
    // BAKED-GEOMETRY MODE (set_bake; test infrastructure, round 4).  The reference intersects object-space triangles with a ray taken
    // through the instance's inverse transform (geometry.cpp:218-260); the device intersects fp32 WORLD-space triangles, baked once by
    // the host (lr_scene.accel.triangles) -- a design choice that moves hits by an ulp and, on a large scene, makes device and
    // oracle stop tracing identical paths after a few bounces (tests/test_gpu_parity.py::test_what_separates_c2_from_the_oracle).
    // In this mode the oracle tests the SAME baked triangles with the world-space ray (its own BVHs still cull), so that what is
    // left between the two is the kernel alone.  The reference-side truth stays the default mode (tests/test_oracle_vs_ref.py).
    bool _bake{false};
    std::vector<uint32_t> _baked_of;       // [instance triangle base + prim] -> index into accel.triangles
    std::vector<uint32_t> _baked_base;     // per instance

public:
    bool set_bake(bool on) {
        if (on && _baked_of.empty()) {
            if (_scene.accel.triangles == nullptr || _scene.accel.triangle_count == 0u) { return false; }
            _baked_base.resize(_scene.instance_count);
            uint32_t total = 0u;
            for (uint32_t i = 0; i < _scene.instance_count; i++) {
                _baked_base[i] = total;
                total += _scene.meshes[_scene.instances[i].handle.x >> 10u].triangle_count;
            }
            _baked_of.assign(total, 0xffffffffu);
            for (uint32_t k = 0; k < _scene.accel.triangle_count; k++) {
                auto &bt = _scene.accel.triangles[k];
                if (bt.inst < _scene.instance_count) { _baked_of[_baked_base[bt.inst] + bt.prim] = k; }
            }
        }
        _bake = on;
        return true;
    }
    // the baked triangle of (instance, primitive) in baked-geometry mode, else nullptr
    const lr_bvh_triangle *baked(uint32_t inst, uint32_t prim) const {
        if (!_bake) { return nullptr; }
        const auto k = _baked_of[_baked_base[inst] + prim];
        return k != 0xffffffffu ? _scene.accel.triangles + k : nullptr;
    }

    explicit Accel(const lr_scene &scene) : _scene{scene} {
        _blas.resize(scene.mesh_count);
        for (uint32_t m = 0; m < scene.mesh_count; m++) {
            auto &mesh = scene.meshes[m];
            std::vector<Aabb> boxes(mesh.triangle_count);
            for (uint32_t i = 0; i < mesh.triangle_count; i++) {
                auto t = scene.triangles[mesh.triangle_offset + i];
                for (auto vi : {t.i0, t.i1, t.i2}) {
                    auto &v = scene.vertices[mesh.vertex_offset + vi];
                    boxes[i].grow(f3(v.px, v.py, v.pz));
                }
            }
            _blas[m].build(boxes);
        }
        std::vector<Aabb> inst_boxes(scene.instance_count);
        _xforms.resize(scene.instance_count);
        for (uint32_t i = 0; i < scene.instance_count; i++) {
            auto &inst = scene.instances[i];
            auto &mesh = scene.meshes[inst.handle.x >> 10u];
            auto m = inst.object_to_world;
            for (uint32_t v = 0; v < mesh.vertex_count; v++) {
                auto &vv = scene.vertices[mesh.vertex_offset + v];
                inst_boxes[i].grow(f3(m[0] * vv.px + m[4] * vv.py + m[8] * vv.pz + m[12],
                                      m[1] * vv.px + m[5] * vv.py + m[9] * vv.pz + m[13],
                                      m[2] * vv.px + m[6] * vv.py + m[10] * vv.pz + m[14]));
            }
            // affine inverse in double precision
            double a[3][3], inv[3][3];
            for (auto r = 0; r < 3; r++) {
                for (auto c = 0; c < 3; c++) { a[r][c] = m[c * 4 + r]; }
            }
            auto det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
            auto id = 1.0 / det;
            inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) * id, inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
            inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id, inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) * id;
            inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id, inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
            inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) * id, inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
            inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
            for (auto r = 0; r < 3; r++) {
                for (auto c = 0; c < 3; c++) { _xforms[i].w2o[r * 4 + c] = static_cast<float>(inv[r][c]); }
                _xforms[i].w2o[r * 4 + 3] = static_cast<float>(-(inv[r][0] * m[12] + inv[r][1] * m[13] + inv[r][2] * m[14]));
            }
        }
        _tlas.build(inst_boxes);
    }

    // closest hit (any = false) or any hit (any = true); `camera_or_shadow` rays respect the
    // instance visibility flag like Accel::emplace_back(..., visible, ...) (geometry.cpp:130)
    Hit trace(const Ray &ray, bool any, TraceCounters &counters) const {
        Hit hit;
        auto t_max = ray.t_max;
        auto inv_d = f3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
        auto visit_instance = [&](uint32_t inst_id) {
            auto &inst = _scene.instances[inst_id];
            if (!inst.visible) { return false; }
            auto &x = _xforms[inst_id];
            auto o = f3(x.w2o[0] * ray.o.x + x.w2o[1] * ray.o.y + x.w2o[2] * ray.o.z + x.w2o[3],
                        x.w2o[4] * ray.o.x + x.w2o[5] * ray.o.y + x.w2o[6] * ray.o.z + x.w2o[7],
                        x.w2o[8] * ray.o.x + x.w2o[9] * ray.o.y + x.w2o[10] * ray.o.z + x.w2o[11]);
            auto d = f3(x.w2o[0] * ray.d.x + x.w2o[1] * ray.d.y + x.w2o[2] * ray.d.z,
                        x.w2o[4] * ray.d.x + x.w2o[5] * ray.d.y + x.w2o[6] * ray.d.z,
                        x.w2o[8] * ray.d.x + x.w2o[9] * ray.d.y + x.w2o[10] * ray.d.z);
            auto mesh_id = inst.handle.x >> 10u;
            auto &mesh = _scene.meshes[mesh_id];
            auto &blas = _blas[mesh_id];
            auto inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            auto test_leaf = [&](uint32_t ref) {
                auto first = Bvh2::leaf_first(ref), count = Bvh2::leaf_count(ref);
                for (auto k = first; k < first + count; k++) {
                    auto prim = blas.prims[k];
                    auto tri = _scene.triangles[mesh.triangle_offset + prim];
                    auto &v0 = _scene.vertices[mesh.vertex_offset + tri.i0];
                    auto &v1 = _scene.vertices[mesh.vertex_offset + tri.i1];
                    auto &v2 = _scene.vertices[mesh.vertex_offset + tri.i2];
                    auto p0 = f3(v0.px, v0.py, v0.pz);
                    counters.tris++;
                    float t, u, v;
                    bool found;
                    if (_bake && _baked_of[_baked_base[inst_id] + prim] != 0xffffffffu) {// the device's triangle, the world-space ray
                        auto &bt = _scene.accel.triangles[_baked_of[_baked_base[inst_id] + prim]];
                        found = hit_triangle(ray.o, ray.d, ray.t_min, t_max, f3(bt.v0[0], bt.v0[1], bt.v0[2]), f3(bt.e1[0], bt.e1[1], bt.e1[2]),
                                             f3(bt.e2[0], bt.e2[1], bt.e2[2]), t, u, v);
                    } else {
                        found = hit_triangle(o, d, ray.t_min, t_max, p0, f3(v1.px, v1.py, v1.pz) - p0, f3(v2.px, v2.py, v2.pz) - p0, t, u, v);
                    }
                    if (found) {
                        if (alpha_skip && (inst.handle.x & LR_SHAPE_MAYBE_NON_OPAQUE) && alpha_skip(inst_id, prim, u, v)) { continue; }
                        t_max = t;
                        hit.inst = inst_id, hit.prim = prim, hit.bary = {u, v}, hit.t = t;
                        if (any) { return true; }
                    }
                }
                return false;
            };
            if (blas.single_leaf) { return test_leaf(blas.root_leaf); }
            uint32_t stack[64];
            auto sp = 0;
            stack[sp++] = 0u;
            while (sp > 0) {
                auto &node = blas.nodes[stack[--sp]];
                counters.nodes++;
                float tn[2];
                bool h[2];
                h[0] = hit_box(node.box[0], o, inv, ray.t_min, t_max, tn[0]);
                h[1] = hit_box(node.box[1], o, inv, ray.t_min, t_max, tn[1]);
                int order[2] = {0, 1};
                if (h[0] && h[1] && tn[1] < tn[0]) { order[0] = 1, order[1] = 0; }
                // push far first so that near is processed first; leaves are tested immediately in near->far order
                uint32_t inner[2];
                auto n_inner = 0;
                for (auto k = 0; k < 2; k++) {
                    auto c = order[k];
                    if (!h[c]) { continue; }
                    if (node.child[c] & Bvh2::leaf_flag) {
                        if (test_leaf(node.child[c]) && any) { return true; }
                    } else {
                        inner[n_inner++] = node.child[c];
                    }
                }
                for (auto k = n_inner - 1; k >= 0; k--) { stack[sp++] = inner[k]; }
            }
            return false;
        };
        if (_tlas.single_leaf) {
            auto first = Bvh2::leaf_first(_tlas.root_leaf), count = Bvh2::leaf_count(_tlas.root_leaf);
            for (auto k = first; k < first + count; k++) {
                if (visit_instance(_tlas.prims[k]) && any) { return hit; }
            }
            return hit;
        }
        uint32_t stack[64];
        auto sp = 0;
        stack[sp++] = 0u;
        while (sp > 0) {
            auto &node = _tlas.nodes[stack[--sp]];
            counters.nodes++;
            float tn[2];
            bool h[2];
            h[0] = hit_box(node.box[0], ray.o, inv_d, ray.t_min, t_max, tn[0]);
            h[1] = hit_box(node.box[1], ray.o, inv_d, ray.t_min, t_max, tn[1]);
            int order[2] = {0, 1};
            if (h[0] && h[1] && tn[1] < tn[0]) { order[0] = 1, order[1] = 0; }
            uint32_t inner[2];
            auto n_inner = 0;
            for (auto k = 0; k < 2; k++) {
                auto c = order[k];
                if (!h[c]) { continue; }
                if (node.child[c] & Bvh2::leaf_flag) {
                    auto first = Bvh2::leaf_first(node.child[c]), count = Bvh2::leaf_count(node.child[c]);
                    for (auto i = first; i < first + count; i++) {
                        if (visit_instance(_tlas.prims[i]) && any) { return hit; }
                    }
                } else {
                    inner[n_inner++] = node.child[c];
                }
            }
            for (auto k = n_inner - 1; k >= 0; k--) { stack[sp++] = inner[k]; }
        }
        return hit;
    }
};

}// namespace oracle
