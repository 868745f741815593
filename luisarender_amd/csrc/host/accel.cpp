// accel.cpp — acceleration structure for the HIP traversal kernel.
//
// The reference delegates BVH build + traversal to LuisaCompute (`Accel`/`Mesh`, absent
// submodule; call sites src/base/geometry.cpp:16,26,66,130,221,250,265).  CDNA4 has no
// ray-tracing hardware, so traversal is ordinary VALU + memory and the layout is ours:
//   * instances are baked to world space (one level, no per-ray instance transform: 288 GB
//     of HBM makes the memory trade irrelevant, the saved dependent fetch + matrix math is
//     paid back on every ray);
//   * binned-SAH BVH2 (16 bins) collapsed to a 4-wide BVH whose 128-byte nodes hold the four
//     child boxes SoA (six float4) + four child references: one node = one 128 B cache line,
//     read with eight coalescable dwordx4 loads per lane;
//   * leaves reference contiguous runs of 48-byte pre-transformed triangles
//     (v0, e1, e2 + instance id, primitive id, flags) for Moeller-Trumbore.
#include "scene.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

namespace lr {

namespace {

struct Box {
    float3 lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float3 hi{-std::numeric_limits<float>::max(), -std::numeric_limits<float>::max(), -std::numeric_limits<float>::max()};
    void grow(float3 p) { lo = min3(lo, p), hi = max3(hi, p); }
    void grow(const Box &b) { lo = min3(lo, b.lo), hi = max3(hi, b.hi); }
    [[nodiscard]] float half_area() const {
        auto d = hi - lo;
        return d.x * d.y + d.y * d.z + d.z * d.x;
    }
};

struct Node2 {
    Box box;
    uint32_t left{0}, right{0};// children (inner) ...
    uint32_t first{0}, count{0};// ... or primitive range (leaf, count > 0)
};

// One triangle per leaf: the megakernel's lanes sit at different tree depths, so a leaf loop would run for the
// longest leaf of the wave every traversal step (dev_trace.h); an extra level of boxes is cheaper.
constexpr auto max_leaf_size = 1u;
constexpr auto bin_count = 16u;

class Builder2 {
    const std::vector<Box> &_boxes;
    const std::vector<float3> &_centroids;
    std::vector<uint32_t> &_indices;
    std::vector<Node2> &_nodes;

public:
    Builder2(const std::vector<Box> &boxes, const std::vector<float3> &centroids,
             std::vector<uint32_t> &indices, std::vector<Node2> &nodes)
        : _boxes{boxes}, _centroids{centroids}, _indices{indices}, _nodes{nodes} {}

    uint32_t build(uint32_t first, uint32_t count) {
        auto index = static_cast<uint32_t>(_nodes.size());
        _nodes.emplace_back();
        Box box, cbox;
        for (auto i = first; i < first + count; i++) {
            box.grow(_boxes[_indices[i]]);
            cbox.grow(_centroids[_indices[i]]);
        }
        _nodes[index].box = box;
        auto make_leaf = [&] {
            _nodes[index].first = first;
            _nodes[index].count = count;
            return index;
        };
        if (count <= max_leaf_size) { return make_leaf(); }
        // binned SAH over the widest centroid axis and the two others
        auto best_cost = std::numeric_limits<float>::max();
        auto best_axis = -1;
        auto best_split = 0u;
        for (auto axis = 0; axis < 3; axis++) {
            auto extent = cbox.hi[axis] - cbox.lo[axis];
            if (!(extent > 0.f)) { continue; }
            std::array<Box, bin_count> bins{};
            std::array<uint32_t, bin_count> counts{};
            auto scale = static_cast<float>(bin_count) / extent;
            for (auto i = first; i < first + count; i++) {
                auto b = std::min(static_cast<uint32_t>((_centroids[_indices[i]][axis] - cbox.lo[axis]) * scale), bin_count - 1u);
                bins[b].grow(_boxes[_indices[i]]);
                counts[b]++;
            }
            std::array<float, bin_count> right_area{};
            std::array<uint32_t, bin_count> right_count{};
            Box acc;
            auto n = 0u;
            for (auto b = bin_count - 1u; b > 0u; b--) {
                acc.grow(bins[b]);
                n += counts[b];
                right_area[b] = acc.half_area();
                right_count[b] = n;
            }
            acc = Box{};
            n = 0u;
            for (auto b = 0u; b + 1u < bin_count; b++) {
                acc.grow(bins[b]);
                n += counts[b];
                if (n == 0u || right_count[b + 1u] == 0u) { continue; }
                auto cost = acc.half_area() * static_cast<float>(n) + right_area[b + 1u] * static_cast<float>(right_count[b + 1u]);
                if (cost < best_cost) { best_cost = cost, best_axis = axis, best_split = b + 1u; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {// all centroids coincide: split in the middle
            mid = first + count / 2u;
        } else {
            auto extent = cbox.hi[best_axis] - cbox.lo[best_axis];
            auto scale = static_cast<float>(bin_count) / extent;
            auto lo = cbox.lo[best_axis];
            auto it = std::partition(_indices.begin() + first, _indices.begin() + first + count, [&](uint32_t i) {
                auto b = std::min(static_cast<uint32_t>((_centroids[i][best_axis] - lo) * scale), bin_count - 1u);
                return b < best_split;
            });
            mid = static_cast<uint32_t>(it - _indices.begin());
            if (mid == first || mid == first + count) { mid = first + count / 2u; }
        }
        auto l = build(first, mid - first);
        auto r = build(mid, first + count - mid);
        _nodes[index].left = l;
        _nodes[index].right = r;
        return index;
    }
};

}// namespace

void build_accel(SceneData &scene) {
    // 1. bake instances into world-space triangles
    uint64_t total = 0;
    for (auto &inst : scene.instances) { total += inst.handle.z; }
    if (total >= (1u << 27u) - 16u) { throw Error{"Too many triangles for the 27-bit leaf reference."}; }
    std::vector<lr_bvh_triangle> tris;
    tris.reserve(total);
    for (uint32_t inst_id = 0; inst_id < scene.instances.size(); inst_id++) {
        auto &inst = scene.instances[inst_id];
        auto &mesh = scene.meshes[inst.handle.x >> 10u];
        float4x4 m;
        std::memcpy(&m, inst.object_to_world, sizeof(m));
        auto opaque = (inst.handle.x & LR_SHAPE_MAYBE_NON_OPAQUE) == 0u;
        for (uint32_t prim = 0; prim < mesh.triangle_count; prim++) {
            auto t = scene.triangles[mesh.triangle_offset + prim];
            auto fetch = [&](uint32_t i) {
                auto &v = scene.vertices[mesh.vertex_offset + i];
                return transform_point(m, {v.px, v.py, v.pz});
            };
            auto p0 = fetch(t.i0), p1 = fetch(t.i1), p2 = fetch(t.i2);
            auto e1 = p1 - p0, e2 = p2 - p0;
            lr_bvh_triangle bt{};
            bt.v0[0] = p0.x, bt.v0[1] = p0.y, bt.v0[2] = p0.z;
            bt.e1[0] = e1.x, bt.e1[1] = e1.y, bt.e1[2] = e1.z;
            bt.e2[0] = e2.x, bt.e2[1] = e2.y, bt.e2[2] = e2.z;
            bt.inst = inst_id, bt.prim = prim;
            bt.flags = (inst.visible ? 1u : 0u) | (opaque ? 2u : 0u);
            tris.emplace_back(bt);
        }
    }
    auto n = static_cast<uint32_t>(tris.size());
    std::vector<Box> boxes(n);
    std::vector<float3> centroids(n);
    for (uint32_t i = 0; i < n; i++) {
        float3 p0{tris[i].v0[0], tris[i].v0[1], tris[i].v0[2]};
        float3 p1 = p0 + float3{tris[i].e1[0], tris[i].e1[1], tris[i].e1[2]};
        float3 p2 = p0 + float3{tris[i].e2[0], tris[i].e2[1], tris[i].e2[2]};
        boxes[i].grow(p0), boxes[i].grow(p1), boxes[i].grow(p2);
        centroids[i] = (boxes[i].lo + boxes[i].hi) * 0.5f;
    }
    // 2. BVH2
    std::vector<uint32_t> indices(n);
    std::iota(indices.begin(), indices.end(), 0u);
    std::vector<Node2> nodes2;
    nodes2.reserve(static_cast<size_t>(n) * 2u / max_leaf_size + 16u);
    Builder2{boxes, centroids, indices, nodes2}.build(0u, n);
    // 3. collapse to BVH4 (expand the child with the largest area until four children)
    scene.bvh_nodes.clear();
    scene.bvh_triangles.resize(n);
    for (uint32_t i = 0; i < n; i++) { scene.bvh_triangles[i] = tris[indices[i]]; }
    auto encode_leaf = [](const Node2 &leaf) { return 0x80000000u | ((leaf.count - 1u) << 27u) | leaf.first; };
    struct Work { uint32_t node2, node4; };
    std::vector<Work> queue;
    if (nodes2[0].count > 0u) {// degenerate: the root is a leaf -> one BVH4 node with one leaf child
        lr_bvh4_node root{};
        for (auto &c : root.child) { c = LR_INVALID_ID; }
        auto &b = nodes2[0].box;
        root.lo_x[0] = b.lo.x, root.lo_y[0] = b.lo.y, root.lo_z[0] = b.lo.z;
        root.hi_x[0] = b.hi.x, root.hi_y[0] = b.hi.y, root.hi_z[0] = b.hi.z;
        root.child[0] = encode_leaf(nodes2[0]);
        scene.bvh_nodes.emplace_back(root);
    } else {
        scene.bvh_nodes.emplace_back();
        queue.push_back({0u, 0u});
    }
    for (size_t qi = 0; qi < queue.size(); qi++) {
        auto work = queue[qi];
        std::array<uint32_t, 4> children{};
        auto child_count = 2u;
        children[0] = nodes2[work.node2].left, children[1] = nodes2[work.node2].right;
        while (child_count < 4u) {
            auto best = -1;
            auto best_area = -1.f;
            for (auto i = 0u; i < child_count; i++) {
                auto &c = nodes2[children[i]];
                if (c.count == 0u && c.box.half_area() > best_area) { best_area = c.box.half_area(), best = static_cast<int>(i); }
            }
            if (best < 0) { break; }
            auto expanded = children[static_cast<size_t>(best)];
            children[static_cast<size_t>(best)] = nodes2[expanded].left;
            children[child_count++] = nodes2[expanded].right;
        }
        lr_bvh4_node node{};
        for (auto &c : node.child) { c = LR_INVALID_ID; }
        for (auto i = 0u; i < child_count; i++) {
            auto &c = nodes2[children[i]];
            node.lo_x[i] = c.box.lo.x, node.lo_y[i] = c.box.lo.y, node.lo_z[i] = c.box.lo.z;
            node.hi_x[i] = c.box.hi.x, node.hi_y[i] = c.box.hi.y, node.hi_z[i] = c.box.hi.z;
            if (c.count > 0u) {
                node.child[i] = encode_leaf(c);
            } else {
                auto id = static_cast<uint32_t>(scene.bvh_nodes.size());
                scene.bvh_nodes.emplace_back();
                node.child[i] = id;
                queue.push_back({children[i], id});
            }
        }
        scene.bvh_nodes[work.node4] = node;
    }
    log_info("BVH4 built: " + std::to_string(scene.bvh_nodes.size()) + " nodes over " + std::to_string(n) + " triangles.");
}

}// namespace lr
