// accel.cpp — acceleration structure for the HIP traversal kernel.
//
// The reference delegates BVH build + traversal to LuisaCompute (`Accel`/`Mesh`, absent
// submodule; call sites src/base/geometry.cpp:16,26,66,130,221,250,265).  CDNA4 has no
// ray-tracing hardware, so traversal is ordinary VALU + memory and the layout is ours:
//   * instances are baked to world space (one level, no per-ray instance transform: 288 GB
//     of HBM makes the memory trade irrelevant, the saved dependent fetch + matrix math is
//     paid back on every ray);
//   * full-sweep SAH BVH2 collapsed (SAH-optimal dynamic programme) to a 4-wide BVH of fp32 child boxes SoA +
//     four child references (lr_bvh4_node); lrhip_upload_scene quantises it to 64-byte packets;
//   * every leaf is ONE 48-byte pre-transformed triangle (v0, e1, e2 + instance id, primitive id,
//     flags) for Moeller-Trumbore, named by index.
#include "scene.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <future>
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
// Builder choices, overridable from the environment for A/B runs (tools/bvh_sim.cpp counts node visits per ray on the CPU):
//   * exact SAH sweep for ranges up to sweep_threshold references, 16-bin SAH above (C2: 19.1 -> 16.7 node steps per ray with
//     the sweep everywhere; 32 / 64 / 128 bins: 18.2 / 18.0 / 17.2; C5 17.6 -> 16.4, C3 19.6 -> 19.2);
//   * SAH-optimal BVH2 -> BVH4 collapse instead of "open the largest child" (20 % fewer nodes in memory, -1 % steps per ray);
//   * insertion-based optimisation of the BVH2 before the collapse (Reinserter below): 16.7 -> 14.7 steps per ray after one
//     pass over all inner nodes (2 / 4 passes: 14.6 / 14.5), +4 s of build time for 600 k triangles; the worse half of the
//     nodes (by area x imbalance) carries most of it: fraction 0.5 -> 14.77 in half the time (0.25: 14.84);
//   * node order in memory: breadth-first, or depth-first over sibling groups (a subtree's nodes are contiguous; measured
//     on the device: no difference).
// Measured on the device (C2 at 256 spp, tools/gpu_call_bvh.sh): binned + greedy 554, sweep + optimal collapse 597,
// + one reinsertion pass 632 Msamples/s; C3 578 -> 608, C5 182 -> 185.
uint32_t sweep_threshold = 1u << 22u;
bool optimal_collapse = true;
bool depth_first = false;
float leaf_cost = 1.f;
uint32_t reinsertion_passes = 1u;
float reinsertion_fraction = 1.f;

class Builder2 {
    const std::vector<Box> &_boxes;
    const std::vector<float3> &_centroids;
    std::vector<uint32_t> &_indices;
    std::vector<Node2> &_nodes;// pre-sized to 2 n - 1 (one-reference leaves): subtrees are built by concurrent tasks
    std::atomic<uint32_t> _next{0u};
    static constexpr auto parallel_threshold = 16384u;

    // the two halves of a split: large ranges go to their own task (disjoint slices of _indices, distinct node slots)
    void build_children(uint32_t index, uint32_t first, uint32_t left_count, uint32_t count) {
        uint32_t l, r;
        if (count >= parallel_threshold) {
            auto task = std::async(std::launch::async, [&] { return build(first, left_count); });
            r = build(first + left_count, count - left_count);
            l = task.get();
        } else {
            l = build(first, left_count);
            r = build(first + left_count, count - left_count);
        }
        _nodes[index].left = l;
        _nodes[index].right = r;
    }

public:
    Builder2(const std::vector<Box> &boxes, const std::vector<float3> &centroids,
             std::vector<uint32_t> &indices, std::vector<Node2> &nodes)
        : _boxes{boxes}, _centroids{centroids}, _indices{indices}, _nodes{nodes} {}
    [[nodiscard]] uint32_t node_count() const { return _next.load(); }

    uint32_t build(uint32_t first, uint32_t count) {
        auto index = _next.fetch_add(1u);
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
        if (count <= sweep_threshold) {// exact SAH sweep over the three axes (small ranges: bins are too coarse there)
            auto best_cost = std::numeric_limits<float>::max();
            auto best_axis = -1;
            auto best_k = 0u;
            std::vector<float> right_area(count);
            for (auto axis = 0; axis < 3; axis++) {
                std::sort(_indices.begin() + first, _indices.begin() + first + count,
                          [&](uint32_t a, uint32_t b) { return _centroids[a][axis] < _centroids[b][axis]; });
                Box acc;
                for (auto k = count - 1u; k > 0u; k--) {
                    acc.grow(_boxes[_indices[first + k]]);
                    right_area[k] = acc.half_area();
                }
                acc = Box{};
                for (auto k = 1u; k < count; k++) {
                    acc.grow(_boxes[_indices[first + k - 1u]]);
                    auto cost = acc.half_area() * static_cast<float>(k) + right_area[k] * static_cast<float>(count - k);
                    // (ties — degenerate boxes of area 0, stacks of identical triangles — go to the most balanced split)
                    auto off_centre = [&](uint32_t q) { return q > count / 2u ? q - count / 2u : count / 2u - q; };
                    if (cost < best_cost || (cost == best_cost && off_centre(k) < off_centre(best_k))) { best_cost = cost, best_axis = axis, best_k = k; }
                }
            }
            if (best_axis != 2) {
                std::sort(_indices.begin() + first, _indices.begin() + first + count,
                          [&](uint32_t a, uint32_t b) { return _centroids[a][best_axis] < _centroids[b][best_axis]; });
            }
            build_children(index, first, best_k, count);
            return index;
        }
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
        build_children(index, first, mid - first, count);
        return index;
    }
};

// Insertion-based optimisation of the BVH2 (after Bittner, Hapala, Havran 2013): take an inner node out of the tree (its
// sibling moves up), then put its two child subtrees back where they increase the tree's total box area least — a
// branch-and-bound search from the root — reusing the two freed nodes as the new parents.  `passes` sweeps over the inner
// nodes, worst first (area x the area the children do not account for).
class Reinserter {
    std::vector<Node2> &_n;
    std::vector<uint32_t> _parent;

    void refit_up(uint32_t i) {
        while (i != LR_INVALID_ID) {
            auto &nd = _n[i];
            Box b = _n[nd.left].box;
            b.grow(_n[nd.right].box);
            nd.box = b;
            i = _parent[i];
        }
    }
    static float union_area(const Box &a, const Box &b) {
        Box u = a;
        u.grow(b);
        return u.half_area();
    }
    // best node to pair `x` with: minimises (area of the new parent) + (growth of all ancestors).  `fallback` is the pairing
    // that restores the tree as it was: another place is taken only if it is STRICTLY cheaper (degenerate inputs — stacks of
    // identical or zero-area boxes, where every place costs the same — keep their balanced sweep tree instead of growing chains)
    uint32_t find_target(uint32_t x, uint32_t fallback) {
        auto &bx = _n[x].box;
        auto ax = bx.half_area();
        auto best = fallback;
        auto best_cost = union_area(_n[fallback].box, bx);
        for (auto a = _parent[fallback]; a != LR_INVALID_ID; a = _parent[a]) { best_cost += union_area(_n[a].box, bx) - _n[a].box.half_area(); }
        struct Item { float induced; uint32_t node; };
        auto cmp = [](const Item &a, const Item &b) { return a.induced > b.induced; };
        std::vector<Item> heap{{0.f, 0u}};
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto it = heap.back();
            heap.pop_back();
            if (it.induced + ax >= best_cost) { break; }// every remaining entry is at least as expensive
            auto &nd = _n[it.node];
            auto direct = union_area(nd.box, bx);
            auto total = it.induced + direct;
            if (total < best_cost) { best_cost = total, best = it.node; }
            if (nd.count == 0u) {
                auto induced = it.induced + direct - nd.box.half_area();
                if (induced + ax < best_cost) {
                    heap.push_back({induced, nd.left});
                    std::push_heap(heap.begin(), heap.end(), cmp);
                    heap.push_back({induced, nd.right});
                    std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
        }
        return best;
    }
    // make `fresh` the parent of (target, x) in target's place
    void insert(uint32_t x, uint32_t target, uint32_t fresh) {
        if (target == 0u) {// pairing with the whole tree: the root keeps index 0, its old content moves into `fresh`
            _n[fresh] = _n[0];
            if (_n[fresh].count == 0u) { _parent[_n[fresh].left] = fresh, _parent[_n[fresh].right] = fresh; }
            _n[0].left = fresh, _n[0].right = x, _n[0].count = 0u;
            _parent[fresh] = 0u, _parent[x] = 0u;
            refit_up(0u);
            return;
        }
        auto p = _parent[target];
        _n[fresh].left = target, _n[fresh].right = x, _n[fresh].count = 0u;
        _parent[fresh] = p;
        if (p != LR_INVALID_ID) { (_n[p].left == target ? _n[p].left : _n[p].right) = fresh; }
        _parent[target] = fresh, _parent[x] = fresh;
        refit_up(fresh);
    }

public:
    explicit Reinserter(std::vector<Node2> &nodes) : _n{nodes}, _parent(nodes.size(), LR_INVALID_ID) {
        for (uint32_t i = 0; i < _n.size(); i++) {
            if (_n[i].count == 0u) { _parent[_n[i].left] = i, _parent[_n[i].right] = i; }
        }
    }
    [[nodiscard]] double total_area() const {
        auto sum = 0.0;
        for (auto &nd : _n) { sum += nd.count == 0u ? nd.box.half_area() : 0.f; }
        return sum;
    }
    void run(uint32_t passes, float fraction) {
        std::vector<std::pair<float, uint32_t>> order;
        for (auto pass = 0u; pass < passes; pass++) {
            order.clear();
            for (uint32_t i = 1; i < _n.size(); i++) {
                auto &nd = _n[i];
                if (nd.count != 0u || _parent[i] == 0u || _parent[i] == LR_INVALID_ID) { continue; }// keep the root and its children in place
                auto al = _n[nd.left].box.half_area(), ar = _n[nd.right].box.half_area();
                auto a = nd.box.half_area();
                // worst first = box area x the part of it the two children do not account for.  The ORDER matters (tools/bvh_sim.cpp,
                // node steps per ray on C2 / C5 / C3): this one 14.16 / 14.59 / 16.51; Bittner's area x imbalance x
                // relative size 14.65 / 14.80 / 16.49; plain area^2 / (al + ar) 14.22 / - / -; area^3 / (al + ar) 14.17 / 14.55 / 16.73
                auto inefficiency = a * (a - 0.5f * (al + ar));
                order.emplace_back(inefficiency, i);
            }
            auto take = std::max<size_t>(1u, static_cast<size_t>(fraction * static_cast<float>(order.size())));
            take = std::min(take, order.size());
            std::partial_sort(order.begin(), order.begin() + static_cast<std::ptrdiff_t>(take), order.end(), [](auto &a, auto &b) { return a.first > b.first; });
            for (size_t k = 0; k < take; k++) {
                auto i = order[k].second;
                auto p = _parent[i];
                if (_n[i].count != 0u || p == LR_INVALID_ID || p == 0u || _parent[p] == LR_INVALID_ID) { continue; }// moved next to the root meanwhile
                auto l = _n[i].left, r = _n[i].right;
                // take i and its parent p out: the sibling of i replaces p
                auto s = _n[p].left == i ? _n[p].right : _n[p].left;
                auto g = _parent[p];
                (_n[g].left == p ? _n[g].left : _n[g].right) = s;
                _parent[s] = g;
                refit_up(g);
                _parent[l] = _parent[r] = LR_INVALID_ID;
                if (_n[l].box.half_area() < _n[r].box.half_area()) { std::swap(l, r); }// the larger subtree first
                insert(l, find_target(l, s), p);// (l next to s, then r next to l, is the tree as it was)
                insert(r, find_target(r, l), i);
            }
        }
    }
};

// renumber the BVH2 in depth-first pre-order (children get larger indices than their parent, which the collapse relies
// on) and collect the leaves' references in tree order
void relinearise(std::vector<Node2> &nodes, std::vector<uint32_t> &indices) {
    std::vector<Node2> out;
    out.reserve(nodes.size());
    std::vector<uint32_t> new_indices;
    new_indices.reserve(indices.size());
    struct Work { uint32_t old_id, new_parent; bool is_right; };
    std::vector<Work> stack{{0u, LR_INVALID_ID, false}};
    while (!stack.empty()) {
        auto w = stack.back();
        stack.pop_back();
        auto id = static_cast<uint32_t>(out.size());
        out.push_back(nodes[w.old_id]);
        if (w.new_parent != LR_INVALID_ID) { (w.is_right ? out[w.new_parent].right : out[w.new_parent].left) = id; }
        auto &nd = out.back();
        if (nd.count > 0u) {
            auto first = static_cast<uint32_t>(new_indices.size());
            for (auto k = 0u; k < nd.count; k++) { new_indices.push_back(indices[nd.first + k]); }
            nd.first = first;
        } else {
            stack.push_back({nd.right, id, true});
            stack.push_back({nd.left, id, false});
        }
    }
    if (new_indices.size() != indices.size()) { throw Error{"BVH2 lost references during optimisation (internal error)."}; }
    nodes = std::move(out);
    indices = std::move(new_indices);
}

}// namespace

void build_accel(SceneData &scene) {
    if (auto e = std::getenv("LR_BVH_SWEEP")) { sweep_threshold = static_cast<uint32_t>(std::atoi(e)); }
    if (auto e = std::getenv("LR_BVH_COLLAPSE")) { optimal_collapse = std::atoi(e) != 0; }
    if (auto e = std::getenv("LR_BVH_REINSERT")) { reinsertion_passes = static_cast<uint32_t>(std::atoi(e)); }
    if (auto e = std::getenv("LR_BVH_REINSERT_FRACTION")) { reinsertion_fraction = static_cast<float>(std::atof(e)); }
    if (auto e = std::getenv("LR_BVH_DFS")) { depth_first = std::atoi(e) != 0; }
    if (auto e = std::getenv("LR_BVH_LEAF_COST")) { leaf_cost = static_cast<float>(std::atof(e)); }
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
    // references: (box, triangle); a one-triangle leaf names its triangle by index, so the builder is free to reorder (and a
    // later splitting builder to duplicate) references.  Early split clipping of large triangles was measured with
    // tools/bvh_sim.cpp and NOT kept: C2 tris/ray 3.8 -> 2.8 but nodes/ray 18.9 -> 22.0, and the node step is the expensive one.
    struct Ref { Box box; uint32_t tri; };
    std::vector<Ref> refs(n);
    for (uint32_t i = 0; i < n; i++) {
        float3 p0{tris[i].v0[0], tris[i].v0[1], tris[i].v0[2]};
        auto p1 = p0 + float3{tris[i].e1[0], tris[i].e1[1], tris[i].e1[2]};
        auto p2 = p0 + float3{tris[i].e2[0], tris[i].e2[1], tris[i].e2[2]};
        refs[i].tri = i;
        refs[i].box.grow(p0), refs[i].box.grow(p1), refs[i].box.grow(p2);
    }
    auto ref_count = static_cast<uint32_t>(refs.size());
    if (ref_count >= (1u << 27u)) { throw Error{"Too many BVH references."}; }
    std::vector<Box> boxes(ref_count);
    std::vector<float3> centroids(ref_count);
    for (uint32_t i = 0; i < ref_count; i++) {
        boxes[i] = refs[i].box;
        centroids[i] = (boxes[i].lo + boxes[i].hi) * 0.5f;
    }
    // 2. BVH2
    std::vector<uint32_t> indices(ref_count);
    std::iota(indices.begin(), indices.end(), 0u);
    std::vector<Node2> nodes2(static_cast<size_t>(ref_count) * 2u);
    {
        Builder2 builder{boxes, centroids, indices, nodes2};
        builder.build(0u, ref_count);
        nodes2.resize(builder.node_count());
    }
    // the concurrent build hands out node slots in whatever order its tasks run: renumber first, so that the optimiser (whose
    // tie-breaks go by index) and everything after it see the same tree in every run and on every rank
    relinearise(nodes2, indices);
    if (reinsertion_passes > 0u && nodes2.size() > 7u) {
        Reinserter opt{nodes2};
        auto before = opt.total_area();
        opt.run(reinsertion_passes, reinsertion_fraction);
        log_info("BVH2 reinsertion: total inner box area " + std::to_string(before) + " -> " + std::to_string(opt.total_area()));
    }
    relinearise(nodes2, indices);// (the concurrent build and the reinsertion leave the nodes in no particular order)
    // 3. collapse to BVH4 (expand the child with the largest area until four children); triangles are stored in the
    // order their first reference appears in the leaf sequence
    scene.bvh_nodes.clear();
    scene.bvh_triangles.clear();
    scene.bvh_triangles.reserve(n);
    std::vector<uint32_t> tri_slot(n, LR_INVALID_ID), ref_slot(ref_count);
    for (uint32_t i = 0; i < ref_count; i++) {
        auto tri = refs[indices[i]].tri;
        if (tri_slot[tri] == LR_INVALID_ID) {
            tri_slot[tri] = static_cast<uint32_t>(scene.bvh_triangles.size());
            scene.bvh_triangles.emplace_back(tris[tri]);
        }
        ref_slot[i] = tri_slot[tri];
    }
    auto encode_leaf = [&](const Node2 &leaf) { return 0x80000000u | ref_slot[leaf.first]; };// count == 1
    // SAH-optimal collapse (dynamic programme over the BVH2, after Ylitie et al. 2017).  t(n) = cost of n as ONE child of a
    // BVH4 node (a leaf: leaf_cost x area; else area(n) + D(n, 4)); F(n, i) = cheapest forest covering n in at most i slots of
    // an ancestor's node = min(F(n, i-1), D(n, i)); D(n, j) = min_k F(left, k) + F(right, j-k).
    std::vector<float> forest;    // [n][i-1], i = 1..3
    std::vector<uint8_t> choice;  // [n][i-1]: the j <= i used by F(n, i) (1 = n stays one child)
    std::vector<uint8_t> k_of;    // [n][j-2], j = 2..4: slots given to the left side by D(n, j)
    if (optimal_collapse) {
        forest.assign(nodes2.size() * 3u, 0.f);
        choice.assign(nodes2.size() * 3u, 1u);
        k_of.assign(nodes2.size() * 3u, 1u);
        for (auto ni = nodes2.size(); ni-- > 0u;) {// children have larger indices than their parent
            auto &nd = nodes2[ni];
            auto f = &forest[ni * 3u];
            auto area = nd.box.half_area();
            if (nd.count > 0u) {
                f[0] = f[1] = f[2] = area * leaf_cost;
                continue;
            }
            auto fl = &forest[static_cast<size_t>(nd.left) * 3u], fr = &forest[static_cast<size_t>(nd.right) * 3u];
            float d[5];
            for (auto j = 2u; j <= 4u; j++) {
                d[j] = std::numeric_limits<float>::max();
                for (auto k = 1u; k < j; k++) {
                    auto v = fl[k - 1u] + fr[j - k - 1u];
                    if (v < d[j]) { d[j] = v, k_of[ni * 3u + j - 2u] = static_cast<uint8_t>(k); }
                }
            }
            f[0] = area + d[4];
            for (auto i = 2u; i <= 3u; i++) {
                if (d[i] < f[i - 2u]) { f[i - 1u] = d[i], choice[ni * 3u + i - 1u] = static_cast<uint8_t>(i); }
                else { f[i - 1u] = f[i - 2u], choice[ni * 3u + i - 1u] = choice[ni * 3u + i - 2u]; }
            }
        }
    }
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
    // node order: breadth-first (queue) or depth-first over sibling groups (stack: a subtree's nodes are contiguous)
    for (size_t qi = 0; depth_first ? !queue.empty() : qi < queue.size(); qi++) {
        Work work;
        if (depth_first) { work = queue.back(), queue.pop_back(); }
        else { work = queue[qi]; }
        std::array<uint32_t, 4> children{};
        auto child_count = 0u;
        if (optimal_collapse) {// follow the DP's choices
            struct Pick { uint32_t node, slots; bool forced; };
            std::vector<Pick> todo{{work.node2, 4u, true}};
            while (!todo.empty()) {
                auto pk = todo.back();
                todo.pop_back();
                auto &c = nodes2[pk.node];
                auto j = pk.slots;
                if (!pk.forced) {
                    if (c.count > 0u) { children[child_count++] = pk.node; continue; }
                    j = choice[static_cast<size_t>(pk.node) * 3u + pk.slots - 1u];
                    if (j == 1u) { children[child_count++] = pk.node; continue; }
                }
                auto k = k_of[static_cast<size_t>(pk.node) * 3u + j - 2u];
                todo.push_back({c.right, j - k, false});
                todo.push_back({c.left, k, false});
            }
        } else {
        child_count = 2u;
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

// Geometry::update (geometry.cpp:194-216) sets the moved instances' transforms and rebuilds the top-level structure.  Instances are
// baked here, so the moved triangles are re-baked and the boxes of the EXISTING tree are refitted bottom-up (children
// have larger indices than their parent in both node orders): O(n), no rebuild per shutter sample.
void refit_accel(SceneData &scene) {
    std::vector<char> moved(scene.instances.size(), 0);
    for (auto &d : scene.dynamic_instances) { moved[d.instance] = 1; }
    for (auto &bt : scene.bvh_triangles) {
        if (!moved[bt.inst]) { continue; }
        auto &inst = scene.instances[bt.inst];
        auto &mesh = scene.meshes[inst.handle.x >> 10u];
        float4x4 m;
        std::memcpy(&m, inst.object_to_world, sizeof(m));
        auto t = scene.triangles[mesh.triangle_offset + bt.prim];
        auto fetch = [&](uint32_t i) {
            auto &v = scene.vertices[mesh.vertex_offset + i];
            return transform_point(m, {v.px, v.py, v.pz});
        };
        auto p0 = fetch(t.i0), p1 = fetch(t.i1), p2 = fetch(t.i2);
        auto e1 = p1 - p0, e2 = p2 - p0;
        bt.v0[0] = p0.x, bt.v0[1] = p0.y, bt.v0[2] = p0.z;
        bt.e1[0] = e1.x, bt.e1[1] = e1.y, bt.e1[2] = e1.z;
        bt.e2[0] = e2.x, bt.e2[1] = e2.y, bt.e2[2] = e2.z;
    }
    for (auto ni = scene.bvh_nodes.size(); ni-- > 0u;) {
        auto &node = scene.bvh_nodes[ni];
        for (auto i = 0; i < 4; i++) {
            auto c = node.child[i];
            if (c == LR_INVALID_ID) { continue; }
            Box b;
            if (c & 0x80000000u) {
                auto &bt = scene.bvh_triangles[c & ((1u << 27u) - 1u)];
                float3 p0{bt.v0[0], bt.v0[1], bt.v0[2]};
                b.grow(p0), b.grow(p0 + float3{bt.e1[0], bt.e1[1], bt.e1[2]}), b.grow(p0 + float3{bt.e2[0], bt.e2[1], bt.e2[2]});
            } else {
                auto &ch = scene.bvh_nodes[c];
                for (auto k = 0; k < 4; k++) {
                    if (ch.child[k] == LR_INVALID_ID) { continue; }
                    b.grow(float3{ch.lo_x[k], ch.lo_y[k], ch.lo_z[k]}), b.grow(float3{ch.hi_x[k], ch.hi_y[k], ch.hi_z[k]});
                }
            }
            node.lo_x[i] = b.lo.x, node.lo_y[i] = b.lo.y, node.lo_z[i] = b.lo.z;
            node.hi_x[i] = b.hi.x, node.hi_y[i] = b.hi.y, node.hi_z[i] = b.hi.z;
        }
    }
}

}// namespace lr
