// catmull_clark.cpp — the `subdivision` property of the Mesh shape (src/shapes/mesh.cpp:39,69,86-93,126-135).
//
// PARITY UNPINNED.  The reference hands the imported polygon mesh to assimp's Subdivider (CATMULL_CLARKE, discard_input = true);
// assimp is an un-vendored submodule (src/ext/assimp is empty in the snapshot), so there is no source and no output of it to
// hold this against.  What is restated here is the published algorithm (Catmull & Clark 1978) in the form assimp documents for
// its subdivider, as far as that is known:
//   * adjacency goes by POSITION (vertices that differ only in normal / uv are one topological vertex), every attribute
//     (position, normal, uv) goes through the same weights;
//   * face point = centroid; edge point = (end points + centroids of the adjacent faces) / (adjacent faces + 2) — a boundary
//     edge has one adjacent face; vertex point = (F + 2 R + (n - 3) P) / n with F the mean face point and R the mean edge
//     MIDPOINT around the vertex; a vertex on a boundary (edges around it != faces around it) stays where it is;
//   * every n-gon becomes n quads (vertex point, edge point towards the next corner, face point, edge point towards the previous
//     corner), and the reference splits a quad (0,1,2,3) into the triangles (0,1,2), (2,3,0) (mesh.cpp:126-135).
// Closed-form checks (tests/test_subdiv_shapes.py): a cube's first level, Euler counts, convergence of a cube to its limit
// surface, planar meshes stay planar, uv interpolation on a quad.  Vertex ORDER differs from assimp's for sure; it decides the
// order of an emitter's triangles in the light-sampling table, nothing else.
#include "scene.h"

#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <unordered_map>

namespace lr {

namespace {

struct Attr {// position, normal, uv: everything a vertex carries goes through the same weights
    float v[8]{};
    Attr operator+(const Attr &o) const {
        Attr r;
        for (auto i = 0; i < 8; i++) { r.v[i] = v[i] + o.v[i]; }
        return r;
    }
    Attr operator*(float s) const {
        Attr r;
        for (auto i = 0; i < 8; i++) { r.v[i] = v[i] * s; }
        return r;
    }
};

Attr attr_of(const lr_vertex &x) { return Attr{{x.px, x.py, x.pz, x.nx, x.ny, x.nz, x.u, x.v}}; }
lr_vertex vertex_of(const Attr &a) {
    lr_vertex x{};
    x.px = a.v[0], x.py = a.v[1], x.pz = a.v[2], x.nx = a.v[3], x.ny = a.v[4], x.nz = a.v[5], x.u = a.v[6], x.v = a.v[7];
    return x;
}

struct Edge {
    Attr sum{}, mid{};// end points + adjacent face points; midpoint
    uint32_t ref{0u};
};

uint64_t edge_key(uint32_t a, uint32_t b) { return (static_cast<uint64_t>(std::min(a, b)) << 32u) | std::max(a, b); }

struct Bits8 {
    uint32_t w[8];
    bool operator<(const Bits8 &o) const { return std::memcmp(w, o.w, sizeof(w)) < 0; }
};

}// namespace

PolygonMesh catmull_clark_level(const PolygonMesh &in) {
    const auto vertex_count = in.vertices.size();
    const auto face_count = in.face_offsets.size() - 1u;
    // 1. topological vertices: one per distinct position
    std::vector<uint32_t> rep(vertex_count);
    {
        std::map<std::array<uint32_t, 3>, uint32_t> by_position;
        for (size_t i = 0; i < vertex_count; i++) {
            std::array<uint32_t, 3> key;
            std::memcpy(key.data(), &in.vertices[i].px, 12);
            for (auto &k : key) { if (k == 0x80000000u) { k = 0u; } }// -0 == +0
            rep[i] = by_position.emplace(key, static_cast<uint32_t>(i)).first->second;
        }
    }
    // 2. face points
    std::vector<Attr> face_point(face_count);
    for (size_t f = 0; f < face_count; f++) {
        auto begin = in.face_offsets[f], end = in.face_offsets[f + 1u];
        if (end - begin < 3u) { throw Error{"Catmull-Clark subdivision: a face with fewer than three corners."}; }
        Attr sum{};
        for (auto k = begin; k < end; k++) { sum = sum + attr_of(in.vertices[in.indices[k]]); }
        face_point[f] = sum * (1.f / static_cast<float>(end - begin));
    }
    // 3. edge points
    std::unordered_map<uint64_t, Edge> edges;
    edges.reserve(in.indices.size());
    for (size_t f = 0; f < face_count; f++) {
        auto begin = in.face_offsets[f], end = in.face_offsets[f + 1u];
        for (auto k = begin; k < end; k++) {
            auto a = in.indices[k], b = in.indices[k + 1u == end ? begin : k + 1u];
            auto &e = edges[edge_key(rep[a], rep[b])];
            if (e.ref == 0u) {
                e.sum = attr_of(in.vertices[a]) + attr_of(in.vertices[b]);
                e.mid = e.sum * .5f;
            }
            e.ref++;
            if (e.ref <= 2u) { e.sum = e.sum + face_point[f]; }// (a non-manifold edge: the first two faces count)
        }
    }
    // 4. what surrounds each topological vertex: faces (one per corner) and distinct edges, summed in a fixed order (by face,
    //    then by corner) so that the result does not depend on the hash map's iteration order
    std::vector<uint32_t> adjacent_faces(vertex_count, 0u), adjacent_edges(vertex_count, 0u);
    std::vector<Attr> face_sum(vertex_count), mid_sum(vertex_count);
    {
        std::unordered_map<uint64_t, bool> seen;
        seen.reserve(edges.size());
        for (size_t f = 0; f < face_count; f++) {
            auto begin = in.face_offsets[f], end = in.face_offsets[f + 1u];
            for (auto k = begin; k < end; k++) {
                auto a = rep[in.indices[k]], b = rep[in.indices[k + 1u == end ? begin : k + 1u]];
                adjacent_faces[a]++;
                face_sum[a] = face_sum[a] + face_point[f];
                auto key = edge_key(a, b);
                if (!seen.emplace(key, true).second) { continue; }
                auto &e = edges[key];
                adjacent_edges[a]++;
                mid_sum[a] = mid_sum[a] + e.mid;
                if (b != a) {
                    adjacent_edges[b]++;
                    mid_sum[b] = mid_sum[b] + e.mid;
                }
            }
        }
    }
    // 5. vertex points, per ORIGINAL vertex (its own normal / uv), adjacency per topological vertex
    std::vector<Attr> vertex_point(vertex_count);
    for (size_t i = 0; i < vertex_count; i++) {
        auto r = rep[i];
        auto n = adjacent_faces[r];
        auto own = attr_of(in.vertices[i]);
        if (n == 0u || adjacent_edges[r] != n) {// unused, or on a boundary: stays
            vertex_point[i] = own;
            continue;
        }
        auto inv = 1.f / static_cast<float>(n);
        auto F = face_sum[r] * inv, R = mid_sum[r] * inv;
        vertex_point[i] = (F + R * 2.f + own * (static_cast<float>(n) - 3.f)) * inv;
    }
    // 6. n quads per n-gon; identical output vertices are shared
    PolygonMesh out;
    out.face_offsets.reserve(in.indices.size() + 1u);
    out.indices.reserve(in.indices.size() * 4u);
    out.face_offsets.emplace_back(0u);
    std::map<Bits8, uint32_t> joined;
    auto emit = [&](const Attr &a) {
        Bits8 key;
        std::memcpy(key.w, a.v, sizeof(key.w));
        auto it = joined.find(key);
        if (it == joined.end()) {
            it = joined.emplace(key, static_cast<uint32_t>(out.vertices.size())).first;
            out.vertices.emplace_back(vertex_of(a));
        }
        out.indices.emplace_back(it->second);
    };
    for (size_t f = 0; f < face_count; f++) {
        auto begin = in.face_offsets[f], end = in.face_offsets[f + 1u];
        for (auto k = begin; k < end; k++) {
            auto cur = in.indices[k];
            auto next = in.indices[k + 1u == end ? begin : k + 1u], prev = in.indices[k == begin ? end - 1u : k - 1u];
            auto &e_next = edges[edge_key(rep[cur], rep[next])], &e_prev = edges[edge_key(rep[cur], rep[prev])];
            emit(vertex_point[cur]);
            emit(e_next.sum * (1.f / static_cast<float>(e_next.ref + 2u)));
            emit(face_point[f]);
            emit(e_prev.sum * (1.f / static_cast<float>(e_prev.ref + 2u)));
            out.face_offsets.emplace_back(static_cast<uint32_t>(out.indices.size()));
        }
    }
    return out;
}

LoadedMesh catmull_clark_subdivide(const PolygonMesh &base, uint32_t levels, uint32_t properties) {
    auto mesh = base;
    for (auto l = 0u; l < levels; l++) { mesh = catmull_clark_level(mesh); }
    LoadedMesh out;
    out.properties = properties;
    out.vertices = std::move(mesh.vertices);
    for (auto &v : out.vertices) {// the reference normalises what the subdivider hands back (mesh.cpp:111-114)
        auto len = std::sqrt(v.nx * v.nx + v.ny * v.ny + v.nz * v.nz);
        if ((properties & LR_SHAPE_HAS_VERTEX_NORMAL) && len > 0.f) { v.nx /= len, v.ny /= len, v.nz /= len; }
        else { v.nx = 0.f, v.ny = 0.f, v.nz = 1.f; }
    }
    for (size_t f = 0; f + 1u < mesh.face_offsets.size(); f++) {
        auto b = mesh.face_offsets[f];
        if (mesh.face_offsets[f + 1u] - b != 4u) { throw Error{"Catmull-Clark subdivision: internal error, a face that is not a quad."}; }
        out.triangles.push_back({mesh.indices[b], mesh.indices[b + 1u], mesh.indices[b + 2u]});// mesh.cpp:131-132
        out.triangles.push_back({mesh.indices[b + 2u], mesh.indices[b + 3u], mesh.indices[b]});
    }
    return out;
}

}// namespace lr
