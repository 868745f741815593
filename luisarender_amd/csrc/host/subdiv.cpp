// subdiv.cpp — Loop subdivision surfaces and the Sphere shape built on them.
//
//   loop_subdivide      src/util/loop_subdiv.cpp:131-377 (PBRT's LoopSubdiv: beta / gamma rules, boundary rules, limit
//                       positions, limit normals from the one-ring tangents)
//   LoopSubdiv shape    src/shapes/loop_subdiv.cpp:17-58   (`mesh` | `shape` | `base` child, `level` <= 10; normals, no uvs)
//   Sphere shape        src/shapes/sphere.cpp:17-131       (icosahedron, `subdivision` <= 8 Loop levels, pushed to the unit sphere)
//
// Restated on index arrays instead of the reference's pointer graph, in the same creation order: the children of the
// existing vertices first, then one new vertex per edge in face / edge order; four child faces per face.  One difference
// that cannot be avoided: the reference orders the two ends of an edge by POINTER value (SDEdge, :79-80), which decides
// the order of the two 3/8 terms of the edge rule and so the last bit of a new vertex; here the lower vertex index comes
// first.
#include "scene.h"

#include <array>
#include <cmath>
#include <unordered_map>

namespace lr {

namespace {

constexpr int next3(int e) { return (e + 1) % 3; }
constexpr int prev3(int e) { return (e + 2) % 3; }

struct SVertex {
    float3 p{};
    bool regular{false}, boundary{false};
    int start_face{-1};
    int child{-1};
};
struct SFace {
    int v[3]{-1, -1, -1};
    int f[3]{-1, -1, -1};
    int children[4]{-1, -1, -1, -1};
};

struct Level {
    std::vector<SVertex> verts;
    std::vector<SFace> faces;

    [[nodiscard]] int vnum(int face, int vert) const {
        for (auto i = 0; i < 3; i++) {
            if (faces[static_cast<size_t>(face)].v[i] == vert) { return i; }
        }
        throw Error{"Loop subdivision: inconsistent mesh topology."};
    }
    [[nodiscard]] int next_face(int face, int vert) const { return faces[static_cast<size_t>(face)].f[vnum(face, vert)]; }
    [[nodiscard]] int prev_face(int face, int vert) const { return faces[static_cast<size_t>(face)].f[prev3(vnum(face, vert))]; }
    [[nodiscard]] int next_vert(int face, int vert) const { return faces[static_cast<size_t>(face)].v[next3(vnum(face, vert))]; }
    [[nodiscard]] int prev_vert(int face, int vert) const { return faces[static_cast<size_t>(face)].v[prev3(vnum(face, vert))]; }
    [[nodiscard]] int other_vert(int face, int v0, int v1) const {
        for (auto i : faces[static_cast<size_t>(face)].v) {
            if (i != v0 && i != v1) { return i; }
        }
        throw Error{"Loop subdivision: degenerate face."};
    }
    [[nodiscard]] uint32_t valence(int vert) const {// SDVertex::valence, :102-116
        auto &v = verts[static_cast<size_t>(vert)];
        auto f = v.start_face;
        auto nf = 1u;
        if (!v.boundary) {
            while ((f = next_face(f, vert)) != v.start_face) { nf++; }
            return nf;
        }
        while ((f = next_face(f, vert)) != -1) { nf++; }
        f = v.start_face;
        while ((f = prev_face(f, vert)) != -1) { nf++; }
        return nf + 1u;
    }
    void one_ring(int vert, std::vector<float3> &ring) const {// SDVertex::oneRing, :388-409
        ring.clear();
        auto &v = verts[static_cast<size_t>(vert)];
        if (!v.boundary) {
            auto face = v.start_face;
            do {
                ring.push_back(verts[static_cast<size_t>(next_vert(face, vert))].p);
                face = next_face(face, vert);
            } while (face != v.start_face);
        } else {
            auto face = v.start_face;
            for (auto f2 = next_face(face, vert); f2 != -1; f2 = next_face(face, vert)) { face = f2; }
            ring.push_back(verts[static_cast<size_t>(next_vert(face, vert))].p);
            do {
                ring.push_back(verts[static_cast<size_t>(prev_vert(face, vert))].p);
                face = prev_face(face, vert);
            } while (face != -1);
        }
    }
    [[nodiscard]] float3 weight_one_ring(int vert, float beta, std::vector<float3> &ring) const {// :379-386
        one_ring(vert, ring);
        auto p = verts[static_cast<size_t>(vert)].p * (1.f - static_cast<float>(ring.size()) * beta);
        for (auto &r : ring) { p = p + r * beta; }
        return p;
    }
    [[nodiscard]] float3 weight_boundary(int vert, float beta, std::vector<float3> &ring) const {// :411-419
        one_ring(vert, ring);
        return verts[static_cast<size_t>(vert)].p * (1.f - 2.f * beta) + ring.front() * beta + ring.back() * beta;
    }
};

float loop_beta(uint32_t valence) { return 3.f / (valence == 3u ? 16.f : 8.f * static_cast<float>(valence)); }
float loop_gamma(uint32_t valence) { return 1.f / (static_cast<float>(valence) + 3.f / (8.f * loop_beta(valence))); }
uint64_t edge_key(int a, int b) { return (static_cast<uint64_t>(static_cast<uint32_t>(std::min(a, b))) << 32u) | static_cast<uint32_t>(std::max(a, b)); }

}// namespace

LoadedMesh loop_subdivide(const std::vector<lr_vertex> &vertices, const std::vector<lr_triangle> &triangles, uint32_t levels) {
    LoadedMesh out;
    if (levels == 0u) {
        out.vertices = vertices, out.triangles = triangles;
        return out;
    }
    Level cur;
    cur.verts.resize(vertices.size());
    for (size_t i = 0; i < vertices.size(); i++) { cur.verts[i].p = {vertices[i].px, vertices[i].py, vertices[i].pz}; }
    cur.faces.resize(triangles.size());
    for (size_t i = 0; i < triangles.size(); i++) {
        int idx[3] = {static_cast<int>(triangles[i].i0), static_cast<int>(triangles[i].i1), static_cast<int>(triangles[i].i2)};
        for (auto j = 0; j < 3; j++) {
            if (static_cast<size_t>(idx[j]) >= vertices.size()) { throw Error{"Loop subdivision: triangle index out of range."}; }
            cur.faces[i].v[j] = idx[j];
            cur.verts[static_cast<size_t>(idx[j])].start_face = static_cast<int>(i);
        }
    }
    {// neighbour faces through shared edges, :167-189
        struct Half { int face, edge; };
        std::unordered_map<uint64_t, Half> open;
        for (size_t i = 0; i < cur.faces.size(); i++) {
            for (auto e = 0; e < 3; e++) {
                auto key = edge_key(cur.faces[i].v[e], cur.faces[i].v[next3(e)]);
                if (auto it = open.find(key); it == open.end()) {
                    open.emplace(key, Half{static_cast<int>(i), e});
                } else {
                    cur.faces[static_cast<size_t>(it->second.face)].f[it->second.edge] = static_cast<int>(i);
                    cur.faces[i].f[e] = it->second.face;
                    open.erase(it);
                }
            }
        }
    }
    for (size_t i = 0; i < cur.verts.size(); i++) {// boundary / regular flags, :192-201
        auto &v = cur.verts[i];
        if (v.start_face < 0) { throw Error{"Loop subdivision: a vertex belongs to no triangle."}; }
        auto f = v.start_face;
        do { f = cur.next_face(f, static_cast<int>(i)); } while (f != -1 && f != v.start_face);
        v.boundary = f == -1;
        auto val = cur.valence(static_cast<int>(i));
        v.regular = (!v.boundary && val == 6u) || (v.boundary && val == 4u);
    }
    std::vector<float3> ring;
    for (auto level = 0u; level < levels; level++) {
        Level nxt;
        nxt.verts.resize(cur.verts.size());
        for (size_t i = 0; i < cur.verts.size(); i++) {// even vertices, :214-219,231-240
            cur.verts[i].child = static_cast<int>(i);
            auto &c = nxt.verts[i];
            c.regular = cur.verts[i].regular, c.boundary = cur.verts[i].boundary;
            auto vi = static_cast<int>(i);
            c.p = cur.verts[i].boundary ? cur.weight_boundary(vi, 1.f / 8.f, ring) :
                                          cur.weight_one_ring(vi, cur.verts[i].regular ? 1.f / 16.f : loop_beta(cur.valence(vi)), ring);
        }
        nxt.faces.resize(cur.faces.size() * 4u);
        for (size_t i = 0; i < cur.faces.size(); i++) {
            for (auto k = 0; k < 4; k++) { cur.faces[i].children[k] = static_cast<int>(i * 4u + static_cast<size_t>(k)); }
        }
        std::unordered_map<uint64_t, int> edge_vertex;// odd vertices, :243-268
        for (size_t i = 0; i < cur.faces.size(); i++) {
            auto &face = cur.faces[i];
            for (auto k = 0; k < 3; k++) {
                auto a = std::min(face.v[k], face.v[next3(k)]), b = std::max(face.v[k], face.v[next3(k)]);
                auto key = edge_key(a, b);
                if (edge_vertex.count(key)) { continue; }
                SVertex v;
                v.regular = true;
                v.boundary = face.f[k] == -1;
                v.start_face = face.children[3];
                auto pa = cur.verts[static_cast<size_t>(a)].p, pb = cur.verts[static_cast<size_t>(b)].p;
                if (v.boundary) {
                    v.p = pa * .5f + pb * .5f;
                } else {
                    v.p = pa * (3.f / 8.f) + pb * (3.f / 8.f) +
                          cur.verts[static_cast<size_t>(cur.other_vert(static_cast<int>(i), a, b))].p * (1.f / 8.f) +
                          cur.verts[static_cast<size_t>(cur.other_vert(face.f[k], a, b))].p * (1.f / 8.f);
                }
                edge_vertex.emplace(key, static_cast<int>(nxt.verts.size()));
                nxt.verts.push_back(v);
            }
        }
        for (size_t i = 0; i < cur.verts.size(); i++) {// :273-276
            auto &v = cur.verts[i];
            nxt.verts[i].start_face = cur.faces[static_cast<size_t>(v.start_face)].children[cur.vnum(v.start_face, static_cast<int>(i))];
        }
        for (size_t i = 0; i < cur.faces.size(); i++) {// child face neighbours, :279-293
            auto &face = cur.faces[i];
            for (auto j = 0; j < 3; j++) {
                nxt.faces[static_cast<size_t>(face.children[3])].f[j] = face.children[next3(j)];
                nxt.faces[static_cast<size_t>(face.children[j])].f[next3(j)] = face.children[3];
                auto f2 = face.f[j];
                nxt.faces[static_cast<size_t>(face.children[j])].f[j] = f2 != -1 ? cur.faces[static_cast<size_t>(f2)].children[cur.vnum(f2, face.v[j])] : -1;
                f2 = face.f[prev3(j)];
                nxt.faces[static_cast<size_t>(face.children[j])].f[prev3(j)] = f2 != -1 ? cur.faces[static_cast<size_t>(f2)].children[cur.vnum(f2, face.v[j])] : -1;
            }
        }
        for (size_t i = 0; i < cur.faces.size(); i++) {// child face vertices, :296-307
            auto &face = cur.faces[i];
            for (auto j = 0; j < 3; j++) {
                nxt.faces[static_cast<size_t>(face.children[j])].v[j] = cur.verts[static_cast<size_t>(face.v[j])].child;
                auto vert = edge_vertex.at(edge_key(face.v[j], face.v[next3(j)]));
                nxt.faces[static_cast<size_t>(face.children[j])].v[next3(j)] = vert;
                nxt.faces[static_cast<size_t>(face.children[next3(j)])].v[j] = vert;
                nxt.faces[static_cast<size_t>(face.children[3])].v[j] = vert;
            }
        }
        cur = std::move(nxt);
    }
    // limit positions, :316-322
    std::vector<float3> limit(cur.verts.size());
    for (size_t i = 0; i < cur.verts.size(); i++) {
        auto vi = static_cast<int>(i);
        limit[i] = cur.verts[i].boundary ? cur.weight_boundary(vi, 1.f / 5.f, ring) : cur.weight_one_ring(vi, loop_gamma(cur.valence(vi)), ring);
    }
    for (size_t i = 0; i < cur.verts.size(); i++) { cur.verts[i].p = limit[i]; }
    // limit normals from the one-ring tangents, :325-358
    constexpr auto pi = 3.14159265358979323846f;
    out.vertices.resize(cur.verts.size());
    for (size_t i = 0; i < cur.verts.size(); i++) {
        auto vi = static_cast<int>(i);
        cur.one_ring(vi, ring);
        auto valence = static_cast<uint32_t>(ring.size());
        float3 S{}, T{};
        auto p = cur.verts[i].p;
        if (!cur.verts[i].boundary) {
            for (auto j = 0u; j < valence; j++) {
                S = S + ring[j] * std::cos(2.f * pi * static_cast<float>(j) / static_cast<float>(valence));
                T = T + ring[j] * std::sin(2.f * pi * static_cast<float>(j) / static_cast<float>(valence));
            }
        } else {
            S = ring[valence - 1u] - ring[0];
            if (valence == 2u) { T = ring[0] + ring[1] - p * 2.f; }
            else if (valence == 3u) { T = ring[1] - p; }
            else if (valence == 4u) { T = ring[0] * -1.f + ring[1] * 2.f + ring[2] * 2.f + ring[3] * -1.f - p * 2.f; }
            else {
                auto theta = pi / static_cast<float>(valence - 1u);
                T = (ring[0] + ring[valence - 1u]) * std::sin(theta);
                for (auto k = 1u; k + 1u < valence; k++) { T = T + ring[k] * ((2.f * std::cos(theta) - 2.f) * std::sin(static_cast<float>(k) * theta)); }
                T = T * -1.f;
            }
        }
        auto n = normalize(cross(T, S));
        out.vertices[i] = lr_vertex{p.x, p.y, p.z, n.x, n.y, n.z, 0.f, 0.f};// (uvs are not carried through, :370 "FIXME: uv")
    }
    out.triangles.resize(cur.faces.size());
    for (size_t i = 0; i < cur.faces.size(); i++) {
        out.triangles[i] = {static_cast<uint32_t>(cur.faces[i].v[0]), static_cast<uint32_t>(cur.faces[i].v[1]), static_cast<uint32_t>(cur.faces[i].v[2])};
    }
    out.properties = LR_SHAPE_HAS_VERTEX_NORMAL;
    return out;
}

LoadedMesh make_sphere_mesh(uint32_t subdivision) {// SphereGeometry::create, sphere.cpp:67-104
    static const float base_positions[12][3] = {
        {0.f, -0.525731f, 0.850651f}, {0.850651f, 0.f, 0.525731f}, {0.850651f, 0.f, -0.525731f}, {-0.850651f, 0.f, -0.525731f},
        {-0.850651f, 0.f, 0.525731f}, {-0.525731f, 0.850651f, 0.f}, {0.525731f, 0.850651f, 0.f}, {0.525731f, -0.850651f, 0.f},
        {-0.525731f, -0.850651f, 0.f}, {0.f, -0.525731f, -0.850651f}, {0.f, 0.525731f, -0.850651f}, {0.f, 0.525731f, 0.850651f}};
    static const uint32_t base_triangles[20][3] = {
        {1, 2, 6}, {1, 7, 2}, {3, 4, 5}, {4, 3, 8}, {6, 5, 11}, {5, 6, 10}, {9, 10, 2}, {10, 9, 3}, {7, 8, 9}, {8, 7, 0},
        {11, 0, 1}, {0, 11, 4}, {6, 2, 10}, {1, 6, 11}, {3, 5, 10}, {5, 4, 11}, {2, 7, 9}, {7, 1, 0}, {3, 9, 8}, {4, 8, 0}};
    std::vector<lr_vertex> vertices(12);
    for (auto i = 0; i < 12; i++) {
        auto p = normalize(float3{base_positions[i][0], base_positions[i][1], base_positions[i][2]});
        vertices[static_cast<size_t>(i)] = lr_vertex{p.x, p.y, p.z, p.x, p.y, p.z, 0.f, 0.f};
    }
    std::vector<lr_triangle> triangles(20);
    for (auto i = 0; i < 20; i++) { triangles[static_cast<size_t>(i)] = {base_triangles[i][0], base_triangles[i][1], base_triangles[i][2]}; }
    auto mesh = loop_subdivide(vertices, triangles, std::min(subdivision, 8u));
    constexpr auto inv_pi = 0.318309886183790671537767526745028724f;
    for (auto &v : mesh.vertices) {
        float3 w{v.px, v.py, v.pz};// (the uv is taken from the position BEFORE it is pushed to the sphere, sphere.cpp:97-99)
        auto theta = std::acos(w.y), phi = std::atan2(w.x, w.z);
        auto fract = [](float x) { return x - std::floor(x); };
        auto p = normalize(w);
        v = lr_vertex{p.x, p.y, p.z, p.x, p.y, p.z, fract(.5f * inv_pi * phi), fract(theta * inv_pi)};
    }
    mesh.properties = LR_SHAPE_HAS_VERTEX_NORMAL | LR_SHAPE_HAS_VERTEX_UV;
    return mesh;
}

}// namespace lr
