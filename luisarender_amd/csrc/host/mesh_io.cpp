// mesh_io.cpp — Wavefront OBJ reader standing in for assimp in the reference's Mesh shape.
// Reproduces the observable effect of the import flags at src/shapes/mesh.cpp:46-69:
//   * polygons are fan-triangulated (aiProcess_Triangulate), lines/points dropped; with `subdivision` > 0 they are kept and
//     subdivided instead (mesh.cpp:69,86-93; catmull_clark.cpp);
//   * identical (position, normal, uv) corners are merged (aiProcess_JoinIdenticalVertices);
//   * V is flipped (v -> 1 - v) unless `flip_uv` (the reference passes aiProcess_FlipUVs when
//     flip_uv is *false*, mesh.cpp:61);
//   * missing normals are generated smooth with a 45 degree crease limit
//     (aiProcess_GenSmoothNormals + AI_CONFIG_PP_GSN_MAX_SMOOTHING_ANGLE);
//   * drop_normal / drop_uv remove the attribute.
// Vertex order after joining is first-use order; assimp's ImproveCacheLocality reorder is not
// reproduced (it changes indices, not geometry).
#include "scene.h"

#include <array>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace lr {

namespace {

struct Corner {
    int p{-1}, t{-1}, n{-1};
};

int resolve_index(int idx, size_t count) {
    if (idx > 0) { return idx - 1; }
    if (idx < 0) { return static_cast<int>(count) + idx; }
    return -1;
}

}// namespace

LoadedMesh load_obj_mesh(const std::string &path, bool flip_uv, bool drop_normal, bool drop_uv, uint32_t subdivision) {
    std::ifstream file{path};
    if (!file) { throw Error{"Failed to load mesh '" + path + "'."}; }
    auto ext_pos = path.find_last_of('.');
    auto ext = ext_pos == std::string::npos ? std::string{} : path.substr(ext_pos);
    for (auto &c : ext) { c = static_cast<char>(std::tolower(c)); }
    if (ext != ".obj") {
        throw Error{"Only Wavefront OBJ meshes are supported without assimp: '" + path + "'."};
    }
    std::vector<float3> positions, normals;
    std::vector<float2> uvs;
    std::vector<std::vector<Corner>> faces;// triangles (fan) unless the mesh is to be subdivided: then the polygons as they are
    std::string line;
    while (std::getline(file, line)) {
        if (line.size() < 2u) { continue; }
        const char *s = line.c_str();
        while (*s == ' ' || *s == '\t') { s++; }
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float3 p{};
            if (std::sscanf(s + 1, "%f %f %f", &p.x, &p.y, &p.z) == 3) { positions.emplace_back(p); }
        } else if (s[0] == 'v' && s[1] == 'n') {
            float3 n{};
            if (std::sscanf(s + 2, "%f %f %f", &n.x, &n.y, &n.z) == 3) { normals.emplace_back(n); }
        } else if (s[0] == 'v' && s[1] == 't') {
            float2 t{};
            if (std::sscanf(s + 2, "%f %f", &t.x, &t.y) >= 1) { uvs.emplace_back(t); }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<Corner> corners;
            const char *c = s + 1;
            while (*c != '\0') {
                while (*c == ' ' || *c == '\t' || *c == '\r') { c++; }
                if (*c == '\0') { break; }
                Corner corner{};
                char *end = nullptr;
                auto ip = static_cast<int>(std::strtol(c, &end, 10));
                if (end == c) { break; }
                corner.p = resolve_index(ip, positions.size());
                c = end;
                if (*c == '/') {
                    c++;
                    if (*c != '/') {
                        auto it = static_cast<int>(std::strtol(c, &end, 10));
                        corner.t = resolve_index(it, uvs.size());
                        c = end;
                    }
                    if (*c == '/') {
                        c++;
                        auto in = static_cast<int>(std::strtol(c, &end, 10));
                        corner.n = resolve_index(in, normals.size());
                        c = end;
                    }
                }
                corners.emplace_back(corner);
            }
            if (subdivision != 0u) {
                if (corners.size() >= 3u) { faces.emplace_back(std::move(corners)); }
            } else {
                for (size_t i = 2; i < corners.size(); i++) { faces.push_back({corners[0], corners[i - 1u], corners[i]}); }
            }
        }
    }
    if (positions.empty() || faces.empty()) { throw Error{"Failed to load mesh '" + path + "': no geometry."}; }
    auto has_uv = !drop_uv && !uvs.empty();
    auto has_file_normals = !drop_normal && !normals.empty();
    for (auto &f : faces) {
        for (auto &c : f) {
            if (c.p < 0 || c.p >= static_cast<int>(positions.size())) { throw Error{"Invalid vertex index in '" + path + "'."}; }
            if (c.t < 0 || c.t >= static_cast<int>(uvs.size())) { if (has_uv) { has_uv = false; } }
            if (c.n < 0 || c.n >= static_cast<int>(normals.size())) { has_file_normals = false; }
        }
    }

    // smooth-normal generation with a 45 degree limit when the file has no normals
    std::vector<std::vector<float3>> generated;
    auto generate = !drop_normal && !has_file_normals;
    if (generate) {
        std::vector<float3> face_normals(faces.size());
        std::vector<std::vector<uint32_t>> incident(positions.size());
        for (size_t i = 0; i < faces.size(); i++) {
            auto &f = faces[i];
            // (a polygon's normal from its first, second and LAST corner, as assimp's normal generation does; a triangle's as before)
            auto n = cross(positions[static_cast<size_t>(f[1].p)] - positions[static_cast<size_t>(f[0].p)],
                           positions[static_cast<size_t>(f.back().p)] - positions[static_cast<size_t>(f[0].p)]);
            auto len = length(n);
            face_normals[i] = len > 0.f ? n / len : float3{0.f, 0.f, 0.f};
            for (auto &c : f) { incident[static_cast<size_t>(c.p)].emplace_back(static_cast<uint32_t>(i)); }
        }
        auto limit = std::cos(radians(45.f));
        generated.resize(faces.size());
        for (size_t i = 0; i < faces.size(); i++) {
            generated[i].resize(faces[i].size());
            for (size_t k = 0; k < faces[i].size(); k++) {
                float3 sum{0.f, 0.f, 0.f};
                for (auto j : incident[static_cast<size_t>(faces[i][k].p)]) {
                    if (dot(face_normals[j], face_normals[i]) >= limit) { sum = sum + face_normals[j]; }
                }
                auto len = length(sum);
                generated[i][k] = len > 0.f ? sum / len : face_normals[i];
            }
        }
    }

    LoadedMesh mesh;
    auto has_normal = has_file_normals || generate;
    mesh.properties = (has_normal ? uint32_t{LR_SHAPE_HAS_VERTEX_NORMAL} : 0u) | (has_uv ? uint32_t{LR_SHAPE_HAS_VERTEX_UV} : 0u);
    using Key = std::tuple<int, int, int, int, int, int>;// position, uv, normal bits
    std::map<Key, uint32_t> joined;
    auto bits = [](float f) {
        int i;
        std::memcpy(&i, &f, sizeof(i));
        return i;
    };
    PolygonMesh polygons;
    polygons.face_offsets.emplace_back(0u);
    for (size_t i = 0; i < faces.size(); i++) {
        std::vector<uint32_t> idx(faces[i].size());
        for (size_t k = 0; k < faces[i].size(); k++) {
            auto &c = faces[i][k];
            lr_vertex v{};
            auto p = positions[static_cast<size_t>(c.p)];
            v.px = p.x, v.py = p.y, v.pz = p.z;
            float3 n{0.f, 0.f, 1.f};
            if (has_file_normals) { n = normalize(normals[static_cast<size_t>(c.n)]); }
            else if (generate) { n = generated[i][k]; }
            v.nx = n.x, v.ny = n.y, v.nz = n.z;
            if (has_uv) {
                auto t = uvs[static_cast<size_t>(c.t)];
                v.u = t.x, v.v = flip_uv ? t.y : 1.f - t.y;
            }
            Key key{c.p, has_uv ? c.t : -1, bits(v.nx), bits(v.ny), bits(v.nz), 0};
            auto it = joined.find(key);
            if (it == joined.end()) {
                it = joined.emplace(key, static_cast<uint32_t>(mesh.vertices.size())).first;
                mesh.vertices.emplace_back(v);
            }
            idx[k] = it->second;
        }
        if (subdivision == 0u) { mesh.triangles.push_back({idx[0], idx[1], idx[2]}); }
        else {
            polygons.indices.insert(polygons.indices.end(), idx.begin(), idx.end());
            polygons.face_offsets.emplace_back(static_cast<uint32_t>(polygons.indices.size()));
        }
    }
    if (subdivision != 0u) {// mesh.cpp:86-93: Assimp::Subdivider::Create(CATMULL_CLARKE)->Subdivide(mesh, out, level, true)
        polygons.vertices = std::move(mesh.vertices);
        auto polygon_count = polygons.face_offsets.size() - 1u;
        mesh = catmull_clark_subdivide(polygons, subdivision, mesh.properties);
        log_info("Subdivided '" + path + "' (Catmull-Clark, " + std::to_string(subdivision) + " levels): " + std::to_string(polygon_count) +
                 " polygons -> " + std::to_string(mesh.triangles.size()) + " triangles.");
        return mesh;
    }
    log_info("Loaded triangle mesh '" + path + "': " + std::to_string(mesh.vertices.size()) + " vertices, " +
             std::to_string(mesh.triangles.size()) + " triangles.");
    return mesh;
}

}// namespace lr
