// bvh_sim.cpp — CPU model of the megakernel's BVH4 walk (dev_trace.h: ordered near->far, one-triangle leaves,
// fp32 child boxes, or the packets' 8-bit planes with BVH_SIM_QUANTISED=1) used to compare host BVH builders offline: node visits / triangle tests per ray
// for camera rays, diffuse bounces and shadow rays of a seeded stand-in path workload.
//   g++ -O2 -std=c++17 -Iinclude tools/bvh_sim.cpp -Lluisarender_amd/lib -llrhost -Wl,-rpath,$PWD/luisarender_amd/lib -o /tmp/bvh_sim
//   /tmp/bvh_sim scene.luisa [pixels_per_axis=192] [bounces=6]
#include "lrhost.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct V { float x, y, z; };
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V norm(V a) { return a * (1.f / std::sqrt(dot(a, a))); }

struct Stats {
    uint64_t rays{0}, nodes{0}, tris{0}, empty{0}, max_stack{0};
    // round 5: what culling at pop time would save -- an entry is STALE when the entry distance it was pushed with no longer passes the slab
    // test's own criterion against the ray's current t_max (BVH_SIM_CULL=1 skips such entries; 0 only counts them)
    uint64_t pops{0}, stale_inner{0}, stale_leaf{0}, stale_chain[8]{};// stale_chain[k]: pops that skipped k stale entries in a row (k capped at 7)
    uint64_t depth_hist[64]{};// per ray: deepest stack
    uint64_t hits_hist[5]{};  // node visits by the number of children hit
};
static bool g_cull = false;
static int g_partial_sort = 0;
static int g_t_bits = 32;// the entry distance is kept with this many of its top bits (truncation rounds DOWN: conservative)

struct Hit { float t; uint32_t tri; float u, v; };

// optional model of the device's 64-byte packets: child boxes snapped outwards to the 8-bit grid of their node's own box
// (lrhip.hip: quantise_node); BVH_SIM_QUANTISED=1
static bool g_quantised = false;
static std::vector<lr_bvh4_node> g_qnodes;
static void quantise(const lr_accel &acc) {
    g_qnodes.assign(acc.nodes, acc.nodes + acc.node_count);
    for (auto &n : g_qnodes) {
        float *lo[3] = {n.lo_x, n.lo_y, n.lo_z}, *hi[3] = {n.hi_x, n.hi_y, n.hi_z};
        for (int a = 0; a < 3; a++) {
            float mn = 3e38f, mx = -3e38f;
            for (int c = 0; c < 4; c++) { if (n.child[c] != ~0u) { mn = std::min(mn, lo[a][c]), mx = std::max(mx, hi[a][c]); } }
            auto sc = (mx - mn) / 255.f;
            if (!(sc > 0.f)) { continue; }
            for (int c = 0; c < 4; c++) {
                if (n.child[c] == ~0u) { continue; }
                lo[a][c] = mn + std::floor((lo[a][c] - mn) / sc) * sc;
                hi[a][c] = mn + std::ceil((hi[a][c] - mn) / sc) * sc;
            }
        }
    }
}

static bool trace(const lr_accel &acc, V o, V d, float t_min, float t_max, bool any, Hit &hit, Stats &st) {
    V inv{1.f / d.x, 1.f / d.y, 1.f / d.z};
    uint32_t stack[256];
    float stack_t[256];
    uint32_t sp = 0, cur = 0, deepest = 0;
    // pop with the stale test: returns the next entry worth visiting (or ~0u)
    auto pop = [&]() -> uint32_t {
        st.pops++;
        int chain = 0;
        while (sp) {
            --sp;
            if (stack_t[sp] <= t_max * 1.0000004f) { st.stale_chain[std::min(chain, 7)]++; return stack[sp]; }
            ((stack[sp] & 0x80000000u) ? st.stale_leaf : st.stale_inner)++;
            if (!g_cull) { st.stale_chain[std::min(chain, 7)]++; return stack[sp]; }
            chain++;
        }
        st.stale_chain[std::min(chain, 7)]++;
        return ~0u;
    };
    hit.tri = ~0u;
    st.rays++;
    for (;;) {
        if (cur == ~0u) { st.depth_hist[std::min(deepest, 63u)]++; break; }
        if (cur & 0x80000000u) {
            auto &t = acc.triangles[cur & ((1u << 27u) - 1u)];
            st.tris++;
            V p0{t.v0[0], t.v0[1], t.v0[2]}, e1{t.e1[0], t.e1[1], t.e1[2]}, e2{t.e2[0], t.e2[1], t.e2[2]};
            auto pvec = cross(d, e2);
            auto det = dot(e1, pvec);
            auto inv_det = 1.f / det;
            auto tvec = o - p0;
            auto u = dot(tvec, pvec) * inv_det;
            auto qvec = cross(tvec, e1);
            auto v = dot(d, qvec) * inv_det;
            auto tt = dot(e2, qvec) * inv_det;
            if (det != 0.f && u >= 0.f && v >= 0.f && u + v <= 1.f && tt > t_min && tt < t_max && (t.flags & 1u)) {
                t_max = tt, hit.t = tt, hit.tri = cur & ((1u << 27u) - 1u), hit.u = u, hit.v = v;
                if (any) { st.depth_hist[std::min(deepest, 63u)]++; return true; }
            }
            cur = pop();
            continue;
        }
        auto &n = g_quantised ? g_qnodes[cur] : acc.nodes[cur];
        st.nodes++;
        uint32_t key[4];
        for (int i = 0; i < 4; i++) {
            auto t0x = (n.lo_x[i] - o.x) * inv.x, t1x = (n.hi_x[i] - o.x) * inv.x;
            auto t0y = (n.lo_y[i] - o.y) * inv.y, t1y = (n.hi_y[i] - o.y) * inv.y;
            auto t0z = (n.lo_z[i] - o.z) * inv.z, t1z = (n.hi_z[i] - o.z) * inv.z;
            auto tn = std::max(std::max(std::min(t0x, t1x), std::min(t0y, t1y)), std::max(std::min(t0z, t1z), t_min));
            auto tf = std::min(std::min(std::max(t0x, t1x), std::max(t0y, t1y)), std::min(std::max(t0z, t1z), t_max));
            auto h = tn <= tf * 1.0000004f && n.child[i] != ~0u;
            uint32_t bits;
            std::memcpy(&bits, &tn, 4);
            key[i] = h ? ((bits & ~3u) | static_cast<uint32_t>(i)) : ~0u;
        }
        if (g_partial_sort) {// (round 5 candidate: three comparators put the nearest child first and leave the rest as they fall)
            auto cs = [&](int a, int b) { if (key[a] > key[b]) { std::swap(key[a], key[b]); } };
            cs(0, 1), cs(2, 3), cs(0, 2);
            if (g_partial_sort >= 2) { cs(1, 3); }
        } else {
            std::sort(key, key + 4);
        }
        {
            int h = 0;
            for (int i = 0; i < 4; i++) { h += key[i] != ~0u; }
            st.hits_hist[h]++;
        }
        if (key[0] == ~0u) { st.empty++; }
        for (int i = 3; i >= 1; i--) {
            if (key[i] != ~0u) {
                auto tb = (key[i] & ~3u) & (g_t_bits >= 32 ? ~0u : ~0u << (32 - g_t_bits));
                std::memcpy(&stack_t[sp], &tb, 4);
                stack[sp++] = n.child[key[i] & 3u];
            }
        }
        st.max_stack = std::max<uint64_t>(st.max_stack, sp);
        deepest = std::max(deepest, sp);
        if (key[0] != ~0u) { cur = n.child[key[0] & 3u]; }
        else { cur = pop(); }
    }
    return hit.tri != ~0u;
}

static uint64_t rng_state = 0x853c49e6748fea9bull;
static float rnd() {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return static_cast<float>((rng_state >> 40) & 0xffffff) * (1.f / 16777216.f);
}

int main(int argc, char **argv) {
    if (argc < 2) { return 1; }
    auto n_px = argc > 2 ? std::atoi(argv[2]) : 192;
    auto bounces = argc > 3 ? std::atoi(argv[3]) : 6;
    lrhost_scene *scene = nullptr;
    lrhost_set_log_level(2);
    if (lrhost_scene_load_file(argv[1], nullptr, nullptr, 0, &scene) != 0) { std::fprintf(stderr, "%s\n", lrhost_last_error()); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    if (lrhost_scene_build_accel(scene) != 0) { std::fprintf(stderr, "%s\n", lrhost_last_error()); return 1; }
    auto build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    lr_scene s{};
    lrhost_scene_view(scene, 0, &s);
    auto &acc = s.accel;
    // emissive triangles (by instance light flag): approximate shadow-ray targets = random points on light instances' triangles
    std::vector<uint32_t> light_tris;
    {
        std::vector<char> is_light(s.instance_count, 0);
        for (uint32_t i = 0; i < s.light_instance_count; i++) { is_light[s.light_instances[i].instance_id] = 1; }
        for (uint32_t i = 0; i < acc.triangle_count; i++) { if (is_light[acc.triangles[i].inst]) { light_tris.push_back(i); } }
    }
    if (auto e = std::getenv("BVH_SIM_QUANTISED"); e != nullptr && std::atoi(e) != 0) { g_quantised = true, quantise(acc); }
    if (auto e = std::getenv("BVH_SIM_CULL"); e != nullptr) { g_cull = std::atoi(e) != 0; }
    if (auto e = std::getenv("BVH_SIM_T_BITS"); e != nullptr) { g_t_bits = std::atoi(e); }
    if (auto e = std::getenv("BVH_SIM_PARTIAL_SORT"); e != nullptr) { g_partial_sort = std::atoi(e); }
    Stats closest, shadow;
    auto &cam = s.camera;
    auto m = cam.camera_to_world;
    for (int py = 0; py < n_px; py++) {
        for (int px = 0; px < n_px; px++) {
            float fx = (px + rnd()) / n_px * cam.width, fy = (py + rnd()) / n_px * cam.height;
            float sx = (fx * 2.f - cam.width) * (cam.tan_half_fov / cam.height), sy = (fy * 2.f - cam.height) * (cam.tan_half_fov / cam.height);
            V dl = norm({sx, -sy, -1.f});
            V o{m[12], m[13], m[14]};
            V d = norm(V{m[0], m[1], m[2]} * dl.x + V{m[4], m[5], m[6]} * dl.y + V{m[8], m[9], m[10]} * dl.z);
            for (int b = 0; b < bounces; b++) {
                Hit h{};
                if (!trace(acc, o, d, 0.f, 1e30f, false, h, closest)) { break; }
                auto &t = acc.triangles[h.tri];
                V e1{t.e1[0], t.e1[1], t.e1[2]}, e2{t.e2[0], t.e2[1], t.e2[2]};
                V ng = norm(cross(e1, e2));
                if (dot(ng, d) > 0.f) { ng = ng * -1.f; }
                V p = o + d * h.t + ng * 1e-4f;
                if (!light_tris.empty()) {// shadow ray to a random light point
                    auto &lt = acc.triangles[light_tris[static_cast<size_t>(rnd() * light_tris.size()) % light_tris.size()]];
                    float a = rnd(), c = rnd();
                    if (a + c > 1.f) { a = 1.f - a, c = 1.f - c; }
                    V lp = V{lt.v0[0], lt.v0[1], lt.v0[2]} + V{lt.e1[0], lt.e1[1], lt.e1[2]} * a + V{lt.e2[0], lt.e2[1], lt.e2[2]} * c;
                    V sd = lp - p;
                    float dist = std::sqrt(dot(sd, sd));
                    Hit sh{};
                    trace(acc, p, sd * (1.f / dist), 0.f, dist * 0.9999f, true, sh, shadow);
                }
                // cosine-weighted bounce
                float u1 = rnd(), u2 = rnd();
                float r = std::sqrt(u1), phi = 6.2831853f * u2;
                V tx = std::fabs(ng.x) > 0.5f ? norm(cross(ng, {0, 1, 0})) : norm(cross(ng, {1, 0, 0}));
                V ty = cross(ng, tx);
                d = norm(tx * (r * std::cos(phi)) + ty * (r * std::sin(phi)) + ng * std::sqrt(std::max(0.f, 1.f - u1)));
                o = p;
            }
        }
    }
    auto per = [](uint64_t a, uint64_t b) { return b ? static_cast<double>(a) / static_cast<double>(b) : 0.0; };
    std::printf("build %.0f ms, nodes %u, tris %u | closest: rays %llu nodes/ray %.2f tris/ray %.2f empty %.1f%% | shadow: rays %llu nodes/ray %.2f tris/ray %.2f | "
                "total steps/ray %.2f max_stack %llu\n",
                build_ms, acc.node_count, acc.triangle_count, (unsigned long long)closest.rays, per(closest.nodes, closest.rays), per(closest.tris, closest.rays),
                100. * per(closest.empty, closest.nodes), (unsigned long long)shadow.rays, per(shadow.nodes, shadow.rays), per(shadow.tris, shadow.rays),
                per(closest.nodes + shadow.nodes, closest.rays + shadow.rays), (unsigned long long)std::max(closest.max_stack, shadow.max_stack));
    std::printf("closest pops %llu (%.2f / ray): stale inner %.1f%% of node visits, stale leaf %.1f%% of triangle tests (cull %d, t bits %d); entries skipped per pop 0..7+:",
                (unsigned long long)closest.pops, per(closest.pops, closest.rays), 100. * per(closest.stale_inner, closest.nodes), 100. * per(closest.stale_leaf, closest.tris), g_cull ? 1 : 0, g_t_bits);
    for (auto c : closest.stale_chain) { std::printf(" %.3f", per(c, closest.pops)); }
    std::printf("\nchildren hit per node visit 0..4 (closest | shadow):");
    for (int i = 0; i < 5; i++) { std::printf(" %.3f|%.3f", per(closest.hits_hist[i], closest.nodes), per(shadow.hits_hist[i], shadow.nodes)); }
    std::printf("\ndeepest stack per ray (closest | shadow), cumulative share of rays:");
    {
        uint64_t cc = 0, cs = 0;
        for (int i = 0; i < 40; i++) {
            cc += closest.depth_hist[i], cs += shadow.depth_hist[i];
            if (i >= 4 && i % 2 == 0) { std::printf(" %d: %.4f|%.4f", i, per(cc, closest.rays), per(cs, shadow.rays)); }
        }
    }
    std::printf("\n");
    lrhost_scene_destroy(scene);
    return 0;
}
