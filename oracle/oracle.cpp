// oracle.cpp — CPU restatement of the reference's megakernel path tracer.
// TEST INFRASTRUCTURE ONLY (see oracle.h): the parity checker and the CPU baseline, never
// linked into or called by the product path.
//
// Restated call stack (reference file:line):
//   ProgressiveIntegrator::Instance::_render_one_camera   src/base/integrator.cpp:51-113
//   MegakernelPathTracingInstance::Li                      src/integrators/mega_path.cpp:49-156
//   IndependentSamplerInstance                             src/samplers/independent.cpp:57-83
//   Camera::Instance::generate_ray / Pinhole / ThinLens    src/base/camera.cpp:212-224, src/cameras/*.cpp
//   Filter::Instance::sample                               src/base/filter.cpp:49-64
//   Geometry::interaction / shading_point                  src/base/geometry.cpp:281-389
//   Interaction::p_robust / spawn_ray / spawn_ray_to       src/base/interaction.cpp:13-30
//   UniformLightSamplerInstance                            src/lightsamplers/uniform.cpp:25-163
//   DiffuseLightClosure::_evaluate                         src/lights/diffuse.cpp:67-88
//   ColorFilmInstance::_accumulate / convert               src/films/color.cpp:87-130
#include "oracle.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "oracle_bsdf.h"
#include "oracle_bvh.h"

namespace oracle {

// LuisaCompute `offset_ray_origin` (absent submodule): the integer-ULP offset of Ray Tracing
// Gems ch. 6, restated from the published algorithm.  `n` is already scaled by the caller.
inline float3 offset_ray_origin(float3 p, float3 n) {
    constexpr auto origin = 1.0f / 32.0f;
    constexpr auto float_scale = 1.0f / 65536.0f;
    constexpr auto int_scale = 256.0f;
    int32_t of_i[3] = {static_cast<int32_t>(int_scale * n.x), static_cast<int32_t>(int_scale * n.y),
                       static_cast<int32_t>(int_scale * n.z)};
    float3 out;
    for (auto i = 0; i < 3; i++) {
        auto pi_bits = static_cast<int32_t>(float_bits(p[i]));
        auto p_i = bits_float(static_cast<uint32_t>(pi_bits + (p[i] < 0.f ? -of_i[i] : of_i[i])));
        out[i] = std::abs(p[i]) < origin ? p[i] + float_scale * n[i] : p_i;
    }
    return out;
}

// Samplers: Independent (independent.cpp:57-83), global Sobol (sobol.cpp:40-169), PaddedSobol
// (padded_sobol.cpp:23-150), and the PCG32 option of the north star (one PCG32 stream per path).
struct Sampler {
    const lr_sampler *cfg{nullptr};
    uint32_t kind{LR_SAMPLER_INDEPENDENT};
    uint32_t state{0u};
    PCG32 pcg;
    uint32_t px{0u}, py{0u}, sample_index{0u}, dimension{0u};
    uint64_t sobol_index{0u};

    static uint32_t reverse_bits(uint32_t v) {
        v = ((v >> 1u) & 0x55555555u) | ((v & 0x55555555u) << 1u);
        v = ((v >> 2u) & 0x33333333u) | ((v & 0x33333333u) << 2u);
        v = ((v >> 4u) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4u);
        v = ((v >> 8u) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8u);
        return (v >> 16u) | (v << 16u);
    }
    static uint32_t fast_owen_scramble(uint32_t seed, uint32_t v) {// sobol.cpp:40-48
        v = reverse_bits(v);
        v ^= v * 0x3d20adeau;
        v += seed;
        v *= (seed >> 16u) | 1u;
        v ^= v * 0x05526c56u;
        v ^= v * 0x53a22864u;
        return reverse_bits(v);
    }
    uint32_t sobol_bits(uint64_t a, uint32_t dim) const {// sobol.cpp:52-60
        auto v = 0u;
        auto i = dim * static_cast<uint32_t>(LR_SOBOL_MATRIX_SIZE);
        while (a != 0u) {
            if (a & 1u) { v ^= cfg->sobol_matrices[i]; }
            a >>= 1u;
            i++;
        }
        return v;
    }
    static uint32_t permutation_element(uint32_t i, uint32_t l, uint32_t p) {// padded_sobol.cpp:59-91
        auto w = l - 1u;
        w |= w >> 1u, w |= w >> 2u, w |= w >> 4u, w |= w >> 8u, w |= w >> 16u;
        do {
            i ^= p;
            i *= 0xe170893du;
            i ^= p >> 16u;
            i ^= (i & w) >> 4u;
            i ^= p >> 8u;
            i *= 0x0929eb3fu;
            i ^= p >> 23u;
            i ^= (i & w) >> 1u;
            i *= 1u | p >> 27u;
            i *= 0x6935fa69u;
            i ^= (i & w) >> 11u;
            i *= 0x74dcb303u;
            i ^= (i & w) >> 2u;
            i *= 0x9e501cc3u;
            i ^= (i & w) >> 2u;
            i *= 0xc860a3dfu;
            i &= w;
            i ^= i >> 5u;
        } while (i >= l);
        return (i + p) % l;
    }

    void start(const lr_sampler &s, uint32_t x, uint32_t y, uint32_t index, uint32_t width = 0u, uint32_t height = 0u) {
        cfg = &s;
        kind = s.kind;
        if (s.tile_size[0] != 0u) {// TileSharedSamplerInstance::start, tile_shared.cpp:51-62: the base is started with the pixel's tile
            if (s.tile_jitter != 0u) {
                auto offset = xxhash32(index);
                auto ox = static_cast<float>(offset >> 16u) * 0x1p-16f, oy = static_cast<float>(offset & 0xffffu) * 0x1p-16f;
                x += static_cast<uint32_t>(ox * static_cast<float>(width)) % width;
                y += static_cast<uint32_t>(oy * static_cast<float>(height)) % height;
            }
            x /= s.tile_size[0], y /= s.tile_size[1];
        }
        px = x, py = y, sample_index = index;
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:131-136 + _sobol_interval_to_index :67-96
            dimension = 2u;
            auto m = 0u;
            while ((1u << m) < s.scale) { m++; }
            if (m == 0u) {
                sobol_index = index;
            } else {
                auto frame = index;
                auto idx = static_cast<uint64_t>(frame) << (m << 1u);
                uint64_t delta = 0u;
                for (auto c = 0u; frame != 0u; frame >>= 1u, c++) {
                    if (frame & 1u) { delta ^= s.vdc_sobol[c]; }
                }
                auto b = delta ^ ((static_cast<uint64_t>(x) << m) | y);
                for (auto d = 0u; b != 0u; b >>= 1u, d++) {
                    if (b & 1u) { idx ^= s.vdc_sobol_inv[d]; }
                }
                sobol_index = idx;
            }
        } else if (kind == LR_SAMPLER_PADDED_SOBOL) {
            dimension = 0u;
        } else {
            state = xxhash32(x, y, s.seed, index);
            if (kind == LR_SAMPLER_PCG32) { pcg = PCG32{static_cast<uint64_t>(state)}; }
        }
    }
    float generate_1d() {
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:147-153
            dimension = dimension >= static_cast<uint32_t>(LR_SOBOL_DIMENSIONS) ? 2u : dimension;
            auto hash = xxhash32(dimension, cfg->seed);
            auto u = static_cast<float>(fast_owen_scramble(hash, sobol_bits(sobol_index, dimension))) * 0x1p-32f;
            dimension += 1u;
            return clampf(u, 0.f, one_minus_epsilon);
        }
        if (kind == LR_SAMPLER_PADDED_SOBOL) {// padded_sobol.cpp:127-136
            auto hash = xxhash32(px, py, sample_index ^ cfg->seed, dimension);
            auto index = permutation_element(sample_index, cfg->spp, hash);
            auto u = std::min(static_cast<float>(fast_owen_scramble(hash, sobol_bits(index, 0u))) * 0x1p-32f, one_minus_epsilon);
            dimension += 1u;
            return u;
        }
        return kind == LR_SAMPLER_PCG32 ? pcg.uniform_float() : lcg(state);
    }
    float2 generate_2d() {
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:154-162
            dimension = dimension + 1u >= static_cast<uint32_t>(LR_SOBOL_DIMENSIONS) ? 2u : dimension;
            auto hx = xxhash32(dimension, cfg->seed), hy = xxhash32(dimension + 1u, cfg->seed);
            auto ux = static_cast<float>(fast_owen_scramble(hx, sobol_bits(sobol_index, dimension))) * 0x1p-32f;
            auto uy = static_cast<float>(fast_owen_scramble(hy, sobol_bits(sobol_index, dimension + 1u))) * 0x1p-32f;
            dimension += 2u;
            return {clampf(ux, 0.f, one_minus_epsilon), clampf(uy, 0.f, one_minus_epsilon)};
        }
        if (kind == LR_SAMPLER_PADDED_SOBOL) {// padded_sobol.cpp:137-149
            auto hx = xxhash32(px, py, sample_index ^ cfg->seed, dimension);
            auto hy = xxhash32(px, py, sample_index ^ cfg->seed, dimension + 1u);
            auto index = permutation_element(sample_index, cfg->spp, hx);
            float2 u;
            u.x = std::min(static_cast<float>(fast_owen_scramble(hx, sobol_bits(index, 0u))) * 0x1p-32f, one_minus_epsilon);
            u.y = std::min(static_cast<float>(fast_owen_scramble(hy, sobol_bits(index, 1u))) * 0x1p-32f, one_minus_epsilon);
            dimension += 2u;
            return u;
        }
        float2 u;
        u.x = generate_1d();
        u.y = generate_1d();
        return u;
    }
    float2 generate_pixel_2d() {
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:163-169: unscrambled dimensions 0 / 1 of the global sequence
            auto ux = static_cast<float>(sobol_bits(sobol_index, 0u)) * 0x1p-32f;
            auto uy = static_cast<float>(sobol_bits(sobol_index, 1u)) * 0x1p-32f;
            auto s = static_cast<float>(cfg->scale);
            return {clampf(ux * s - static_cast<float>(px), 0.f, one_minus_epsilon),
                    clampf(uy * s - static_cast<float>(py), 0.f, one_minus_epsilon)};
        }
        return generate_2d();// sampler.h:48
    }
};

struct FilterSample {
    float2 offset;
    float weight;
};
inline FilterSample filter_sample(const lr_filter &f, float2 u) {// filter.cpp:49-64
    constexpr auto n = static_cast<uint32_t>(LR_FILTER_LUT_SIZE) - 1u;
    auto prob = [&](uint32_t i) { return f.alias_prob[i]; };
    auto alias = [&](uint32_t i) { return f.alias_index[i]; };
    auto sy = sample_alias_table(prob, alias, n, u.x);// note the x/y swap of the reference
    auto sx = sample_alias_table(prob, alias, n, u.y);
    auto pdf = f.pdf[sy.index] * f.pdf[sx.index];
    auto fv = lerp(f.lut[sx.index], f.lut[sx.index + 1u], sx.u) * lerp(f.lut[sy.index], f.lut[sy.index + 1u], sy.u);
    float2 p{static_cast<float>(sx.index) + sx.u, static_cast<float>(sy.index) + sy.u};
    auto inv_size = 1.0f / static_cast<float>(LR_FILTER_LUT_SIZE);
    float2 pixel{(p.x * inv_size * 2.0f - 1.0f) * f.radius, (p.y * inv_size * 2.0f - 1.0f) * f.radius};
    return {{pixel.x + f.shift[0], pixel.y + f.shift[1]}, fv / pdf};
}

struct CameraSample {
    Ray ray;
    float weight;
};
inline CameraSample generate_camera_ray(const lr_scene &scene, uint32_t px, uint32_t py, float2 u_filter, float2 u_lens) {
    auto &cam = scene.camera;
    auto fs = filter_sample(scene.filter, u_filter);
    float2 pixel{static_cast<float>(px) + .5f + fs.offset.x, static_cast<float>(py) + .5f + fs.offset.y};// camera.cpp:215
    float2 res{static_cast<float>(cam.width), static_cast<float>(cam.height)};
    float3 o{0.f, 0.f, 0.f}, d;
    if (cam.kind == LR_CAMERA_PINHOLE) {// pinhole.cpp:60-67
        float2 p{(pixel.x * 2.0f - res.x) * (cam.tan_half_fov / res.y), (pixel.y * 2.0f - res.y) * (cam.tan_half_fov / res.y)};
        d = normalize(f3(p.x, -p.y, -1.f));
    } else if (cam.kind == LR_CAMERA_THIN_LENS) {// thin_lens.cpp:91-101
        float2 pixel_offset{.5f * res.x, .5f * res.y};
        float2 coord_focal{(pixel.x - pixel_offset.x) * cam.projected_pixel_size, (pixel.y - pixel_offset.y) * cam.projected_pixel_size};
        auto p_focal = f3(coord_focal.x, -coord_focal.y, -cam.focus_distance);
        auto disk = sample_uniform_disk_concentric(u_lens);
        auto p_lens = f3(disk.x * cam.lens_radius, disk.y * cam.lens_radius, 0.f);
        o = p_lens;
        d = normalize(p_focal - p_lens);
    } else {// ortho.cpp:52-58
        float2 p{(pixel.x * 2.0f - res.x) / res.y * cam.ortho_scale, (pixel.y * 2.0f - res.y) / res.y * cam.ortho_scale};
        o = f3(p.x, -p.y, 0.f);
        d = f3(0.f, 0.f, -1.f);
    }
    // ClipPlaneCameraWrapper, camera.h:147-157
    auto cos_axis = dot(d, f3(0.f, 0.f, -1.f));
    auto t_min = cam.clip_near / cos_axis, t_max = cam.clip_far / cos_axis;
    auto m = cam.camera_to_world;// camera.cpp:218-222
    auto ow = f3(m[0] * o.x + m[4] * o.y + m[8] * o.z + m[12], m[1] * o.x + m[5] * o.y + m[9] * o.z + m[13],
                 m[2] * o.x + m[6] * o.y + m[10] * o.z + m[14]);
    auto dw = normalize(f3(m[0], m[1], m[2]) * d.x + f3(m[4], m[5], m[6]) * d.y + f3(m[8], m[9], m[10]) * d.z);
    return {{ow, t_min, dw, t_max}, 1.f * fs.weight};
}

}// namespace oracle

using namespace oracle;

struct oracle_ctx {
    const lr_scene *scene;
    Accel accel;
    float shutter_weight{1.f};// Camera::ShutterSample::point.weight of the samples being rendered (integrator.cpp:74)
    explicit oracle_ctx(const lr_scene *s) : scene{s}, accel{*s} {
        if (s->any_non_opaque) {// Geometry::trace_closest / trace_any take the ray-query branch (geometry.cpp:219,264)
            accel.alpha_skip = [this](uint32_t inst, uint32_t prim, float u, float v) { return alpha_skip(inst, prim, u, v); };
        }
    }

    // Surface::Instance::evaluate_opacity: OpacitySurfaceWrapper (surface.h:183-189), MixSurfaceInstance (mix.cpp:63-70)
    float surface_opacity(uint32_t tag, float2 uv) const {
        auto one = [&](uint32_t t) {
            auto alpha_tex = scene->surfaces[t].alpha_tex;
            return alpha_tex >= 0 ? texture_evaluate(*scene, alpha_tex, uv).x : 1.f;
        };
        auto &s = scene->surfaces[tag];
        if (s.kind != LR_SURFACE_MIX) { return one(tag); }
        return surface_opacity(s.u[0], uv) * surface_opacity(s.u[1], uv);// (Mix children may be Mix surfaces themselves)
    }
    // Geometry::_alpha_skip, geometry.cpp:165-192
    bool alpha_skip(uint32_t inst_id, uint32_t prim, float u, float v) const {
        auto it = make_interaction(inst_id, prim, f3(1.f - u - v, u, v), true, f3(0.f, 0.f, 1.f));
        if (!(it.flags() & LR_SHAPE_MAYBE_NON_OPAQUE) || !(it.flags() & LR_SHAPE_HAS_SURFACE)) { return false; }
        uint32_t ub, vb;
        std::memcpy(&ub, &u, 4), std::memcpy(&vb, &v, 4);
        auto xi = static_cast<float>(xxhash32(inst_id, prim, ub, vb)) * 0x1p-32f;
        return xi > surface_opacity(it.surface_tag(), it.uv);
    }

    // Geometry::shading_point + Interaction ctor, geometry.cpp:345-389, interaction.h:94-99
    Interaction make_interaction(uint32_t inst_id, uint32_t prim_id, float3 bary, bool use_wo, float3 wo_or_pfrom) const {
        auto &s = *scene;
        auto &inst = s.instances[inst_id];
        Interaction it;
        it.handle = inst.handle;
        it.inst = inst_id, it.prim = prim_id;
        auto &mesh = s.meshes[it.mesh_index()];
        auto tri = s.triangles[mesh.triangle_offset + prim_id];
        auto &v0 = s.vertices[mesh.vertex_offset + tri.i0];
        auto &v1 = s.vertices[mesh.vertex_offset + tri.i1];
        auto &v2 = s.vertices[mesh.vertex_offset + tri.i2];
        auto interp3 = [&](float3 a, float3 b, float3 c) { return bary.x * a + bary.y * b + bary.z * c; };
        auto p0 = f3(v0.px, v0.py, v0.pz), p1 = f3(v1.px, v1.py, v1.pz), p2 = f3(v2.px, v2.py, v2.pz);
        auto ns_local = interp3(f3(v0.nx, v0.ny, v0.nz), f3(v1.nx, v1.ny, v1.nz), f3(v2.nx, v2.ny, v2.nz));
        float2 uv0{v0.u, v0.v}, uv1{v1.u, v1.v}, uv2{v2.u, v2.v};
        auto duv0 = uv1 - uv0, duv1 = uv2 - uv0;
        auto det = duv0.x * duv1.y - duv0.y * duv1.x;
        auto inv_det = 1.f / det;
        auto dp0 = p1 - p0, dp1 = p2 - p0;
        auto dpdu_local = (dp0 * duv1.y - dp1 * duv0.y) * inv_det;
        auto o2w = inst.object_to_world;
        mat3 m{{f3(o2w[0], o2w[1], o2w[2]), f3(o2w[4], o2w[5], o2w[6]), f3(o2w[8], o2w[9], o2w[10])}};
        auto t = f3(o2w[12], o2w[13], o2w[14]);
        auto p = m * interp3(p0, p1, p2) + t;
        auto c = cross(m * dp0, m * dp1);
        it.area = length(c) * .5f;
        auto ng = normalize(c);
        auto fallback = Frame::make(ng);
        auto dpdu = det == 0.f ? fallback.s : m * dpdu_local;
        auto mn = transpose(inverse(m));
        auto has_normal = (it.flags() & LR_SHAPE_HAS_VERTEX_NORMAL) != 0u;
        auto has_uv = (it.flags() & LR_SHAPE_HAS_VERTEX_UV) != 0u;
        auto ns = has_normal ? normalize(mn * ns_local) : ng;
        it.uv = has_uv ? float2{bary.x * uv0.x + bary.y * uv1.x + bary.z * uv2.x, bary.x * uv0.y + bary.y * uv1.y + bary.z * uv2.y} :
                         float2{bary.y, bary.z};
        // BAKED-GEOMETRY MODE (oracle_bvh.h: Accel::set_bake; test infrastructure): a traced hit is reconstructed from the fp32
        // world-space triangle the device intersected -- point, geometric normal, area, tangent and the interpolation of the
        // world-space vertex normals in the device's own order (csrc/hip/dev_shade.h: reconstruct_baked, lrhip.hip: build_shade_tris)
        if (const auto bt = use_wo ? accel.baked(inst_id, prim_id) : nullptr; bt != nullptr) {
            auto b0 = f3(bt->v0[0], bt->v0[1], bt->v0[2]), e1 = f3(bt->e1[0], bt->e1[1], bt->e1[2]), e2 = f3(bt->e2[0], bt->e2[1], bt->e2[2]);
            p = b0 + e1 * bary.y + e2 * bary.z;
            c = cross(e1, e2);
            it.area = length(c) * .5f;
            ng = normalize(c);
            fallback = Frame::make(ng);
            dpdu = det == 0.f ? fallback.s : (e1 * duv1.y - e2 * duv0.y) * inv_det;
            auto nw = [&](const lr_vertex &v) { return mn.c[0] * v.nx + mn.c[1] * v.ny + mn.c[2] * v.nz; };
            ns = has_normal ? normalize(nw(v0) * bary.x + nw(v1) * bary.y + nw(v2) * bary.z) : ng;
        }
        it.pg = p, it.ps = p, it.ng = ng;
        it.shading = Frame::make(face_forward(ns, ng), dpdu);
        it.back_facing = use_wo ? dot(wo_or_pfrom, ng) < 0.0f :          // geometry.cpp:290
                                  dot(ng, wo_or_pfrom - p) < 0.f;          // uniform.cpp:122
        return it;
    }

    // Interaction::p_robust / spawn_ray / spawn_ray_to, interaction.cpp:13-30
    static float3 p_robust(const Interaction &it, float3 w) {
        auto front = dot(it.shading.n, w) > 0.f;
        auto n = front ? it.ng : -it.ng;
        return offset_ray_origin(it.pg, it.intersection_offset_factor() * n);
    }
    static Ray spawn_ray(const Interaction &it, float3 wi) {
        return {p_robust(it, wi), 0.f, wi, std::numeric_limits<float>::max()};
    }
    static Ray spawn_ray_to(const Interaction &it, float3 p) {
        auto p_from = p_robust(it, p - it.pg);
        auto L = p - p_from;
        auto d = length(L);
        return {p_from, 0.f, L * (1.f / d), d * .9999f};
    }

    struct LightEval {
        Spectrum3 L{0.f, 0.f, 0.f};
        float pdf{0.f};
    };
    // DiffuseLightClosure::_evaluate, diffuse.cpp:67-88
    LightEval light_evaluate(const Interaction &it_light, float3 p_from) const {
        auto &s = *scene;
        auto &light = s.lights[it_light.light_tag()];
        auto &mesh = s.meshes[it_light.mesh_index()];
        auto pdf_triangle = s.tri_pdf[mesh.triangle_offset + it_light.prim];
        auto pdf_area = pdf_triangle / it_light.area;
        auto cos_wo = abs_dot(normalize(p_from - it_light.pg), it_light.ng);
        auto L = illuminant(s, light.emission_tex, it_light.uv).value * light.scale;
        auto diff = it_light.pg - p_from;
        auto pdf = dot(diff, diff) * pdf_area * (1.0f / cos_wo);
        auto invalid = std::abs(cos_wo) < 1e-6f || (!light.two_sided && it_light.back_facing);
        return {invalid ? f3(0.f) : L, invalid ? 0.f : pdf};
    }

    // ---- Environment::Instance::evaluate / sample: spherical.cpp:88-141 (constant emission -> uniform sphere;
    // image emission -> 2048 x 1024 alias/pdf tables), directional.cpp:62-98
    static float3 mul3(const float m[9], float3 v) {// column-major 3x3
        return f3(m[0], m[1], m[2]) * v.x + f3(m[3], m[4], m[5]) * v.y + f3(m[6], m[7], m[8]) * v.z;
    }
    static float directional_pdf(float p, float theta) {// spherical.cpp:76-80
        auto sn = std::sin(theta);
        auto inv_s = sn > 0.f ? 1.f / sn : 0.f;
        return p * inv_s * (.5f * inv_pi * inv_pi);
    }
    static bool env_is_image(const lr_environment &env) { return env.kind == LR_ENV_SPHERICAL && env.map_width != 0u; }
    LightEval env_directional(const lr_environment &env, float3 wi_local) const {// DirectionalInstance::_evaluate
        auto L = illuminant(*scene, env.emission_tex, {.5f, .5f}).value;
        auto pdf = 1.f / (2.f * pi * (1.f - env.cos_half_angle));// uniform_cone_pdf, sampling.cpp:119-121
        auto valid = env.cos_half_angle < cos_theta(wi_local);
        return {L * (valid ? env.scale : 0.f), valid ? pdf : 0.f};
    }
    LightEval env_evaluate(float3 wi) const { return env_evaluate(scene->environment, wi); }
    LightEval env_evaluate(const lr_environment &env, float3 wi) const {
        if (env.kind == LR_ENV_COMBINED) {// CombinedInstance::evaluate, combined.cpp:57-78 (both children live)
            auto wi_local = normalize(mul3(env.world_to_env, wi));
            auto a = env_evaluate(scene->environment_children[env.child[0]], wi_local), b = env_evaluate(scene->environment_children[env.child[1]], wi_local);
            auto sa = env.child_scale[0], sb = env.child_scale[1];
            auto t = sb / (sa + sb);
            return {a.L * sa + b.L * sb, lerp(a.pdf, b.pdf, t)};
        }
        if (env.kind == LR_ENV_DIRECTIONAL) {
            if (!env.visible) { return {}; }
            auto frame = Frame::make(f3(env.direction[0], env.direction[1], env.direction[2]));
            return env_directional(env, normalize(frame.world_to_local(mul3(env.world_to_env, wi))));
        }
        auto w = normalize(mul3(env.world_to_env, wi));
        auto theta = std::acos(w.y), phi = std::atan2(w.x, w.z);// Spherical::direction_to_uv, spherical.cpp:51-57
        float2 uv{fract(1.f - 0.5f * inv_pi * phi), fract(theta * inv_pi)};
        auto L = illuminant(*scene, env.emission_tex, uv).value * env.scale;
        if (!env_is_image(env)) { return {L, uniform_sphere_pdf}; }
        auto sx = static_cast<float>(env.map_width), sy = static_cast<float>(env.map_height);
        auto ix = static_cast<uint32_t>(clampf(uv.x * sx, 0.f, sx - 1.f)), iy = static_cast<uint32_t>(clampf(uv.y * sy, 0.f, sy - 1.f));
        return {L, directional_pdf(env.pdf[iy * env.map_width + ix], theta)};
    }
    struct EnvSample {
        LightEval eval;
        float3 wi;
    };
    EnvSample env_sample(float2 u) const { return env_sample(scene->environment, u); }
    EnvSample env_sample(const lr_environment &env, float2 u) const {
        if (env.kind == LR_ENV_COMBINED) {// CombinedInstance::sample, combined.cpp:80-111
            auto &ca = scene->environment_children[env.child[0]], &cb = scene->environment_children[env.child[1]];// (either may be a Combined node)
            auto sa = env.child_scale[0], sb = env.child_scale[1];
            auto weight_a = sa / (sa + sb);
            EnvSample s;
            if (u.x < weight_a) {
                u.x = u.x / weight_a;
                s = env_sample(ca, u);
                auto eb = env_evaluate(cb, s.wi);
                s.eval.L = s.eval.L * sa + eb.L * sb;
                s.eval.pdf = lerp(s.eval.pdf, eb.pdf, 1.f - weight_a);
            } else {
                u.x = (u.x - weight_a) / (1.f - weight_a);
                s = env_sample(cb, u);
                auto ea = env_evaluate(ca, s.wi);
                s.eval.L = ea.L * sa + s.eval.L * sb;
                s.eval.pdf = lerp(ea.pdf, s.eval.pdf, 1.f - weight_a);
            }
            s.wi = normalize(mul3(env.env_to_world, s.wi));
            return s;
        }
        if (env.kind == LR_ENV_DIRECTIONAL) {
            auto cos_t = (1.f - u.x) + u.x * env.cos_half_angle;// sample_uniform_cone, sampling.cpp:123-131
            auto sin_t = std::sqrt(std::max(1.f - cos_t * cos_t, 0.f));
            auto phi = 2.f * pi * u.y;
            auto wi_local = f3(sin_t * std::cos(phi), sin_t * std::sin(phi), cos_t);
            auto frame = Frame::make(f3(env.direction[0], env.direction[1], env.direction[2]));
            return {env_directional(env, wi_local), normalize(mul3(env.env_to_world, frame.local_to_world(wi_local)))};
        }
        if (!env_is_image(env)) {
            auto w = sample_uniform_sphere(u);
            auto L = illuminant(*scene, env.emission_tex, {0.f, 0.f}).value * env.scale;
            return {{L, uniform_sphere_pdf}, normalize(mul3(env.env_to_world, w))};
        }
        auto W = env.map_width, H = env.map_height;
        auto sy = sample_alias_table([&](uint32_t i) { return env.alias[i].prob; }, [&](uint32_t i) { return env.alias[i].alias; }, H, u.y);
        auto row = env.alias + H + sy.index * W;
        auto sx = sample_alias_table([&](uint32_t i) { return row[i].prob; }, [&](uint32_t i) { return row[i].alias; }, W, u.x);
        float2 uv{(static_cast<float>(sx.index) + sx.u) / static_cast<float>(W), (static_cast<float>(sy.index) + sy.u) / static_cast<float>(H)};
        auto p = env.pdf[sy.index * W + sx.index];
        auto phi = 2.f * pi * (1.f - uv.x), theta = pi * uv.y;// Spherical::uv_to_direction, spherical.cpp:42-50
        auto sin_theta = std::sin(theta);
        auto w = normalize(f3(std::sin(phi) * sin_theta, std::cos(theta), std::cos(phi) * sin_theta));
        auto L = illuminant(*scene, env.emission_tex, uv).value * env.scale;
        return {{L, directional_pdf(p, theta)}, normalize(mul3(env.env_to_world, w))};
    }

    struct LightSample {
        LightEval eval;
        Ray shadow_ray{{0.f, 0.f, 0.f}, 0.f, {0.f, 0.f, 0.f}, 0.f};
    };
    // LightSampler::Instance::sample, light_sampler.cpp:57-63 with UniformLightSamplerInstance::select / _sample_*
    LightSample light_sample(const Interaction &it, float u_sel, float2 u_light) const {
        auto &s = *scene;
        auto env_prob = s.integrator.env_prob;
        auto n = static_cast<float>(s.integrator.light_count);
        auto is_env = false;
        auto tag = 0u;
        auto prob = 0.f;
        if (env_prob == 1.f) {
            is_env = true, prob = 1.f;
        } else if (env_prob == 0.f) {
            tag = static_cast<uint32_t>(clampf(u_sel * n, 0.f, n - 1.f)), prob = 1.f / n;
        } else {
            auto uu = (u_sel - env_prob) / (1.f - env_prob);
            tag = static_cast<uint32_t>(clampf(uu * n, 0.f, n - 1.f));
            is_env = u_sel < env_prob;
            prob = is_env ? env_prob : (1.f - env_prob) / n;
        }
        LightSample out;
        if (is_env) {// _sample_environment, uniform.cpp:125-137
            auto es = env_sample(u_light);
            out.eval = {es.eval.L, es.eval.pdf * prob};
            out.shadow_ray = spawn_ray(it, es.wi);
            return out;
        }
        // _sample_area, uniform.cpp:107-123
        auto handle = s.light_instances[tag];
        auto &light_inst = s.instances[handle.instance_id];
        auto &mesh = s.meshes[light_inst.handle.x >> 10u];
        auto table = s.tri_alias + mesh.triangle_offset;
        auto as = sample_alias_table([&](uint32_t i) { return table[i].prob; }, [&](uint32_t i) { return table[i].alias; },
                                     light_inst.handle.z, u_light.x);
        auto uvw = sample_uniform_triangle({as.u, u_light.y});
        auto it_light = make_interaction(handle.instance_id, as.index, uvw, false, it.pg);
        auto eval = light_evaluate(it_light, it.ps);// uniform.cpp:131-134
        eval.pdf *= prob;                            // light_sampler.cpp:83
        out.eval = eval;
        out.shadow_ray = spawn_ray_to(it, it_light.pg);
        return out;
    }

    struct PathStats {
        TraceCounters trace;
        uint64_t closest{0}, shadow{0}, hits{0}, nee{0}, bounces{0};
    };

    // NormalVisualizerInstance::Li, normal.cpp:36-70
    float3 Li_normal(uint32_t px, uint32_t py, uint32_t sample_index, PathStats &stats) const {
        auto &s = *scene;
        Sampler sampler;
        sampler.start(s.sampler, px, py, sample_index, s.camera.width, s.camera.height);
        auto u_filter = sampler.generate_pixel_2d();
        auto u_lens = s.camera.kind == LR_CAMERA_THIN_LENS ? sampler.generate_2d() : float2{.5f, .5f};
        auto cs = generate_camera_ray(s, px, py, u_filter, u_lens);
        (void)sampler.generate_1d();// spectrum()->sample(generate_1d()), :42 (sRGB ignores it)
        stats.closest++;
        auto hit = accel.trace(cs.ray, false, stats.trace);
        auto ns = f3(0.f);
        auto wo = -cs.ray.d;
        if (!hit.miss()) {
            stats.hits++;
            auto it = make_interaction(hit.inst, hit.prim, f3(1.f - hit.bary.x - hit.bary.y, hit.bary.x, hit.bary.y), true, wo);
            if (s.integrator.flags & LR_NORMAL_SHADING) {
                ns = it.has_surface() ? Closure::populate(s, it, wo, 1.f).shading.n : it.shading.n;
            } else {
                ns = it.ng;
            }
            if (s.integrator.flags & LR_NORMAL_REMAP) { ns = ns * .5f + f3(.5f); }
        }
        return f3(cs.weight) * ns;
    }

    // DirectLightingInstance::Li, direct.cpp:66-200: the loop body runs once (alpha_skip is never set)
    float3 Li_direct(uint32_t px, uint32_t py, uint32_t sample_index, PathStats &stats) const {
        auto &s = *scene;
        Sampler sampler;
        sampler.start(s.sampler, px, py, sample_index, s.camera.width, s.camera.height);
        auto u_filter = sampler.generate_pixel_2d();
        auto u_lens = s.camera.kind == LR_CAMERA_THIN_LENS ? sampler.generate_2d() : float2{.5f, .5f};
        auto cs = generate_camera_ray(s, px, py, u_filter, u_lens);// (fixed sRGB spectrum: no wavelength draw, :72)
        const auto samples_lights = (s.integrator.flags & LR_DIRECT_SAMPLE_LIGHTS) != 0u;
        const auto samples_surfaces = (s.integrator.flags & LR_DIRECT_SAMPLE_SURFACES) != 0u;
        const auto has_env = s.environment.kind != LR_ENV_NONE;
        const auto weight = f3(cs.weight);
        Spectrum3 Li = f3(0.f);
        auto ray = cs.ray;
        auto wo = -ray.d;
        stats.closest++;
        auto hit = accel.trace(ray, false, stats.trace);
        if (hit.miss()) {
            if (has_env) { Li += weight * env_evaluate(ray.d).L; }// :92-98
            return Li;
        }
        stats.hits++;
        auto it = make_interaction(hit.inst, hit.prim, f3(1.f - hit.bary.x - hit.bary.y, hit.bary.x, hit.bary.y), true, wo);
        if (s.light_count != 0u && it.has_light()) { Li += weight * light_evaluate(it, ray.o).L; }// :101-106
        if (!it.has_surface()) { return Li; }
        stats.bounces++;
        LightSample light_sample{};
        auto occluded = false;
        if (samples_lights) {// :114-127
            auto u_light_selection = sampler.generate_1d();
            auto u_light_surface = sampler.generate_2d();
            stats.nee++;
            light_sample = this->light_sample(it, u_light_selection, u_light_surface);
            auto &L = light_sample.eval.L;
            if (light_sample.eval.pdf > 0.f && (L.x > 0.f || L.y > 0.f || L.z > 0.f)) {
                stats.shadow++;
                occluded = !accel.trace(light_sample.shadow_ray, true, stats.trace).miss();
            }
        }
        auto u_lobe = sampler.generate_1d();
        auto u_bsdf = samples_surfaces ? sampler.generate_2d() : float2{0.f, 0.f};
        auto closure = Closure::populate(s, it, wo, 1.f);
        if (samples_lights && light_sample.eval.pdf > 0.0f && !occluded) {// :146-157
            auto eval = closure.evaluate(wo, light_sample.shadow_ray.d);
            if (eval.pdf > 0.f) {
                auto w = samples_surfaces ? balance_heuristic(light_sample.eval.pdf, eval.pdf) : 1.f;
                Li += w * weight * eval.f * light_sample.eval.L / light_sample.eval.pdf;
            }
        }
        if (samples_surfaces) {// :159-192
            auto surface_sample = closure.sample(wo, u_lobe, u_bsdf);
            ray = spawn_ray(it, surface_sample.wi);
            stats.closest++;
            auto bsdf_hit = accel.trace(ray, false, stats.trace);
            Spectrum3 light_L = f3(0.f);
            auto light_pdf = 0.f;
            if (bsdf_hit.miss()) {
                if (has_env) {// evaluate_miss, uniform.cpp:67-76
                    auto eval = env_evaluate(ray.d);
                    light_L = eval.L, light_pdf = eval.pdf * s.integrator.env_prob;
                }
            } else {
                stats.hits++;
                auto bsdf_it = make_interaction(bsdf_hit.inst, bsdf_hit.prim, f3(1.f - bsdf_hit.bary.x - bsdf_hit.bary.y, bsdf_hit.bary.x, bsdf_hit.bary.y), true, -ray.d);
                if (s.light_count != 0u && bsdf_it.has_light()) {// evaluate_hit, uniform.cpp:50-65
                    auto eval = light_evaluate(bsdf_it, ray.o);
                    light_L = eval.L, light_pdf = eval.pdf * (1.f - s.integrator.env_prob) / static_cast<float>(s.integrator.light_count);
                }
            }
            if (light_pdf > 0.f && surface_sample.eval.pdf > 0.f) {
                auto w = samples_lights ? balance_heuristic(surface_sample.eval.pdf, light_pdf) : 1.f;
                Li += weight * w * surface_sample.eval.f * light_L / surface_sample.eval.pdf;
            }
        }
        return Li;
    }

    // ---------------------------------------------------------------- volumetric megakernel
    // MediumTracker, src/util/medium_tracker.{h,cpp}: media the ray is inside of, sorted by ascending priority
    struct MediumInfo {
        uint32_t priority{LR_MEDIUM_VACUUM_PRIORITY}, medium_tag{LR_INVALID_ID};
        bool operator==(const MediumInfo &o) const { return priority == o.priority && medium_tag == o.medium_tag; }
    };
    struct MediumTracker {
        static constexpr auto capacity = 32u;
        uint32_t priority_list[capacity];
        MediumInfo medium_list[capacity];
        uint32_t size{0u};
        MediumTracker() {
            for (auto &p : priority_list) { p = LR_MEDIUM_VACUUM_PRIORITY; }
        }
        bool vacuum() const { return priority_list[0] == LR_MEDIUM_VACUUM_PRIORITY; }
        bool true_hit(uint32_t priority) const { return priority <= priority_list[0]; }// medium_tracker.cpp:20-22
        MediumInfo current() const { return vacuum() ? MediumInfo{} : medium_list[0]; }
        void enter(uint32_t priority, MediumInfo value) {// :24-45 (overflow: error + no-op)
            if (size == capacity) { return; }
            size++;
            auto x = priority;
            auto v = value;
            for (auto i = 0u; i < capacity; i++) {
                auto p = priority_list[i];
                auto m = medium_list[i];
                auto should_swap = p > x;
                priority_list[i] = should_swap ? x : p;
                medium_list[i] = should_swap ? v : m;
                x = should_swap ? p : x;
                v = should_swap ? m : v;
            }
        }
        void exit(uint32_t priority, MediumInfo value) {// :47-65 (a nonexistent entry: error + no-op)
            auto remove_num = 0u;
            for (auto i = 0u; i < capacity - 1u; i++) {
                auto should_remove = priority_list[i] == priority && medium_list[i] == value && remove_num == 0u;
                remove_num += should_remove ? 1u : 0u;
                priority_list[i] = priority_list[i + remove_num];
                medium_list[i] = medium_list[i + remove_num];
            }
            if (remove_num != 0u) {
                size--;
                priority_list[size] = LR_MEDIUM_VACUUM_PRIORITY;
                medium_list[size] = MediumInfo{};
            }
        }
    };
    enum : uint32_t { MEDIUM_ABSORB = 0u, MEDIUM_SCATTER = 1u, MEDIUM_HIT_SURFACE = 3u, MEDIUM_INVALID = ~0u };// medium.h:28-32
    struct MediumSample {// Medium::Sample::zero, medium.h:56-61
        Spectrum3 f{0.f, 0.f, 0.f};
        float pdf{1e16f};
        Ray ray{{0.f, 0.f, 0.f}, 0.f, {0.f, 0.f, 0.f}, 0.f};
        uint32_t event{MEDIUM_INVALID};
    };
    static uint32_t sample_discrete3(float3 w, float u) {// sampling.cpp:182-194 (out-of-range by rounding: last channel)
        auto u_rescaled = u * (w.x + w.y + w.z);
        auto accum = 0.f;
        for (auto i = 0u; i < 3u; i++) {
            accum += w[i];
            if (u_rescaled <= accum) { return i; }
        }
        return 2u;
    }
    static float3 random_channel_pdf(PCG32 &rng) {// homogeneous.cpp:50-54
        float3 pc;
        pc.x = rng.uniform_float(), pc.y = rng.uniform_float(), pc.z = rng.uniform_float();
        return pc / (pc.x + pc.y + pc.z);
    }
    static float hg_p(float g, float3 wo, float3 wi) {// henyey_greenstein.cpp:22-27
        auto denom = 1.f + g * g + 2.f * g * dot(wo, wi);
        return inv_pi * 0.25f * (1.f - g * g) / (denom * std::sqrt(std::max(0.f, denom)));
    }
    // HomogeneousMediumClosure::sample, homogeneous.cpp:48-122
    static MediumSample medium_sample(const lr_medium &m, const Ray &ray, float t_max, PCG32 &rng) {
        MediumSample out;
        float3 sigma_a{m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]}, sigma_s{m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]};
        auto sigma_t = sigma_a + sigma_s;
        auto pdf_channels = random_channel_pdf(rng);
        auto channel = sample_discrete3(pdf_channels, rng.uniform_float());
        auto u = rng.uniform_float();
        auto t = -std::log(std::max(1.f - u, std::numeric_limits<float>::min())) / sigma_t[channel];
        auto transmittance = [&](float d) { return float3{std::exp(-sigma_t.x * d), std::exp(-sigma_t.y * d), std::exp(-sigma_t.z * d)}; };
        if (t > t_max) {// hit surface
            out.event = MEDIUM_HIT_SURFACE;
            auto Tr = transmittance(t_max);
            out.ray = Ray{ray.o + ray.d * t_max, 0.f, ray.d, std::numeric_limits<float>::max()};
            auto pdf = pdf_channels * Tr;
            out.f = Tr, out.pdf = pdf.x + pdf.y + pdf.z;
        } else {
            auto p_absorb = sigma_a[channel] / sigma_t[channel], p_scatter = sigma_s[channel] / sigma_t[channel];
            auto index = rng.uniform_float() * (p_absorb + p_scatter) <= p_absorb ? 0u : 1u;// sample_discrete(float2), sampling.cpp:162-166
            if (index == 0u) {// absorb
                out.event = MEDIUM_ABSORB;
                out.ray = ray;
                out.f = f3(0.f);
                auto pdf = pdf_channels * sigma_t;
                out.pdf = pdf.x + pdf.y + pdf.z;
            } else {// scatter: the sampled direction is built around the WORLD y axis, not around wo (henyey_greenstein.cpp:28-45; kept)
                out.event = MEDIUM_SCATTER;
                auto Tr = transmittance(t);
                float2 up{0.f, 0.f};
                up.x = rng.uniform_float(), up.y = rng.uniform_float();
                auto g = m.g;
                auto cos_theta = std::abs(g) < 1e-3f ? 1.f - 2.f * up.x :
                                                       -1.f / (2.f * g) * (1.f + g * g - sqr((1.f - g * g) / (1.f + g - 2.f * g * up.x)));
                auto sin_theta = std::sqrt(std::max(0.f, 1.f - cos_theta * cos_theta));
                auto phi = 2.f * pi * up.y;
                float3 wi{sin_theta * std::cos(phi), cos_theta, sin_theta * std::sin(phi)};
                out.ray = Ray{ray.o + ray.d * t, 0.f, wi, std::numeric_limits<float>::max()};
                auto pdf = pdf_channels * (sigma_t * Tr);
                out.f = Tr * sigma_s, out.pdf = pdf.x + pdf.y + pdf.z;
            }
        }
        return out;
    }
    struct TrEval {
        Spectrum3 f{1.f, 1.f, 1.f};
        float pdf{0.f};
    };
    // HomogeneousMediumClosure::transmittance, homogeneous.cpp:124-137; Vacuum: Evaluation::zero (f = 0, pdf = 1e16)
    static TrEval medium_transmittance(const lr_medium &m, float t, PCG32 &rng) {
        if (m.kind == LR_MEDIUM_VACUUM) { return TrEval{f3(0.f), 1e16f}; }
        auto pdf_channels = random_channel_pdf(rng);
        float3 Tr{std::exp(-(m.sigma_a[0] + m.sigma_s[0]) * t), std::exp(-(m.sigma_a[1] + m.sigma_s[1]) * t), std::exp(-(m.sigma_a[2] + m.sigma_s[2]) * t)};
        auto pdf = pdf_channels * Tr;
        return TrEval{Tr, pdf.x + pdf.y + pdf.z};
    }
    // MegakernelVolumePathTracingNaiveInstance::_event, mega_vpt_naive.cpp:68-93
    uint32_t surface_event(const Interaction &it, float3 wo, float3 wi) const {
        auto shading = it.has_surface() ? Closure::populate(*scene, it, wo, 1.f).shading : it.shading;
        auto wo_local = shading.world_to_local(wo), wi_local = shading.world_to_local(wi);
        return wo_local.z * wi_local.z > 0.f ? EVENT_REFLECT : (wi_local.z > 0.f ? EVENT_EXIT : EVENT_ENTER);
    }
    // _transmittance, mega_vpt_naive.cpp:95-168: walks the shadow segment through every surface on it
    TrEval transmittance(PCG32 &rng, MediumTracker tracker, Ray origin_ray, PathStats &stats) const {
        auto &s = *scene;
        auto dir = origin_ray.d;
        auto ray = origin_ray;
        auto light_p = origin_ray.o + dir * origin_ray.t_max;
        TrEval tr;
        // The reference loops `while any(f > 0)` with no bound.  A shadow segment that lies IN a surface (a sample on the lamp
        // the segment starts on) re-hits that surface every few ULPs and never gets there; on the GPU that is a hang.  Both
        // the oracle and the device stop after kMaxCrossings surfaces (conscious deviation; real segments cross a handful).
        constexpr auto kMaxCrossings = 64u;
        auto crossings = 0u;
        while (tr.f.x > 0.f || tr.f.y > 0.f || tr.f.z > 0.f) {
            if (crossings++ == kMaxCrossings) { break; }
            stats.closest++;
            auto hit = accel.trace(ray, false, stats.trace);
            if (hit.miss()) { break; }
            stats.hits++;
            auto wo = -dir, wi = dir;
            auto it = make_interaction(hit.inst, hit.prim, f3(1.f - hit.bary.x - hit.bary.y, hit.bary.x, hit.bary.y), true, -ray.d);
            auto t2surface = length(it.pg - ray.o);
            auto has_medium = (it.flags() & LR_SHAPE_HAS_MEDIUM) != 0u;
            auto medium_tag = it.handle.y >> 24u;
            auto event = surface_event(it, wo, wi);
            if (!tracker.vacuum()) {
                auto e = medium_transmittance(s.media[tracker.current().medium_tag], t2surface, rng);
                tr.f = tr.f * e.f, tr.pdf += e.pdf;
            }
            if (has_medium) {
                auto priority = s.media[medium_tag].priority;
                MediumInfo info{priority, medium_tag};
                if (event == EVENT_EXIT) { tracker.exit(priority, info); } else { tracker.enter(priority, info); }
            }
            if (it.has_surface()) {
                auto e = Closure::populate(s, it, wo, 1.f).evaluate(wo, wi);
                tr.f = tr.f * e.f, tr.pdf += e.pdf;
            }
            ray = spawn_ray_to(it, light_p);
        }
        return tr;
    }
    // MegakernelVolumePathTracingNaiveInstance::Li, mega_vpt_naive.cpp:170-483 (VPT_NAIVE_ENABLE_DIRECT_LIGHTING defined, :16)
    float3 Li_vpt(uint32_t px, uint32_t py, uint32_t sample_index, PathStats &stats) const {
        auto &s = *scene;
        Sampler sampler;
        sampler.start(s.sampler, px, py, sample_index, s.camera.width, s.camera.height);
        auto u_filter = sampler.generate_pixel_2d();
        auto u_lens = s.camera.kind == LR_CAMERA_THIN_LENS ? sampler.generate_2d() : float2{.5f, .5f};
        auto cs = generate_camera_ray(s, px, py, u_filter, u_lens);
        Spectrum3 beta = f3(cs.weight);
        Spectrum3 Li = f3(0.f);
        MediumTracker tracker;
        auto u_rng = sampler.generate_2d();// PCG32 rng(U64(as<UInt2>(generate_2d()))): x -> high word, y -> low word (u64.h:48-51)
        uint32_t hi, lo;
        std::memcpy(&hi, &u_rng.x, 4u), std::memcpy(&lo, &u_rng.y, 4u);
        PCG32 rng{(static_cast<uint64_t>(hi) << 32u) | lo};
        if (auto env = s.integrator.environment_medium_tag; env != LR_INVALID_ID) {
            tracker.enter(s.media[env].priority, MediumInfo{s.media[env].priority, env});
        }
        auto ray = cs.ray;
        auto pdf_bsdf = 1e16f;
        auto eta_scale = 1.f;
        auto has_env = s.environment.kind != LR_ENV_NONE;
        auto rr_depth = s.integrator.rr_depth;
        for (auto depth = 0u; depth < s.integrator.max_depth; depth++) {
            auto eta = 1.f;
            auto u_rr = 0.f;
            if (depth + 1u >= rr_depth) { u_rr = sampler.generate_1d(); }
            stats.closest++;
            auto hit = accel.trace(ray, false, stats.trace);
            Interaction it;
            if (!hit.miss()) {
                stats.hits++;
                it = make_interaction(hit.inst, hit.prim, f3(1.f - hit.bary.x - hit.bary.y, hit.bary.x, hit.bary.y), true, -ray.d);
            }
            auto has_medium = it.valid() && (it.flags() & LR_SHAPE_HAS_MEDIUM) != 0u;
            auto t_max = it.valid() ? length(it.pg - ray.o) : std::numeric_limits<float>::max();
            MediumSample ms;
            if (!tracker.vacuum()) {
                // direct lighting of the medium point at the ray origin, :283-299
                auto u_light_selection = sampler.generate_1d();
                auto u_light_surface = sampler.generate_2d();
                Interaction it_medium;// Interaction{pg}: ng = pg (interaction.h:77-78), identity frame, null shape
                it_medium.pg = ray.o, it_medium.ng = ray.o, it_medium.ps = f3(0.f);
                it_medium.shading = Frame{};
                stats.nee++;
                auto light_sample = this->light_sample(it_medium, u_light_selection, u_light_surface);
                auto tr = transmittance(rng, tracker, light_sample.shadow_ray, stats);
                if (tr.pdf > 0.f) {
                    auto w = 1.f / (pdf_bsdf + tr.pdf + light_sample.eval.pdf);
                    Li += w * beta * tr.f * light_sample.eval.L;
                }
                auto &medium = s.media[tracker.current().medium_tag];
                eta = medium.eta;
                if (medium.kind != LR_MEDIUM_VACUUM) {// :305-313
                    ms = medium_sample(medium, ray, t_max, rng);
                    ray = ms.ray;
                    auto w = ms.pdf > 0.f ? 1.f / ms.pdf : 0.f;
                    beta *= ms.f * w;
                    pdf_bsdf = ms.pdf;
                }
            }
            if (ms.event == MEDIUM_INVALID || ms.event == MEDIUM_HIT_SURFACE) {// sample the surface, :318-457
                if (!it.valid()) {
                    if (has_env) {
                        auto eval = env_evaluate(ray.d);
                        Li += beta * eval.L * balance_heuristic(pdf_bsdf, eval.pdf * s.integrator.env_prob);
                    }
                    break;
                }
                if (s.light_count != 0u && it.has_light()) {
                    auto eval = light_evaluate(it, ray.o);
                    eval.pdf *= (1.f - s.integrator.env_prob) / static_cast<float>(s.integrator.light_count);
                    Li += beta * eval.L * balance_heuristic(pdf_bsdf, eval.pdf);
                }
                if (!it.has_surface()) { break; }
                stats.bounces++;
                auto u_light_selection = sampler.generate_1d();
                auto u_light_surface = sampler.generate_2d();
                auto u_lobe = sampler.generate_1d();
                auto u_bsdf = sampler.generate_2d();
                stats.nee++;
                auto light_sample = this->light_sample(it, u_light_selection, u_light_surface);
                auto tr = transmittance(rng, tracker, light_sample.shadow_ray, stats);
                auto medium_tag = it.handle.y >> 24u;
                auto medium_priority = LR_MEDIUM_VACUUM_PRIORITY;
                auto eta_next = 1.f;
                if (has_medium) { medium_priority = s.media[medium_tag].priority, eta_next = s.media[medium_tag].eta; }
                MediumInfo info{medium_priority, medium_tag};
                auto event_skip = surface_event(it, -ray.d, ray.d);
                auto wo = -ray.d;
                auto closure = Closure::populate(s, it, wo, eta);
                uint32_t event;
                if (!tracker.true_hit(info.medium_tag)) {// (the TAG is passed where a priority is expected, :384; kept)
                    event = event_skip;
                    ray = spawn_ray(it, ray.d);
                    pdf_bsdf = 1e16f;
                } else {
                    if (light_sample.eval.pdf > 0.0f) {
                        auto eval = closure.evaluate(wo, light_sample.shadow_ray.d);
                        auto w = 1.f / (light_sample.eval.pdf + eval.pdf + tr.pdf);
                        Li += w * beta * eval.f * light_sample.eval.L * tr.f;
                    }
                    auto ss = closure.sample(wo, u_lobe, u_bsdf);
                    event = ss.event;
                    auto w = ss.eval.pdf > 0.f ? 1.f / ss.eval.pdf : 0.f;
                    pdf_bsdf = ss.eval.pdf;
                    ray = spawn_ray(it, ss.wi);
                    beta *= w * ss.eval.f;
                    if (has_medium) {
                        if (event == EVENT_ENTER) { eta_scale = sqr(eta_next / eta); }
                        else if (event == EVENT_EXIT) { eta_scale = sqr(eta / eta_next); }
                    }
                }
                if (has_medium) {
                    if (event == EVENT_ENTER) { tracker.enter(medium_priority, info); }
                    else if (event == EVENT_EXIT) { tracker.exit(medium_priority, info); }
                }
            }
            if (any_nan(beta)) { beta = f3(0.f); }
            if (beta.x <= 0.f && beta.y <= 0.f && beta.z <= 0.f) { break; }
            auto q = std::max(max_component(beta) * eta_scale, .05f);
            if (depth + 1u >= rr_depth) {
                if (q < s.integrator.rr_threshold && u_rr >= q) { break; }
                beta *= q < s.integrator.rr_threshold ? 1.0f / q : 1.f;
            }
        }
        return Li;
    }

    // MegakernelPathTracingInstance::Li, mega_path.cpp:49-156
    float3 Li(uint32_t px, uint32_t py, uint32_t sample_index, PathStats &stats) const {
        auto &s = *scene;
        if (s.integrator.kind == LR_INTEGRATOR_NORMAL) { return Li_normal(px, py, sample_index, stats); }
        if (s.integrator.kind == LR_INTEGRATOR_DIRECT) { return Li_direct(px, py, sample_index, stats); }
        if (s.integrator.kind == LR_INTEGRATOR_VPT_NAIVE) { return Li_vpt(px, py, sample_index, stats); }
        Sampler sampler;
        sampler.start(s.sampler, px, py, sample_index, s.camera.width, s.camera.height);
        auto u_filter = sampler.generate_pixel_2d();
        auto u_lens = s.camera.kind == LR_CAMERA_THIN_LENS ? sampler.generate_2d() : float2{.5f, .5f};
        auto cs = generate_camera_ray(s, px, py, u_filter, u_lens);
        Spectrum3 beta = f3(cs.weight);
        Spectrum3 Li = f3(0.f);
        auto ray = cs.ray;
        auto pdf_bsdf = 1e16f;
        auto has_env = s.environment.kind != LR_ENV_NONE;
        for (auto depth = 0u; depth < s.integrator.max_depth; depth++) {
            auto wo = -ray.d;
            stats.closest++;
            auto hit = accel.trace(ray, false, stats.trace);
            if (hit.miss()) {
                if (has_env) {// evaluate_miss, uniform.cpp:67-76
                    auto eval = env_evaluate(ray.d);
                    Li += beta * eval.L * balance_heuristic(pdf_bsdf, eval.pdf * s.integrator.env_prob);
                }
                break;
            }
            stats.hits++;
            auto it = make_interaction(hit.inst, hit.prim, f3(1.f - hit.bary.x - hit.bary.y, hit.bary.x, hit.bary.y), true, wo);
            if (s.light_count != 0u && it.has_light()) {// evaluate_hit, uniform.cpp:50-65
                auto eval = light_evaluate(it, ray.o);
                eval.pdf *= (1.f - s.integrator.env_prob) / static_cast<float>(s.integrator.light_count);
                Li += beta * eval.L * balance_heuristic(pdf_bsdf, eval.pdf);
            }
            if (!it.has_surface()) { break; }
            stats.bounces++;
            auto u_light_selection = sampler.generate_1d();
            auto u_light_surface = sampler.generate_2d();
            auto u_lobe = sampler.generate_1d();
            auto u_bsdf = sampler.generate_2d();
            auto u_rr = 0.f;
            auto rr_depth = s.integrator.rr_depth;
            if (depth + 1u >= rr_depth) { u_rr = sampler.generate_1d(); }
            stats.nee++;
            auto light_sample = this->light_sample(it, u_light_selection, u_light_surface);
            stats.shadow++;
            auto occluded = !accel.trace(light_sample.shadow_ray, true, stats.trace).miss();
            auto eta_scale = 1.f;
            auto closure = Closure::populate(s, it, wo, 1.f);
            if (light_sample.eval.pdf > 0.0f && !occluded) {
                auto wi = light_sample.shadow_ray.d;
                auto eval = closure.evaluate(wo, wi);
                auto w = balance_heuristic(light_sample.eval.pdf, eval.pdf) / light_sample.eval.pdf;
                Li += w * beta * eval.f * light_sample.eval.L;
            }
            auto surface_sample = closure.sample(wo, u_lobe, u_bsdf);
            ray = spawn_ray(it, surface_sample.wi);
            pdf_bsdf = surface_sample.eval.pdf;
            auto w = surface_sample.eval.pdf > 0.f ? 1.f / surface_sample.eval.pdf : 0.f;
            beta *= w * surface_sample.eval.f;
            auto eta = closure.has_eta ? closure.eta_value : 1.f;
            if (surface_sample.event == EVENT_ENTER) { eta_scale = sqr(eta); }
            else if (surface_sample.event == EVENT_EXIT) { eta_scale = sqr(1.f / eta); }
            if (any_nan(beta)) { beta = f3(0.f); }// zero_if_any_nan, spec.cpp:404-407
            if (beta.x <= 0.f && beta.y <= 0.f && beta.z <= 0.f) { break; }
            auto rr_threshold = s.integrator.rr_threshold;
            auto q = std::max(max_component(beta) * eta_scale, .05f);
            if (depth + 1u >= rr_depth) {
                if (q < rr_threshold && u_rr >= q) { break; }
                beta *= q < rr_threshold ? 1.0f / q : 1.f;
            }
        }
        return Li;
    }
};

// ColorFilmInstance::_accumulate (effective_spp = 1), color.cpp:107-130
static inline void film_accumulate(const lr_scene &scene, float *film, uint32_t px, uint32_t py, float3 rgb) {
    auto p = film + (static_cast<size_t>(py) * scene.camera.width + px) * 4u;
    if (!(any_nan(rgb) || any_inf(rgb))) {
        auto threshold = scene.film.clamp * std::max(1.f, 1.f);
        auto a = abs3(rgb);
        auto strength = std::max(std::max(std::max(a.x, a.y), a.z), 0.f);
        auto c = rgb * (threshold / std::max(strength, threshold));
        if (c.x != 0.f || c.y != 0.f || c.z != 0.f) { p[0] += c.x, p[1] += c.y, p[2] += c.z; }
        p[3] += 1.f;
    }
}

extern "C" {

oracle_ctx *oracle_create(const lr_scene *scene) { return new oracle_ctx{scene}; }
void oracle_destroy(oracle_ctx *ctx) { delete ctx; }
void oracle_set_shutter_weight(oracle_ctx *ctx, float weight) { ctx->shutter_weight = weight; }
// baked-geometry mode (oracle_bvh.h: Accel::set_bake): 0 = ok, -1 = the scene holds no baked triangles (lrhost_scene_build_accel not run)
int oracle_set_bake(oracle_ctx *ctx, int on) { return ctx->accel.set_bake(on != 0) ? 0 : -1; }

int oracle_render(oracle_ctx *ctx, uint32_t spp_begin, uint32_t spp_end, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                  int threads, float *film, oracle_counters *counters) {
    auto &scene = *ctx->scene;
    if (x1 > scene.camera.width || y1 > scene.camera.height || x0 > x1 || y0 > y1) { return -1; }
    // mega_path.cpp:40-46, direct.cpp:57-63: no lights -> black (the normal visualiser needs none)
    if (scene.light_count == 0u && scene.environment.kind == LR_ENV_NONE && scene.integrator.kind != LR_INTEGRATOR_NORMAL) { return 0; }
    threads = std::max(threads, 1);
    std::atomic<uint32_t> next_row{y0};
    std::vector<oracle_ctx::PathStats> stats(static_cast<size_t>(threads));
    auto work = [&](int tid) {
        auto &st = stats[static_cast<size_t>(tid)];
        for (;;) {
            auto y = next_row.fetch_add(1u);
            if (y >= y1) { break; }
            for (auto x = x0; x < x1; x++) {
                for (auto sidx = spp_begin; sidx < spp_end; sidx++) {
                    auto L = ctx->Li(x, y, sidx, st);
                    film_accumulate(scene, film, x, y, ctx->shutter_weight * L);// integrator.cpp:74
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (auto t = 1; t < threads; t++) { pool.emplace_back(work, t); }
    work(0);
    for (auto &t : pool) { t.join(); }
    if (counters != nullptr) {
        counters->paths += static_cast<uint64_t>(x1 - x0) * (y1 - y0) * (spp_end - spp_begin);
        for (auto &st : stats) {
            counters->closest_rays += st.closest, counters->shadow_rays += st.shadow;
            counters->nodes_visited += st.trace.nodes, counters->tris_tested += st.trace.tris;
            counters->surface_hits += st.hits, counters->nee_samples += st.nee;
            counters->path_length_sum += st.bounces;
        }
    }
    return 0;
}

void oracle_film_convert(const lr_scene *scene, const float *film, float *out) {// color.cpp:87-93
    auto n = static_cast<size_t>(scene->camera.width) * scene->camera.height;
    for (size_t i = 0; i < n; i++) {
        auto c = film + i * 4u;
        auto cnt = std::max(c[3], 1.f);
        for (auto k = 0; k < 3; k++) { out[i * 4u + static_cast<size_t>(k)] = ((1.f / cnt) * scene->film.scale[k]) * c[k]; }
        out[i * 4u + 3u] = 1.f;
    }
}

void oracle_li(oracle_ctx *ctx, uint32_t px, uint32_t py, uint32_t sample_index, float rgb_out[3]) {
    oracle_ctx::PathStats st;
    auto L = ctx->Li(px, py, sample_index, st);
    rgb_out[0] = L.x, rgb_out[1] = L.y, rgb_out[2] = L.z;
}

void oracle_trace_closest(oracle_ctx *ctx, const float o[3], const float d[3], float t_min, float t_max,
                          uint32_t out_ids[2], float out_bary_t[3]) {
    TraceCounters tc;
    auto hit = ctx->accel.trace({f3(o[0], o[1], o[2]), t_min, f3(d[0], d[1], d[2]), t_max}, false, tc);
    out_ids[0] = hit.inst, out_ids[1] = hit.prim;
    out_bary_t[0] = hit.bary.x, out_bary_t[1] = hit.bary.y, out_bary_t[2] = hit.t;
}

void oracle_camera_ray(oracle_ctx *ctx, uint32_t px, uint32_t py, uint32_t sample_index, float out[7]) {
    auto &s = *ctx->scene;
    Sampler sampler;
    sampler.start(s.sampler, px, py, sample_index, s.camera.width, s.camera.height);
    auto u_filter = sampler.generate_pixel_2d();
    auto u_lens = s.camera.kind == LR_CAMERA_THIN_LENS ? sampler.generate_2d() : float2{.5f, .5f};
    auto cs = generate_camera_ray(s, px, py, u_filter, u_lens);
    out[0] = cs.ray.o.x, out[1] = cs.ray.o.y, out[2] = cs.ray.o.z;
    out[3] = cs.ray.d.x, out[4] = cs.ray.d.y, out[5] = cs.ray.d.z;
    out[6] = cs.weight;
}

void oracle_sampler_stream(const lr_scene *scene, uint32_t px, uint32_t py, uint32_t sample_index, uint32_t n, float *out) {
    Sampler sampler;
    sampler.start(scene->sampler, px, py, sample_index, scene->camera.width, scene->camera.height);
    auto p = sampler.generate_pixel_2d();
    out[0] = p.x, out[1] = p.y;
    for (uint32_t i = 0; i < n; i++) { out[2u + i] = sampler.generate_1d(); }
}

uint32_t oracle_xxhash32_1(uint32_t x) { return xxhash32(x); }
uint32_t oracle_xxhash32_2(uint32_t x, uint32_t y) { return xxhash32(x, y); }
uint32_t oracle_xxhash32_3(uint32_t x, uint32_t y, uint32_t z) { return xxhash32(x, y, z); }
uint32_t oracle_xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return xxhash32(x, y, z, w); }
float oracle_lcg(uint32_t *state) { return lcg(*state); }
uint32_t oracle_pcg32_next(uint64_t *state, uint64_t *inc) {
    PCG32 g;
    g.state = *state, g.inc = *inc;
    auto v = g.uniform_uint();
    *state = g.state;
    return v;
}
void oracle_pcg32_seed(uint64_t seq_index, uint64_t *state, uint64_t *inc) {
    PCG32 g{seq_index};
    *state = g.state, *inc = g.inc;
}

void oracle_create_alias_table(const float *values, uint32_t n, lr_alias_entry *table, float *pdf) {// sampling.cpp:38-87
    auto sum = 0.0;
    for (uint32_t i = 0; i < n; i++) { sum += std::abs(values[i]); }
    if (sum == 0.) {
        for (uint32_t i = 0; i < n; i++) { pdf[i] = static_cast<float>(1.0 / static_cast<double>(n)); }
    } else {
        auto inv_sum = 1.0 / sum;
        for (uint32_t i = 0; i < n; i++) { pdf[i] = static_cast<float>(std::abs(values[i]) * inv_sum); }
    }
    auto ratio = static_cast<double>(n) / sum;
    std::vector<uint32_t> over, under;
    for (uint32_t i = 0; i < n; i++) {
        auto p = static_cast<float>(values[i] * ratio);
        table[i] = {p, i};
        (p > 1.0f ? over : under).push_back(i);
    }
    while (!over.empty() && !under.empty()) {
        auto o = over.back(), u = under.back();
        over.pop_back(), under.pop_back();
        table[o].prob -= 1.0f - table[u].prob;
        table[u].alias = o;
        if (table[o].prob > 1.0f) { over.push_back(o); }
        else if (table[o].prob < 1.0f) { under.push_back(o); }
    }
    for (auto i : over) { table[i] = {1.0f, i}; }
    for (auto i : under) { table[i] = {1.0f, i}; }
}

void oracle_sample_alias_table(const lr_alias_entry *table, uint32_t n, float u, uint32_t *index, float *u_remapped) {
    auto s = sample_alias_table([&](uint32_t i) { return table[i].prob; }, [&](uint32_t i) { return table[i].alias; }, n, u);
    *index = s.index, *u_remapped = s.u;
}

void oracle_filter_sample(const lr_filter *filter, float ux, float uy, float out[3]) {
    auto s = filter_sample(*filter, {ux, uy});
    out[0] = s.offset.x, out[1] = s.offset.y, out[2] = s.weight;
}

void oracle_encode_handle(uint32_t buffer_base, uint32_t flags, uint32_t surface_tag, uint32_t light_tag, uint32_t medium_tag,
                          uint32_t tri_count, float shadow_term, float isect_offset, uint32_t out[4]) {// shape.cpp:46-70
    auto fixed = [](float x) {
        x = clampf(x, 0.f, 1.f);
        return static_cast<uint32_t>(clampf(std::round(x / (1.f / 65536.f)), 0.f, 65535.f));
    };
    out[0] = (buffer_base << 10u) | flags;
    out[1] = (surface_tag << 12u) | (light_tag << 0u) | (medium_tag << 24u);
    out[2] = tri_count;
    out[3] = (fixed(shadow_term) << 16u) | fixed(isect_offset);
}

void oracle_offset_ray_origin(const float p[3], const float n[3], float out[3]) {
    auto r = offset_ray_origin(f3(p[0], p[1], p[2]), f3(n[0], n[1], n[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}

static Interaction flat_patch(uint32_t surface_tag, const float ns_in[3]) {
    Interaction it;
    it.handle.x = LR_SHAPE_HAS_SURFACE;
    it.handle.y = surface_tag << 12u;
    it.inst = 0u, it.prim = 0u;
    it.ng = f3(0.f, 0.f, 1.f);
    auto ns = normalize(f3(ns_in[0], ns_in[1], ns_in[2]));
    it.shading = Frame::make(face_forward(ns, it.ng), Frame::make(it.ng).s);
    it.uv = {0.25f, 0.75f};
    return it;
}

void oracle_surface_evaluate(const lr_scene *scene, uint32_t surface_tag, const float ns[3], const float wo_in[3],
                             const float wi_in[3], float out[4]) {
    auto it = flat_patch(surface_tag, ns);
    auto wo = f3(wo_in[0], wo_in[1], wo_in[2]), wi = f3(wi_in[0], wi_in[1], wi_in[2]);
    auto e = Closure::populate(*scene, it, wo, 1.f).evaluate(wo, wi);
    out[0] = e.f.x, out[1] = e.f.y, out[2] = e.f.z, out[3] = e.pdf;
}

void oracle_surface_sample(const lr_scene *scene, uint32_t surface_tag, const float ns[3], const float wo_in[3],
                           float u_lobe, float ux, float uy, float out[8]) {
    auto it = flat_patch(surface_tag, ns);
    auto wo = f3(wo_in[0], wo_in[1], wo_in[2]);
    auto s = Closure::populate(*scene, it, wo, 1.f).sample(wo, u_lobe, {ux, uy});
    out[0] = s.eval.f.x, out[1] = s.eval.f.y, out[2] = s.eval.f.z, out[3] = s.eval.pdf;
    out[4] = s.wi.x, out[5] = s.wi.y, out[6] = s.wi.z, out[7] = static_cast<float>(s.event);
}

}// extern "C"
