// dev_shade.h — per-hit work of the megakernel: samplers, camera rays, hit reconstruction,
// light sampling/evaluation, texture lookups and closure resolution.
//
// Reference map (device halves of the L1 plugins inlined into the reference's render kernel):
//   xxhash32 / lcg / PCG32            src/util/rng.cpp:53-69,128-176
//   IndependentSamplerInstance        src/samplers/independent.cpp:57-83
//   Filter::Instance::sample          src/base/filter.cpp:49-64
//   Camera::Instance::generate_ray    src/base/camera.cpp:212-224, src/cameras/{pinhole,thin_lens,ortho}.cpp
//   Geometry::shading_point           src/base/geometry.cpp:345-389
//   Interaction::spawn_ray[_to]       src/base/interaction.cpp:13-30
//   UniformLightSamplerInstance       src/lightsamplers/uniform.cpp:50-137
//   DiffuseLightClosure::_evaluate    src/lights/diffuse.cpp:67-88
//   Texture evaluate_*_spectrum       src/base/texture.cpp:21-79, src/textures/{constant,image}.cpp
#pragma once
#include "dev_bsdf.h"
#include "dev_trace.h"

namespace lrd {

// ---------------------------------------------------------------- RNG / sampler

LR_HD uint32_t xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    constexpr uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    auto rot = [](uint32_t h) { return (h << 17u) | (h >> 15u); };
    auto h = w + P5 + x * P3;
    h = P4 * rot(h);
    h += y * P3;
    h = P4 * rot(h);
    h += z * P3;
    h = P4 * rot(h);
    h = P2 * (h ^ (h >> 15u));
    h = P3 * (h ^ (h >> 13u));
    return h ^ (h >> 16u);
}

LR_HD uint32_t xxhash32_1(uint32_t p) {// rng.cpp:12-23
    constexpr uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    auto h = p + P5;
    h = P4 * ((h << 17u) | (h >> 15u));
    h = P2 * (h ^ (h >> 15u));
    h = P3 * (h ^ (h >> 13u));
    return h ^ (h >> 16u);
}

LR_HD float uint_to_unit_float(uint32_t u) { return fminf(kOneMinusEpsilon, static_cast<float>(u) * 0x1p-32f); }

// TileSharedSamplerInstance::start (src/samplers/tile_shared.cpp:51-62): a wrapped sampler is started with the TILE of the pixel
// (after an optional per-sample jitter of the pixel), so that all pixels of a tile share one sequence
LR_D void tile_shared_pixel(const DScene &scene, uint32_t &x, uint32_t &y, uint32_t index) {
    if (scene.sampler_tile != 0u) {
        if (scene.sampler_tile_jitter != 0u) {
            const auto offset = xxhash32_1(index);
            const auto ox = static_cast<float>(offset >> 16u) * 0x1p-16f, oy = static_cast<float>(offset & 0xffffu) * 0x1p-16f;
            x += static_cast<uint32_t>(ox * static_cast<float>(scene.camera.width)) % scene.camera.width;
            y += static_cast<uint32_t>(oy * static_cast<float>(scene.camera.height)) % scene.camera.height;
        }
        x /= scene.sampler_tile & 0xffffu, y /= scene.sampler_tile >> 16u;
    }
}

// One sampler object per path.  GENERIC = false reproduces the reference's default stream bit for bit
// (IndependentSampler: xxhash32 seed + LCG); GENERIC = true: PCG32 / Sobol / PaddedSobol chosen at run time.
template<bool GENERIC>
struct PathSampler;

template<>
struct PathSampler<false> {
    uint32_t state;
    LR_D void start(const DScene &scene, uint32_t px, uint32_t py, uint32_t index) {
        tile_shared_pixel(scene, px, py, index);
        state = xxhash32_4(px, py, scene.seed, index);
    }
    LR_D float next_1d() {
        state = 1664525u * state + 1013904223u;// lcg, rng.cpp:132-140
        return uint_to_unit_float(state);
    }
    LR_D f2 next_2d() {
        f2 u;
        u.x = next_1d();
        u.y = next_1d();
        return u;
    }
    LR_D f2 next_pixel_2d() { return next_2d(); }// Sampler::Instance::generate_pixel_2d default, sampler.h:48
    // the stream position of a path that leaves its lane (megapath_kernel.h: deferred heavy hits)
    static constexpr uint32_t kSavedWords = 1u;
    LR_D uint32_t save(uint32_t *w) const { w[0] = state; return kSavedWords; }
    LR_D void restore(const DScene &, const uint32_t *w) { state = w[0]; }
};

LR_HD uint32_t xxhash32_2(uint32_t x, uint32_t y) {// rng.cpp:25-36
    constexpr uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    auto h = y + P5 + x * P3;
    h = P4 * ((h << 17u) | (h >> 15u));
    h = P2 * (h ^ (h >> 15u));
    h = P3 * (h ^ (h >> 13u));
    return h ^ (h >> 16u);
}

// Generic sampler (runtime kind): PCG32 streams, the global Owen-scrambled Sobol sampler
// (src/samplers/sobol.cpp:40-169) and PaddedSobol (src/samplers/padded_sobol.cpp:23-150).  Lives in its own
// kernel instantiation so the default Independent path keeps a 1-register sampler.
// -DLR_ONLY_SAMPLER=<kind> (experiment): the generic sampler's run-time kind as a compile-time constant -- what a kernel variant per sampler would be
#ifdef LR_ONLY_SAMPLER
#define LR_SAMPLER_KIND_OF(s) static_cast<uint32_t>(LR_ONLY_SAMPLER)
#else
#define LR_SAMPLER_KIND_OF(s) ((s).sampler_kind)
#endif
template<>
struct PathSampler<true> {
    // FOUR words of per-path state (eight until round 3: two 64-bit words and four more, and the kernel spilled 25 VGPRs around them):
    //   PCG32        lo | hi = state,              w2 | w3 = increment
    //   Sobol        lo | hi = sequence index,     w2 = dimension,  w3 = pixel (x | y << 16: generate_pixel_2d)
    //   PaddedSobol  lo = sample index,            w2 = dimension,  w3 = pixel
    uint32_t lo, hi, w2, w3;
    const DScene *scene;

    LR_D uint64_t wide() const { return lo | (static_cast<uint64_t>(hi) << 32u); }
    LR_D void set_wide(uint64_t v) { lo = static_cast<uint32_t>(v), hi = static_cast<uint32_t>(v >> 32u); }
    LR_D uint32_t pcg_next() {// rng.cpp:142-148
        auto old = wide();
        set_wide(old * 0x5851f42d4c957f2dull + (w2 | (static_cast<uint64_t>(w3) << 32u)));
        auto xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
        auto rot = static_cast<uint32_t>(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    static LR_D uint32_t owen(uint32_t seed, uint32_t v) {// _fast_owen_scramble, sobol.cpp:40-48
        v = __brev(v);
        v ^= v * 0x3d20adeau;
        v += seed;
        v *= (seed >> 16u) | 1u;
        v ^= v * 0x05526c56u;
        v ^= v * 0x53a22864u;
        return __brev(v);
    }
    // sobol.cpp:52-60: the XOR of the matrix columns the index has bits in -- taken a BYTE of the index at a time from the table
    // lrhip_upload_scene builds out of the same matrices (the product is linear over GF(2)): 4-5 independent loads per draw instead
    // of a 30-40 trip loop with a dependent load per set bit (C2 stand-in, Sobol sampler: 425 -> 703 Msamples/s, profiles/archive/r03ac_*)
    LR_D uint32_t sobol_bits(uint64_t idx, uint32_t dim) const {
        constexpr uint32_t kBytes = (static_cast<uint32_t>(LR_SOBOL_MATRIX_SIZE) + 7u) / 8u;
        auto t = scene->sobol_bytes + dim * (kBytes * 256u);
        auto lo = static_cast<uint32_t>(idx), hi = static_cast<uint32_t>(idx >> 32u);
        auto v = t[lo & 255u] ^ t[256u + ((lo >> 8u) & 255u)] ^ t[512u + ((lo >> 16u) & 255u)] ^ t[768u + (lo >> 24u)];
        for (t += 1024u; hi != 0u; hi >>= 8u, t += 256u) { v ^= t[hi & 255u]; }// (bits 32+: frames of 4k and more at high spp)
        return v;
    }
    // Dimensions 0 and 1 of the Sobol sequence in closed form, for 32-bit indices (all PaddedSobol ever asks for,
    // padded_sobol.cpp:127-149): the generator matrix of dimension 0 is the bit reversal and that of dimension 1 is Pascal's triangle
    // mod 2 (each column = the previous one ^ itself >> 1), whose product with a vector is a five-step butterfly.  The table walk
    // of sobol_bits -- one dependent, wave-uniform global load per set bit of the index -- cost the generic-sampler kernel 31 % on
    // the C2 stand-in (round 3: 580 vs 835 Msamples/s).  tests/test_sobol.py holds both forms to the table word for word.
    static LR_D uint32_t sobol_bits_dim0(uint32_t idx) { return __brev(idx); }
    static LR_D uint32_t sobol_bits_dim1(uint32_t idx) {
        idx ^= (idx >> 1u) & 0x55555555u;
        idx ^= (idx >> 2u) & 0x33333333u;
        idx ^= (idx >> 4u) & 0x0f0f0f0fu;
        idx ^= (idx >> 8u) & 0x00ff00ffu;
        idx ^= (idx >> 16u) & 0x0000ffffu;
        return __brev(idx);
    }
    static LR_D uint32_t permutation_element(uint32_t i, uint32_t l, uint32_t p) {// padded_sobol.cpp:59-91
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
        // (l a power of two -- every spp anyone renders at: w == l - 1, the loop ran once, and the modulo is a mask instead of the ~35
        // instructions of a 32-bit division, once per draw; wave-uniform)
        return (l & w) == 0u ? (i + p) & w : (i + p) % l;
    }
    static constexpr uint32_t kSavedWords = 4u;// (the stream position of a path that leaves its lane: wavefront mode)
    LR_D uint32_t save(uint32_t *w) const {
        w[0] = lo, w[1] = hi, w[2] = w2, w[3] = w3;
        return kSavedWords;
    }
    LR_D void restore(const DScene &s, const uint32_t *w) {
        scene = &s;
        lo = w[0], hi = w[1], w2 = w[2], w3 = w[3];
    }
    LR_D void start(const DScene &s, uint32_t x, uint32_t y, uint32_t index) {
        scene = &s;
        tile_shared_pixel(s, x, y, index);
        w3 = x | (y << 16u);// (frames are at most 65535 pixels wide / high: lr_camera, lrhip_upload_scene)
        if (LR_SAMPLER_KIND_OF(s) == LR_SAMPLER_SOBOL) {// sobol.cpp:131-136 + _sobol_interval_to_index :67-96
            w2 = 2u;
            auto m = 31u - static_cast<uint32_t>(__clz(static_cast<int>(s.sobol_scale)));
            if (m == 0u) {
                lo = index, hi = 0u;
            } else {
                // (both products a byte at a time from the tables lrhip_upload_scene builds out of vdc_sobol / vdc_sobol_inv, like sobol_bits)
                auto frame = index;
                auto idx = static_cast<uint64_t>(frame) << (m << 1u);
                uint64_t delta = 0u;
                for (auto t = s.vdc_bytes; frame != 0u; frame >>= 8u, t += 256u) { delta ^= t[frame & 255u]; }
                auto bb = delta ^ ((static_cast<uint64_t>(x) << m) | y);
                for (auto t = s.vdc_inv_bytes; bb != 0u; bb >>= 8u, t += 256u) { idx ^= t[bb & 255u]; }
                set_wide(idx);
            }
        } else if (LR_SAMPLER_KIND_OF(s) == LR_SAMPLER_PADDED_SOBOL) {
            lo = index, hi = 0u, w2 = 0u;
        } else {// PCG32::set_sequence(xxhash32 seed), rng.cpp:150-156
            lo = 0u, hi = 0u;
            const auto inc = (static_cast<uint64_t>(xxhash32_4(x, y, s.seed, index)) << 1u) | 1u;
            w2 = static_cast<uint32_t>(inc), w3 = static_cast<uint32_t>(inc >> 32u);
            (void)pcg_next();
            set_wide(wide() + 0x853c49e6748fea9bull);
            (void)pcg_next();
        }
    }
    LR_D float next_1d() {
        auto kind = LR_SAMPLER_KIND_OF(*scene);
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:147-153
            w2 = w2 >= static_cast<uint32_t>(LR_SOBOL_DIMENSIONS) ? 2u : w2;
            auto u = static_cast<float>(owen(xxhash32_2(w2, scene->seed), sobol_bits(wide(), w2))) * 0x1p-32f;
            w2 += 1u;
            return clampf(u, 0.f, kOneMinusEpsilon);
        }
        if (kind == LR_SAMPLER_PADDED_SOBOL) {// padded_sobol.cpp:127-136
            auto hash = xxhash32_4(w3 & 0xffffu, w3 >> 16u, lo ^ scene->seed, w2);
            auto index = permutation_element(lo, scene->sampler_spp, hash);
            w2 += 1u;
            return fminf(static_cast<float>(owen(hash, sobol_bits_dim0(index))) * 0x1p-32f, kOneMinusEpsilon);
        }
        return uint_to_unit_float(pcg_next());
    }
    LR_D f2 next_2d() {
        auto kind = LR_SAMPLER_KIND_OF(*scene);
        f2 u;
        if (kind == LR_SAMPLER_SOBOL) {// sobol.cpp:154-162
            w2 = w2 + 1u >= static_cast<uint32_t>(LR_SOBOL_DIMENSIONS) ? 2u : w2;
            u.x = clampf(static_cast<float>(owen(xxhash32_2(w2, scene->seed), sobol_bits(wide(), w2))) * 0x1p-32f, 0.f, kOneMinusEpsilon);
            u.y = clampf(static_cast<float>(owen(xxhash32_2(w2 + 1u, scene->seed), sobol_bits(wide(), w2 + 1u))) * 0x1p-32f, 0.f, kOneMinusEpsilon);
            w2 += 2u;
            return u;
        }
        if (kind == LR_SAMPLER_PADDED_SOBOL) {// padded_sobol.cpp:137-149
            const auto x = w3 & 0xffffu, y = w3 >> 16u;
            auto hx = xxhash32_4(x, y, lo ^ scene->seed, w2);
            auto hy = xxhash32_4(x, y, lo ^ scene->seed, w2 + 1u);
            auto index = permutation_element(lo, scene->sampler_spp, hx);
            u.x = fminf(static_cast<float>(owen(hx, sobol_bits_dim0(index))) * 0x1p-32f, kOneMinusEpsilon);
            u.y = fminf(static_cast<float>(owen(hy, sobol_bits_dim1(index))) * 0x1p-32f, kOneMinusEpsilon);
            w2 += 2u;
            return u;
        }
        u.x = next_1d();
        u.y = next_1d();
        return u;
    }
    LR_D f2 next_pixel_2d() {// generate_pixel_2d: sobol.cpp:163-169, default sampler.h:48
        if (LR_SAMPLER_KIND_OF(*scene) == LR_SAMPLER_SOBOL) {
            auto s = static_cast<float>(scene->sobol_scale);
            return {clampf(static_cast<float>(sobol_bits(wide(), 0u)) * 0x1p-32f * s - static_cast<float>(w3 & 0xffffu), 0.f, kOneMinusEpsilon),
                    clampf(static_cast<float>(sobol_bits(wide(), 1u)) * 0x1p-32f * s - static_cast<float>(w3 >> 16u), 0.f, kOneMinusEpsilon)};
        }
        return next_2d();
    }
};

struct AliasPick {
    uint32_t index;
    float u;
};
// sample_alias_table, src/util/sampling.h:38-66
LR_HD AliasPick alias_pick(float prob_i, uint32_t alias_i, uint32_t i, float u_remapped) {
    AliasPick p;
    p.index = u_remapped < prob_i ? i : alias_i;
    p.u = u_remapped < prob_i ? u_remapped / prob_i : (u_remapped - prob_i) / (1.0f - prob_i);
    return p;
}
LR_HD uint32_t alias_slot(float u, uint32_t n, float &u_remapped) {
    auto x = u * static_cast<float>(n);
    u_remapped = fract(x);
    return min(static_cast<uint32_t>(fmaxf(x, 0.f)), n - 1u);
}

// ---------------------------------------------------------------- camera

struct FilterTables {// LDS-resident copy of lr_filter
    const lr_filter *f;
};

LR_D void filter_sample(const lr_filter *f, f2 u, f2 &offset, float &weight) {// filter.cpp:49-64
    constexpr uint32_t n = LR_FILTER_LUT_SIZE - 1u;
    float ry, rx;
    auto sy = alias_slot(u.x, n, ry);// x/y swap as in the reference
    auto sx = alias_slot(u.y, n, rx);
    auto py = alias_pick(f->alias_prob[sy], f->alias_index[sy], sy, ry);
    auto px = alias_pick(f->alias_prob[sx], f->alias_index[sx], sx, rx);
    auto pdf = f->pdf[py.index] * f->pdf[px.index];
    auto fv = lerp(f->lut[px.index], f->lut[px.index + 1u], px.u) * lerp(f->lut[py.index], f->lut[py.index + 1u], py.u);
    f2 p{static_cast<float>(px.index) + px.u, static_cast<float>(py.index) + py.u};
    constexpr auto inv_size = 1.0f / static_cast<float>(LR_FILTER_LUT_SIZE);
    offset = {(p.x * inv_size * 2.0f - 1.0f) * f->radius + f->shift[0], (p.y * inv_size * 2.0f - 1.0f) * f->radius + f->shift[1]};
    weight = fv / pdf;
}

LR_D void camera_ray(const DScene &scene, const lr_filter *filter, uint32_t px, uint32_t py, f2 u_filter, f2 u_lens,
                     Ray &ray, float &weight) {
    auto &cam = scene.camera;
    f2 off;
    float fw;
    filter_sample(filter, u_filter, off, fw);
    f2 pixel{static_cast<float>(px) + .5f + off.x, static_cast<float>(py) + .5f + off.y};
    f2 res{static_cast<float>(cam.width), static_cast<float>(cam.height)};
    f3 o = mk3(0.f), d;
    if (cam.kind == LR_CAMERA_PINHOLE) {// pinhole.cpp:60-67
        auto k = cam.tan_half_fov / res.y;
        d = normalize(mk3((pixel.x * 2.0f - res.x) * k, -((pixel.y * 2.0f - res.y) * k), -1.f));
    } else if (cam.kind == LR_CAMERA_THIN_LENS) {// thin_lens.cpp:91-101
        f2 cf{(pixel.x - .5f * res.x) * cam.projected_pixel_size, (pixel.y - .5f * res.y) * cam.projected_pixel_size};
        auto p_focal = mk3(cf.x, -cf.y, -cam.focus_distance);
        auto disk = sample_disk_concentric(u_lens);
        o = mk3(disk.x * cam.lens_radius, disk.y * cam.lens_radius, 0.f);
        d = normalize(p_focal - o);
    } else {// ortho.cpp:52-58
        o = mk3((pixel.x * 2.0f - res.x) / res.y * cam.ortho_scale, -((pixel.y * 2.0f - res.y) / res.y * cam.ortho_scale), 0.f);
        d = mk3(0.f, 0.f, -1.f);
    }
    auto cos_axis = dot(d, mk3(0.f, 0.f, -1.f));// clip planes, camera.h:147-157
    ray.t_min = cam.clip_near / cos_axis;
    ray.t_max = cam.clip_far / cos_axis;
    auto m = cam.c2w;
    ray.o = mk3(m[0] * o.x + m[4] * o.y + m[8] * o.z + m[12], m[1] * o.x + m[5] * o.y + m[9] * o.z + m[13],
                m[2] * o.x + m[6] * o.y + m[10] * o.z + m[14]);
    ray.d = normalize(mk3(m[0], m[1], m[2]) * d.x + mk3(m[4], m[5], m[6]) * d.y + mk3(m[8], m[9], m[10]) * d.z);
    weight = 1.f * fw;
}

// ---------------------------------------------------------------- textures

// Texel coordinate under the texture's address mode.  The floor-modulo of REPEAT / MIRROR goes through the float reciprocal of the
// period and one corrective step either way instead of two integer divisions (`((v % n) + n) % n`: ~70 VALU instructions on gfx950,
// sixteen of them per bilinear lookup before round 3).  One step is exact while |v / period| < 2^20; beyond that a first step brings
// the quotient below 2^10 (its own error: |v / period| * 2^-22 periods) and the second one is exact -- no integer division anywhere.
LR_D int texel_wrap(uint32_t address, int v, int n, bool &zero) {
    if (address == LR_TEX_ADDR_EDGE) { return min(max(v, 0), n - 1); }
    if (address == LR_TEX_ADDR_ZERO) {
        if (v < 0 || v >= n) { zero = true; return 0; }
        return v;
    }
    const auto period = address == LR_TEX_ADDR_MIRROR ? 2 * n : n;
    const auto inv_period = 1.f / static_cast<float>(period);
    auto m = v;
    if (!(v > -(1 << 20) && v < (1 << 20))) {// (rare: texel coordinates in the millions; the products wrap around consistently)
        const auto q = static_cast<int>(floorf(static_cast<float>(v) * inv_period));
        m = static_cast<int>(static_cast<uint32_t>(v) - static_cast<uint32_t>(q) * static_cast<uint32_t>(period));
    }
    const auto q = static_cast<int>(floorf(static_cast<float>(m) * inv_period));
    m -= q * period;// off by at most one period when the quotient is within rounding of an integer
    m += m < 0 ? period : 0;
    m -= m >= period ? period : 0;
    return address == LR_TEX_ADDR_MIRROR && m >= n ? period - 1 - m : m;
}
// pow(c, g) of a texel that is not positive (a negative or zero value under a gamma encoding: practically never), by the cases of
// IEEE pow instead of libm's general 158 instructions at every decode site
LR_D float pow_nonpositive(float c, float g) {
    if (c != c || g != g) { return c + g; }
    if (c == 0.f) { return g > 0.f ? 0.f : (g == 0.f ? 1.f : __builtin_inff()); }
    const auto gi = truncf(g);
    if (gi != g) { return __builtin_nanf(""); }// a negative base with a fractional exponent
    const auto magnitude = __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(-c));
    const auto odd = fabsf(gi) < 16777216.f && (static_cast<int>(gi) & 1) != 0;
    return odd ? -magnitude : magnitude;
}
// EIGHT-BIT TEXELS STAY EIGHT BITS ON THE DEVICE (round 5).  The host decodes every image to float RGBA (what the reference's textures hold
// and what the oracle reads); an image whose every texel is an 8-bit code's float -- a PNG / JPEG / BMP / TGA albedo or roughness map --
// is uploaded as one 32-bit word per texel instead of sixteen bytes (lrhip.hip: pack_byte_textures), behind the float texels in the same
// buffer, and decoded here to EXACTLY the floats the host made: `pad` bits 0-1 say how the host converted (1: b * (1 / 255.f), the PNG
// reader; 2: b / 255.f, the other readers -- byte_over_255 below, checked against the division for all 256 codes at upload), bits 4-7
// mark channels that hold ONE value over the whole image (a padded alpha of 1: lr_texture::v carries it).  A quarter of the
// footprint: the camera-class stand-in's eight 2k x 2k maps are 128 MB instead of 512 MB, and the frame runs 3.5-4 % faster
// (profiles/r05zd_byte_textures.txt; profiles/r05zb_c4_texture_size.txt: the same frame with smaller images).  Which scenes get it:
// lrhip_set_texture_storage (lrhip.h).
// (compiled into the variants that make real calls, the heavy kernels and the lean kernels of the kFeatByteTex bit: dev_wavefront.h)
#ifndef LR_BYTE_TEXELS
#if defined(LR_VARIANT) && !((LR_VARIANT) & (96 | 256 | 8192))
#define LR_BYTE_TEXELS 0
#else
#define LR_BYTE_TEXELS 1
#endif
#endif
LR_HD float byte_over_255(float b) {// correctly rounded b / 255.f for b = 0 .. 255: one Newton step on the reciprocal product
    const auto q = b * (1.f / 255.f);
    const auto r = fmaf(-q, 255.f, b);
    return fmaf(r, 1.f / 255.f, q);
}
// LR_TEX_WIDE_RECORD: the out-of-line lookup reads a texture's record in one round trip and everything through explicitly global loads (texture_record below).
// Measured (profiles/r06u_texture_record_loads.txt, films bit-identical): camera class <12308> 1146 -> 1163 Msamples/s at 64 spp; the kitchen
// class' lean wavefront passes LOSE 0.4 % (the looked-up record's 28 registers across the call: 32 -> 48 spilled VGPRs in <5128>) and keep the
// field-by-field reads.  The lookup as a free function with its arguments by value instead of a capturing lambda (LR_TEX_BY_VALUE, dev_heavy.h):
// camera class -2.7 %, kitchen class -1 %: not kept.
#ifndef LR_TEX_WIDE_RECORD
#if defined(LR_VARIANT) && ((LR_VARIANT) & 16) && !((LR_VARIANT) & (96 | 256))
#define LR_TEX_WIDE_RECORD 1// the lean kernels with the Disney closure: what textured scenes of the camera class' kind render on
#else
#define LR_TEX_WIDE_RECORD 0
#endif
#endif
// sixteen bytes from GLOBAL memory, said so
typedef float lr_v4f __attribute__((ext_vector_type(4)));
LR_D float4 global_load_f4(const void *base, uint64_t index) {
    const auto v = ((const __attribute__((address_space(1))) lr_v4f *)base)[index];
    return make_float4(v.x, v.y, v.z, v.w);
}
LR_D float4 texel_at(const float *texels, const lr_texture &t, int xx, int yy) {
    const auto index = static_cast<uint64_t>(yy) * t.width + static_cast<uint64_t>(xx);
    // (explicitly global: behind an out-of-line lookup the compiler knows nothing of the pointer's origin and would read through flat_load)
#if LR_TEX_WIDE_RECORD
    typedef const __attribute__((address_space(1))) uint32_t global_u32;
    if (!LR_BYTE_TEXELS || t.pad == 0u) { return global_load_f4(texels, t.texel_offset + index); }
    const auto p = ((global_u32 *)texels)[t.texel_offset + index];
#else
    if (!LR_BYTE_TEXELS || t.pad == 0u) { return reinterpret_cast<const float4 *>(texels)[t.texel_offset + index]; }
    const auto p = reinterpret_cast<const uint32_t *>(texels)[t.texel_offset + index];
#endif
    // b * (1 / 255.f) is byte_over_255 without its correction step: one expression for both forms, the correction's weight 0 under form 1.
    // (Measured on the camera-class frame, films bit-identical: this against a select between the two forms +0.5 %, the texels in tiles of
    // 8 x 4 -- one 128-byte line per tile -- instead of rows +0.4 %, both +1.0 %, a repeat of the base +0.4 %: the tiles were not kept,
    // profiles/r05ze_texel_layout_ab.txt.)
    const auto k = (t.pad & 3u) == 1u ? 0.f : 1.f / 255.f;
    auto code = [&](uint32_t b, uint32_t c) {
        const auto f = static_cast<float>(b);
        const auto q = f * (1.f / 255.f);
        const auto v = fmaf(fmaf(-q, 255.f, f), k, q);
        return (t.pad & (16u << c)) != 0u ? t.v[c] : v;
    };
    // (x, y, z only: no consumer on the device reads a texture's fourth channel -- closure parameters, emission, normal maps take xyz, opacity
    // and scalars x -- and behind the out-of-line lookup the compiler cannot know that; round 6)
    return make_float4(code(p & 255u, 0u), code((p >> 8u) & 255u, 1u), code((p >> 16u) & 255u, 2u), 0.f);
}
LR_D float4 texel_fetch(const float *texels, const lr_texture &t, int x, int y) {
    auto zero = false;
    const auto xx = texel_wrap(t.address, x, static_cast<int>(t.width), zero), yy = texel_wrap(t.address, y, static_cast<int>(t.height), zero);
    if (zero) { return make_float4(0.f, 0.f, 0.f, 0.f); }
    return texel_at(texels, t, xx, yy);
}

// Texture::Instance::evaluate (texture.cpp:21-79; image.cpp:132-168; checkerboard.cpp).  A REAL call (LR_CALL): one
// copy of the format / address-mode / decode switches per kernel instead of one per use (it was 17 KB of code inlined
// at every site: the <environment> variant carried eight copies), and its registers are not the megakernel's.
// A texture's whole record in ONE round trip: seven 16-byte global loads issued together (round 6).  Read field by field where each was
// used -- kind, then size and uv transform, then filter, address mode, encoding -- the out-of-line lookup waited for six to eight DEPENDENT
// loads in a row before its texels were even requested (and through flat_load: see texel_at).
static_assert(sizeof(lr_texture) == 112 && alignof(lr_texture) == 8, "lr_texture: seven 16-byte words (the table's base is 256-byte aligned)");
LR_D lr_texture texture_record(const lr_texture *textures, int32_t id) {
    float4 w[7];
#pragma unroll
    for (auto i = 0; i < 7; i++) { w[i] = global_load_f4(textures + id, static_cast<uint64_t>(i)); }
    lr_texture t;
    __builtin_memcpy(&t, w, sizeof(t));
    return t;
}
LR_CALL float4 texture_eval_tables(const lr_texture *textures, const float *texels, int32_t id, f2 uv_it) {
#if LR_TEX_WIDE_RECORD
    auto ti = texture_record(textures, id);
    if (ti.kind == LR_TEX_CONSTANT) { return make_float4(ti.v[0], ti.v[1], ti.v[2], ti.v[3]); }
    if (ti.kind == LR_TEX_CHECKERBOARD) {
        auto parity = (static_cast<int>(floorf(uv_it.x * ti.checker_scale)) + static_cast<int>(floorf(uv_it.y * ti.checker_scale))) & 1;
        auto child = parity ? ti.child[1] : ti.child[0];
        if (child < 0) { return parity ? make_float4(0.f, 0.f, 0.f, 1.f) : make_float4(1.f, 1.f, 1.f, 1.f); }
        ti = texture_record(textures, child);// one level of nesting: children are constant or image
        if (ti.kind == LR_TEX_CONSTANT) { return make_float4(ti.v[0], ti.v[1], ti.v[2], ti.v[3]); }
    }
#else
    auto &t = textures[id];
    if (t.kind == LR_TEX_CONSTANT) { return make_float4(t.v[0], t.v[1], t.v[2], t.v[3]); }
    if (t.kind == LR_TEX_CHECKERBOARD) {
        auto parity = (static_cast<int>(floorf(uv_it.x * t.checker_scale)) + static_cast<int>(floorf(uv_it.y * t.checker_scale))) & 1;
        auto child = t.child[parity ? 1 : 0];
        if (child < 0) { return parity ? make_float4(0.f, 0.f, 0.f, 1.f) : make_float4(1.f, 1.f, 1.f, 1.f); }
        auto &c = textures[child];// one level of nesting: children are constant or image
        if (c.kind == LR_TEX_CONSTANT) { return make_float4(c.v[0], c.v[1], c.v[2], c.v[3]); }
        id = child;
    }
    auto &ti = textures[id];
#endif
    f2 uv{uv_it.x * ti.uv_scale[0] + ti.uv_offset[0], uv_it.y * ti.uv_scale[1] + ti.uv_offset[1]};
    float4 v;
    if (ti.filter == LR_TEX_FILTER_POINT) {
        v = texel_fetch(texels, ti, static_cast<int>(floorf(uv.x * static_cast<float>(ti.width))),
                        static_cast<int>(floorf(uv.y * static_cast<float>(ti.height))));
    } else {
        auto fx = uv.x * static_cast<float>(ti.width) - 0.5f, fy = uv.y * static_cast<float>(ti.height) - 0.5f;
        auto x0 = floorf(fx), y0 = floorf(fy);
        auto tx = fx - x0, ty = fy - y0;
        auto ix = static_cast<int>(x0), iy = static_cast<int>(y0);
        // the four texels share two columns and two rows: four coordinate wraps, not eight
        const auto tw = static_cast<int>(ti.width), th = static_cast<int>(ti.height);
        bool zx0 = false, zx1 = false, zy0 = false, zy1 = false;
        const auto xa = texel_wrap(ti.address, ix, tw, zx0), xb = texel_wrap(ti.address, ix + 1, tw, zx1);
        const auto ya = texel_wrap(ti.address, iy, th, zy0), yb = texel_wrap(ti.address, iy + 1, th, zy1);
        const auto none = make_float4(0.f, 0.f, 0.f, 0.f);
        auto c00 = zx0 || zy0 ? none : texel_at(texels, ti, xa, ya), c10 = zx1 || zy0 ? none : texel_at(texels, ti, xb, ya);
        auto c01 = zx0 || zy1 ? none : texel_at(texels, ti, xa, yb), c11 = zx1 || zy1 ? none : texel_at(texels, ti, xb, yb);
        auto mix = [&](float a, float b, float c, float d) {
            return (a * (1.f - tx) + b * tx) * (1.f - ty) + (c * (1.f - tx) + d * tx) * ty;
        };
        v = make_float4(mix(c00.x, c10.x, c01.x, c11.x), mix(c00.y, c10.y, c01.y, c11.y), mix(c00.z, c10.z, c01.z, c11.z), 0.f);
    }
    auto decode = [&](float c, int ch) {// image.cpp:138-153
        // x^y through the hardware's log2 / exp2 (v_log_f32 / v_exp_f32, ~1 ulp each) instead of powf (158 instructions with the
        // project's flags, four of them per lookup); x^2.4 = x^2 * x^0.4 keeps the exponent error small: ~3e-7 relative
        if (ti.encoding == LR_TEX_ENC_SRGB) {
            const auto x = (c + 0.055f) * (1.0f / 1.055f);
            c = c <= 0.04045f ? c * (1.0f / 12.92f) : x * x * __builtin_amdgcn_exp2f(0.4f * __builtin_amdgcn_logf(x));
        } else if (ti.encoding == LR_TEX_ENC_GAMMA) {
            const auto g = ti.gamma[min(ch, 2)];
            c = c > 0.f ? __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(c)) : pow_nonpositive(c, g);
        }
        return ti.scale[ch] * c;
    };
    return make_float4(decode(v.x, 0), decode(v.y, 1), decode(v.z, 2), 0.f);// (the fourth channel of an image: never read on the device, see texel_at)
}
// the lookup load_lobe makes per looked-up slot (dev_heavy.h): out of line in the lean variants too (dev_math.h: LR_TEX_LAMBDA), with its
// arguments BY VALUE -- as a capturing lambda it read the tables' pointers and uv back from the closure object through flat loads
LR_TEX_LAMBDA float4 texture_eval_slot(const lr_texture *textures, const float *texels, int32_t id, float u, float v) {
    return texture_eval_tables(textures, texels, id, f2{u, v});
}
LR_D float4 texture_eval(const DScene &scene, int32_t id, f2 uv_it) { return texture_eval_tables(scene.textures, scene.texels, id, uv_it); }

// ---------------------------------------------------------------- closure resolution
// Texture access is abstracted so the same code folds constants on the host at upload time
// (TexFn reads lr_texture::v) and evaluates image textures per hit on the device.

LR_HD f3 extend_rgb(float4 c, uint32_t n) {// texture.cpp:14-18
    if (n == 1u) { return mk3(c.x, c.x, c.x); }
    if (n == 2u) { return mk3(c.x, c.y, 1.f); }
    return mk3(c.x, c.y, c.z);
}

template<typename TexFn, typename ChannelsFn>
LR_HD DClosure resolve_closure(const lr_surface &s, TexFn &&tex, ChannelsFn &&channels, float eta_i) {
    DClosure c{};
    c.kind = s.kind;
    auto albedo = [&](int slot, float dv, f3 &value, float &strength) {// evaluate_albedo_spectrum + srgb decode
        if (s.tex[slot] < 0) { value = mk3(dv), strength = dv; return; }
        value = saturate(extend_rgb(tex(slot), channels(slot)));
        strength = cie_y(value);
    };
    auto alpha = [&](int slot, float dv) {// e.g. mirror.cpp:145-154
        if (s.tex[slot] < 0) { c.alpha_x = c.alpha_y = dv; return; }
        auto r = tex(slot);
        auto remap = (s.flags & LR_SURFACE_FLAG_REMAP_ROUGHNESS) != 0u;
        if (channels(slot) == 1u) { c.alpha_x = c.alpha_y = remap ? roughness_to_alpha(r.x) : r.x; }
        else { c.alpha_x = remap ? roughness_to_alpha(r.x) : r.x, c.alpha_y = remap ? roughness_to_alpha(r.y) : r.y; }
    };
    auto store = [](float *dst, f3 v) { dst[0] = v.x, dst[1] = v.y, dst[2] = v.z; };
    f3 v;
    float strength;
    switch (s.kind) {
        case LR_SURFACE_MATTE: {// matte.cpp:119-134
            albedo(0, 1.f, v, strength);
            store(c.c0, v);
            // the host drops a black sigma texture (`_sigma && !_sigma->node()->is_black()`, matte.cpp:125)
            c.s0 = s.tex[1] >= 0 ? saturate(tex(1).x) * 90.f : 0.f;
            break;
        }
        case LR_SURFACE_MIRROR: {// mirror.cpp:141-163
            alpha(1, 0.f);
            albedo(0, 1.f, v, strength);
            store(c.c0, v);
            break;
        }
        case LR_SURFACE_GLASS: {// glass.cpp:232-285 (fixed spectrum: eta = first channel)
            alpha(2, 0.f);
            float kr_lum, kt_lum;
            albedo(0, 1.f, v, kr_lum);
            store(c.c0, v);
            albedo(1, 1.f, v, kt_lum);
            store(c.c1, v);
            c.s2 = kr_lum == 0.f ? 0.f : kr_lum / (kr_lum + kt_lum);
            c.s0 = eta_i;
            c.s1 = s.tex[3] >= 0 ? tex(3).x : 1.5f;
            break;
        }
        case LR_SURFACE_PLASTIC: {// plastic.cpp:256-291
            alpha(1, 0.f);
            auto eta = (s.tex[3] >= 0 ? tex(3).x : 1.5f) / eta_i;
            f3 kd, sigma_a;
            float kd_lum, sigma_lum;
            albedo(0, 1.f, kd, kd_lum);
            albedo(2, 0.f, sigma_a, sigma_lum);
            auto thickness = s.tex[4] >= 0 ? tex(4).x : 1.f;
            auto average_transmittance = expf(-2.f * sigma_lum * thickness);
            auto diffuse_fresnel = fresnel_dielectric_integral(eta);
            store(c.c0, kd / (mk3(1.f) - kd * diffuse_fresnel));
            c.s0 = kd_lum * average_transmittance;
            store(c.c1, sigma_a);
            c.s1 = eta;
            break;
        }
        case LR_SURFACE_METAL: {// metal.cpp:273-308
            alpha(1, .5f);
            store(c.c0, mk3(s.f[0], s.f[1], s.f[2]));
            store(c.c1, mk3(s.f[3], s.f[4], s.f[5]));
            if (s.tex[0] >= 0) { albedo(0, 1.f, v, strength); } else { v = mk3(1.f); }
            store(c.c2, v);
            c.s0 = eta_i;
            break;
        }
        case LR_SURFACE_DISNEY: {// disney.cpp:932-1003
            albedo(0, 1.f, v, strength);
            store(c.c0, v);
            c.s0 = strength;
            auto scalar = [&](int slot, float dv) { return s.tex[slot] >= 0 ? tex(slot).x : dv; };
            c.e[kDisneyMetallic] = scalar(1, 0.f);
            c.e[kDisneyEtaI] = eta_i, c.e[kDisneyEtaT] = scalar(2, 1.5f);
            auto roughness = scalar(3, .5f);
            if (s.flags & LR_SURFACE_FLAG_REMAP_ROUGHNESS) { roughness = roughness_to_alpha(roughness); }
            c.e[kDisneyRoughness] = roughness;
            c.e[kDisneySpecularTint] = scalar(4, 0.f), c.e[kDisneyAnisotropic] = scalar(5, 0.f);
            c.e[kDisneySheen] = scalar(6, 0.f), c.e[kDisneySheenTint] = scalar(7, 0.f);
            c.e[kDisneyClearcoat] = scalar(8, 0.f), c.e[kDisneyClearcoatGloss] = scalar(9, 1.f);
            c.e[kDisneySpecularTrans] = scalar(10, 0.f), c.e[kDisneyFlatness] = scalar(11, 0.f);
            c.s1 = scalar(12, 0.f);// diffuse_trans
            c.x[0] = s.u[0], c.x[1] = s.u[1], c.x[2] = (s.flags & LR_SURFACE_FLAG_THIN) ? 1u : 0u;
            break;
        }
        case LR_SURFACE_LAYERED: {// layered.cpp:478-500
            c.s0 = s.tex[0] >= 0 ? fmaxf(tex(0).x, 1.17549435e-38f) : 1e-2f;
            c.s1 = s.tex[1] >= 0 ? tex(1).x : 0.f;
            albedo(2, 1.f, v, strength);
            store(c.c0, v);
            c.x[0] = s.u[0], c.x[1] = s.u[1], c.x[2] = s.u[2], c.x[3] = s.u[3];
            break;
        }
        case LR_SURFACE_MIX: {// mix.cpp:198-212
            c.s0 = s.tex[0] >= 0 ? clampf(tex(0).x, 0.f, 1.f) : 0.5f;
            c.x[0] = s.u[0], c.x[1] = s.u[1], c.x[2] = s.u[2];// children, nesting depth below this node
            break;
        }
        default: c.kind = LR_SURFACE_NULL; break;
    }
    return c;
}

// ---------------------------------------------------------------- hit reconstruction

struct SurfacePoint {// the subset of the reference's Interaction the hot path reads
    f3 p, ng;
    Frame shading;
    f2 uv;
    float area;
    uint32_t flags;       // Shape property flags
    uint32_t tags;        // handle.y
    uint32_t offset_bits; // handle.w
    uint32_t tri_offset;  // mesh triangle slice (pdf / alias tables)
    bool back_facing;
};

LR_D float intersection_offset_factor(uint32_t handle_w) {// shape.cpp:88-93
    auto x = static_cast<float>(handle_w & 0xffffu) * (1.0f / 65536.f);
    return clampf(x * 255.f + 1.f, 1.f, 256.f);
}

// Geometry::shading_point, geometry.cpp:345-389.  FULL = false computes only what a sampled light
// point needs (p, ng, area, uv) with the same arithmetic.
template<bool FULL>
LR_D void reconstruct(const DScene &scene, uint32_t inst_id, uint32_t prim, f3 bary, SurfacePoint &sp) {
    auto ip = reinterpret_cast<const float4 *>(scene.instances + inst_id);
    auto h = reinterpret_cast<const uint4 *>(ip)[0];
    auto q0 = ip[1], q1 = ip[2], q2 = ip[3], q3 = ip[4];
    auto vertex_offset = __float_as_uint(q0.w), tri_offset = __float_as_uint(q1.w);
    auto tri = scene.triangles[tri_offset + prim];
    auto vp = reinterpret_cast<const float4 *>(scene.vertices + vertex_offset);
    auto a0 = vp[tri.i0 * 2u], a1 = vp[tri.i0 * 2u + 1u];
    auto b0 = vp[tri.i1 * 2u], b1 = vp[tri.i1 * 2u + 1u];
    auto c0 = vp[tri.i2 * 2u], c1 = vp[tri.i2 * 2u + 1u];
    f3 m0 = mk3(q0.x, q0.y, q0.z), m1 = mk3(q1.x, q1.y, q1.z), m2 = mk3(q2.x, q2.y, q2.z), mt = mk3(q3.x, q3.y, q3.z);
    auto mul = [&](f3 v) { return m0 * v.x + m1 * v.y + m2 * v.z; };
    f3 p0 = mk3(a0.x, a0.y, a0.z), p1 = mk3(b0.x, b0.y, b0.z), p2 = mk3(c0.x, c0.y, c0.z);
    f2 uv0{a1.z, a1.w}, uv1{b1.z, b1.w}, uv2{c1.z, c1.w};
    auto dp0 = p1 - p0, dp1 = p2 - p0;
    sp.p = mul(bary.x * p0 + bary.y * p1 + bary.z * p2) + mt;
    auto c = cross(mul(dp0), mul(dp1));
    sp.area = length(c) * .5f;
    sp.ng = normalize(c);
    sp.flags = h.x & 1023u;
    sp.tags = h.y;
    sp.offset_bits = h.w;
    sp.tri_offset = tri_offset;
    sp.uv = (sp.flags & LR_SHAPE_HAS_VERTEX_UV) ?
                f2{bary.x * uv0.x + bary.y * uv1.x + bary.z * uv2.x, bary.x * uv0.y + bary.y * uv1.y + bary.z * uv2.y} :
                f2{bary.y, bary.z};
    if (FULL) {
        auto r0 = ip[5], r1 = ip[6], r2 = ip[7];
        f3 n0 = mk3(a0.w, a1.x, a1.y), n1 = mk3(b0.w, b1.x, b1.y), n2 = mk3(c0.w, c1.x, c1.y);
        auto ns_local = bary.x * n0 + bary.y * n1 + bary.z * n2;
        f2 duv0{uv1.x - uv0.x, uv1.y - uv0.y}, duv1{uv2.x - uv0.x, uv2.y - uv0.y};
        auto det = duv0.x * duv1.y - duv0.y * duv1.x;
        auto inv_det = 1.f / det;
        auto dpdu_local = (dp0 * duv1.y - dp1 * duv0.y) * inv_det;
        auto fallback = frame_from_normal(sp.ng);
        auto dpdu = det == 0.f ? fallback.s : mul(dpdu_local);
        auto ns = (sp.flags & LR_SHAPE_HAS_VERTEX_NORMAL) ?
                      normalize(mk3(r0.x, r0.y, r0.z) * ns_local.x + mk3(r1.x, r1.y, r1.z) * ns_local.y + mk3(r2.x, r2.y, r2.z) * ns_local.z) :
                      sp.ng;
        sp.shading = frame_from_normal_tangent(face_forward(ns, sp.ng), dpdu);
    }
}

// Geometry::shading_point (geometry.cpp:345-389) of a traversal hit from the baked triangle's shading record: the same
// quantities as reconstruct<true> (p, ng, area, uv, shading frame with the dpdu tangent) from ONE 128-byte gather instead of
// three dependent ones.  World-space arithmetic (the record holds M p_i and M^-T n_i), so results differ from the
// object-space form in the last bits only.
LR_D void reconstruct_baked(const DScene &scene, uint32_t tri, float u, float v, SurfacePoint &sp) {
    auto q = reinterpret_cast<const float4 *>(scene.shade_tris + tri);
    // (plain loads.  MEASURED, NOT KEPT: non-temporal loads of this one-touch record, so that it does not push BVH levels out of the L1 -- C2 854 -> 823 Msamples/s, round 4)
    auto q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
    auto w = 1.f - u - v;
    f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    sp.p = p0 + e1 * u + e2 * v;
    auto c = cross(e1, e2);
    sp.area = length(c) * .5f;
    sp.ng = normalize(c);
    sp.flags = __float_as_uint(q0.w);
    sp.tags = __float_as_uint(q1.w);
    sp.offset_bits = __float_as_uint(q2.w);
    sp.tri_offset = reinterpret_cast<const uint32_t *>(scene.shade_tris + tri)[30];
    f2 uv0{q3.w, q4.w}, uv1{q5.w, q6.x}, uv2{q6.y, q6.z};
    sp.uv = (sp.flags & LR_SHAPE_HAS_VERTEX_UV) ? f2{w * uv0.x + u * uv1.x + v * uv2.x, w * uv0.y + u * uv1.y + v * uv2.y} : f2{u, v};
    f2 duv0{uv1.x - uv0.x, uv1.y - uv0.y}, duv1{uv2.x - uv0.x, uv2.y - uv0.y};
    auto det = duv0.x * duv1.y - duv0.y * duv1.x;
    auto inv_det = 1.f / det;
    auto fallback = frame_from_normal(sp.ng);
    auto dpdu = det == 0.f ? fallback.s : (e1 * duv1.y - e2 * duv0.y) * inv_det;
    auto ns = (sp.flags & LR_SHAPE_HAS_VERTEX_NORMAL) ?
                  normalize(mk3(q3.x, q3.y, q3.z) * w + mk3(q4.x, q4.y, q4.z) * u + mk3(q5.x, q5.y, q5.z) * v) :
                  sp.ng;
    sp.shading = frame_from_normal_tangent(face_forward(ns, sp.ng), dpdu);
}



// Surface::Instance::evaluate_opacity: OpacitySurfaceWrapper (surface.h:183-189) and MixSurfaceInstance (mix.cpp:63-70)
LR_D float surface_opacity(const DScene &scene, uint32_t tag, f2 uv) {
    auto one = [&](uint32_t t) {
        auto alpha_tex = scene.surfaces[t].raw.alpha_tex;
        return alpha_tex >= 0 ? texture_eval(scene, alpha_tex, uv).x : 1.f;
    };
    if (scene.surfaces[tag].raw.kind != LR_SURFACE_MIX) { return one(tag); }
    // a Mix tree (children may be Mix surfaces, at most 3 levels below the root): the product over its leaves, multiplied in
    // the order the recursion a.opacity * b.opacity visits them
    uint32_t stack[8];
    auto sp = 0u;
    stack[sp++] = tag;
    auto opacity = 1.f;
    while (sp > 0u) {
        const auto tag_here = stack[--sp];
        auto &s = scene.surfaces[tag_here].raw;
        if (s.kind == LR_SURFACE_MIX && sp + 2u <= 8u) { stack[sp++] = s.u[1], stack[sp++] = s.u[0]; }
        else { opacity *= one(tag_here); }
    }
    return opacity;
}

// Geometry::_alpha_skip, geometry.cpp:165-192: a candidate hit is skipped when
// xxhash32(inst, prim, bits(bary)) * 2^-32 > opacity(uv).  Called only for instances flagged maybe_non_opaque.
LR_D bool alpha_skip(const DScene &scene, uint32_t inst_id, uint32_t prim, float u, float v) {
    auto ip = reinterpret_cast<const float4 *>(scene.instances + inst_id);
    auto h = reinterpret_cast<const uint4 *>(ip)[0];
    f2 uv{u, v};
    if (h.x & LR_SHAPE_HAS_VERTEX_UV) {
        auto vertex_offset = __float_as_uint(ip[1].w), tri_offset = __float_as_uint(ip[2].w);
        auto tri = scene.triangles[tri_offset + prim];
        auto vp = reinterpret_cast<const float4 *>(scene.vertices + vertex_offset);
        auto a1 = vp[tri.i0 * 2u + 1u], b1 = vp[tri.i1 * 2u + 1u], c1 = vp[tri.i2 * 2u + 1u];
        auto w = 1.f - u - v;
        uv = f2{w * a1.z + u * b1.z + v * c1.z, w * a1.w + u * b1.w + v * c1.w};
    }
    auto xi = static_cast<float>(xxhash32_4(inst_id, prim, __float_as_uint(u), __float_as_uint(v))) * 0x1p-32f;
    return xi > surface_opacity(scene, (h.y >> 12u) & 4095u, uv);
}

// The alpha test of the candidates the traversal loop parked (dev_trace.h kPhasePendingAlpha): commit the hit or skip it, then
// move on to the next stack entry exactly as the leaf step would have.
LR_D void resolve_pending_alpha(const DScene &scene, const TraversalStack &stack, TravState &tr) {
    if ((tr.phase & kPhasePendingAlpha) != 0u) {
        const auto phase = tr.phase & ~kPhasePendingAlpha;
        const auto tri_index = tr.cur & kLeafIndexMask;
        const auto tb = reinterpret_cast<const float4 *>(scene.bvh_tris) + static_cast<size_t>(tri_index) * 3u;
        const auto inst = __float_as_uint(tb[0].w), prim = __float_as_uint(tb[1].w);
        if (!alpha_skip(scene, inst, prim, tr.pend_u, tr.pend_v)) {
            tr.t_max = tr.pend_t;
            if (phase == kPhaseClosest) {
                tr.hit.inst = inst, tr.hit.prim = prim, tr.hit.u = tr.pend_u, tr.hit.v = tr.pend_v, tr.hit.tri = tri_index;
            } else {
                tr.occluded = true;
                tr.sp = 0u;// any-hit: drop the rest of the stack
            }
        }
        tr.cur = tr.sp > 0u ? stack.pop(--tr.sp) : kInvalid;
        tr.phase = phase;
    }
}

// Runs the wave's traversal until `refill` lanes have results to shade (or nothing is in flight), resolving parked alpha
// candidates in between: the shading block is only entered for the reasons it was entered before.
template<bool COUNT, bool ALPHA>
LR_D void trace_until_refill(const DScene &scene, const TraversalStack &stack, TravState &tr, bool has_next, const Ray &next_closest,
                             int refill, TraceStats &stats) {
    const auto idle_at_entry = tr.phase == kPhaseIdle;
    for (;;) {
        trace_steps<COUNT, ALPHA>(scene, stack, tr, has_next, next_closest, refill, stats, idle_at_entry);
        if (!ALPHA) { break; }
        // (candidates wait in batches, dev_trace.h: LR_ALPHA_BATCH; the wave may also have left for the shading block with some still waiting:
        // they are resolved either way, and trace_steps returns at once when its reason to leave still stands)
        if (!lr_any((tr.phase & kPhasePendingAlpha) != 0u)) { break; }
        resolve_pending_alpha(scene, stack, tr);
    }
}

// LuisaCompute `offset_ray_origin` (Ray Tracing Gems ch. 6), restated from the published algorithm
LR_D f3 offset_ray_origin(f3 p, f3 n) {
    constexpr auto origin = 1.0f / 32.0f;
    constexpr auto float_scale = 1.0f / 65536.0f;
    constexpr auto int_scale = 256.0f;
    auto one = [&](float pc, float nc) {
        auto of_i = static_cast<int>(int_scale * nc);
        auto p_i = __int_as_float(__float_as_int(pc) + (pc < 0.f ? -of_i : of_i));
        return fabsf(pc) < origin ? pc + float_scale * nc : p_i;
    };
    return mk3(one(p.x, n.x), one(p.y, n.y), one(p.z, n.z));
}
LR_D f3 robust_origin(const SurfacePoint &sp, f3 w) {// Interaction::p_robust, interaction.cpp:13-19
    auto front = dot(sp.shading.n, w) > 0.f;
    auto n = front ? sp.ng : -sp.ng;
    return offset_ray_origin(sp.p, intersection_offset_factor(sp.offset_bits) * n);
}

// DiffuseLightClosure::_evaluate, diffuse.cpp:67-88
LR_D void light_evaluate_with(const DScene &scene, const SurfacePoint &lp, float tri_pdf, f3 p_from, f3 &L, float &pdf);
LR_D void light_evaluate(const DScene &scene, const SurfacePoint &lp, uint32_t prim, f3 p_from, f3 &L, float &pdf) {
    light_evaluate_with(scene, lp, scene.tri_pdf[lp.tri_offset + prim], p_from, L, pdf);
}
LR_D void light_evaluate_with(const DScene &scene, const SurfacePoint &lp, float tri_pdf, f3 p_from, f3 &L, float &pdf) {
    auto &light = scene.lights[lp.tags & 4095u];
    auto pdf_area = tri_pdf / lp.area;
    auto cos_wo = abs_dot(normalize(p_from - lp.p), lp.ng);
    f3 Le = mk3(light.L[0], light.L[1], light.L[2]);
    if (light.dynamic) {// evaluate_illuminant_spectrum of a non-constant texture: xyz as-is, clamp >= 0
        auto v = texture_eval(scene, light.emission_tex, lp.uv);
        Le = max0(mk3(v.x, v.y, v.z)) * light.scale;
    }
    auto diff = lp.p - p_from;
    auto p = dot(diff, diff) * pdf_area * (1.0f / cos_wo);
    auto invalid = fabsf(cos_wo) < 1e-6f || (!light.two_sided && lp.back_facing);
    L = invalid ? mk3(0.f) : Le;
    pdf = invalid ? 0.f : p;
}

// ---------------------------------------------------------------- environments (FULL kernels)

LR_D f3 mul3(const float *m, f3 v) { return mk3(m[0], m[1], m[2]) * v.x + mk3(m[3], m[4], m[5]) * v.y + mk3(m[6], m[7], m[8]) * v.z; }

struct EnvTables {// what the environment code reads of the scene (passed by value to the out-of-line functions)
    const lr_texture *textures;
    const float *texels;
};
struct EnvEval {
    f3 L;
    float pdf;
};
struct EnvSample {
    f3 wi, L;
    float pdf;
};

LR_D f3 env_radiance(const EnvTables &tb, const DEnvironment &env, f2 uv) {// evaluate_illuminant_spectrum(...).value * scale
    auto v = texture_eval_tables(tb.textures, tb.texels, env.emission_tex, uv);
    auto rgb = env.constant_emission ? extend_rgb(v, tb.textures[env.emission_tex].channels) : mk3(v.x, v.y, v.z);
    return max0(rgb);
}
LR_D float env_directional_pdf(float p, float theta) {// SphericalInstance::_directional_pdf, spherical.cpp:76-80
    float sn, cs;
    sincos_2pi(theta * (0.5f * kInvPi), sn, cs);// theta in [0, pi]
    auto inv_s = sn > 0.f ? 1.f / sn : 0.f;
    return p * inv_s * (.5f * kInvPi * kInvPi);
}
LR_D EnvEval env_directional(const EnvTables &tb, const DEnvironment &env, f3 wi_local) {// directional.cpp:64-73
    auto valid = env.cos_half_angle < wi_local.z;
    return EnvEval{env_radiance(tb, env, f2{.5f, .5f}) * (valid ? env.scale : 0.f), valid ? 1.f / (2.f * kPi * (1.f - env.cos_half_angle)) : 0.f};
}
// Environment::Instance::evaluate of one Spherical / Directional record: spherical.cpp:88-108, directional.cpp:80-88.
// Out of line (LR_CALL): reached once per miss / environment NEE sample, and acosf / atan2f / the texture lookup are
// large; inlined at every use (root, both Combined children, the evaluate-the-other-child of Combined::sample) the
// environment code alone was 320 KB of the kernel.
LR_CALL EnvEval env_evaluate_one(EnvTables tb, const DEnvironment *envp, f3 wi) {
    auto &env = *envp;
    if (env.kind == kEnvDirectional) {
        if (!env.visible) { return EnvEval{mk3(0.f), 0.f}; }
        auto frame = frame_from_normal(mk3(env.direction[0], env.direction[1], env.direction[2]));
        return env_directional(tb, env, normalize(to_local(frame, mul3(env.world_to_env, wi))));
    }
    auto w = normalize(mul3(env.world_to_env, wi));
    auto theta = acosf(w.y), phi = atan2f(w.x, w.z);// Spherical::direction_to_uv
    f2 uv{fract(1.f - 0.5f * kInvPi * phi), fract(theta * kInvPi)};
    EnvEval r;
    r.L = env_radiance(tb, env, uv) * env.scale;
    if (env.kind == kEnvConstant) { r.pdf = kInvPi * 0.25f; return r; }
    auto sx = static_cast<float>(env.map_width), sy = static_cast<float>(env.map_height);
    auto ix = static_cast<uint32_t>(clampf(uv.x * sx, 0.f, sx - 1.f)), iy = static_cast<uint32_t>(clampf(uv.y * sy, 0.f, sy - 1.f));
    r.pdf = env_directional_pdf(env.pdf[iy * env.map_width + ix], theta);
    return r;
}
// Environment::Instance::sample of one record: spherical.cpp:110-141, directional.cpp:90-98
LR_CALL EnvSample env_sample_one(EnvTables tb, const DEnvironment *envp, f2 u) {
    auto &env = *envp;
    EnvSample r;
    if (env.kind == kEnvDirectional) {
        auto cos_t = (1.f - u.x) + u.x * env.cos_half_angle;// sample_uniform_cone
        auto sin_t = sqrtf(fmaxf(1.f - cos_t * cos_t, 0.f));
        float sn, cs;
        sincos_2pi(u.y, sn, cs);// phi = 2 pi u.y
        auto wi_local = mk3(sin_t * cs, sin_t * sn, cos_t);
        auto frame = frame_from_normal(mk3(env.direction[0], env.direction[1], env.direction[2]));
        auto e = env_directional(tb, env, wi_local);
        r.L = e.L, r.pdf = e.pdf;
        r.wi = normalize(mul3(env.env_to_world, to_world(frame, wi_local)));
        return r;
    }
    if (env.kind == kEnvConstant) {// uniform sphere, spherical.cpp:114-118
        auto z = 1.0f - 2.0f * u.x;
        auto rr = sqrtf(fmaxf(1.0f - z * z, 0.0f));
        float sn, cs;
        sincos_2pi(u.y, sn, cs);// phi = 2 pi u.y
        r.L = env_radiance(tb, env, f2{0.f, 0.f}) * env.scale;
        r.pdf = kInvPi * 0.25f;
        r.wi = normalize(mul3(env.env_to_world, mk3(rr * cs, rr * sn, z)));
        return r;
    }
    auto W = env.map_width, H = env.map_height;
    float ry, rx;
    auto sy = alias_slot(u.y, H, ry);
    auto ey = env.alias[sy];
    auto py = alias_pick(ey.prob, ey.alias, sy, ry);
    auto row = env.alias + H + py.index * W;
    auto sx = alias_slot(u.x, W, rx);
    auto ex = row[sx];
    auto px = alias_pick(ex.prob, ex.alias, sx, rx);
    f2 uv{(static_cast<float>(px.index) + px.u) / static_cast<float>(W), (static_cast<float>(py.index) + py.u) / static_cast<float>(H)};
    auto p = env.pdf[py.index * W + px.index];
    auto theta = kPi * uv.y;// Spherical::uv_to_direction: phi = 2 pi (1 - uv.x), theta = pi uv.y
    float sin_phi, cos_phi, sin_theta, cos_theta;
    sincos_2pi(1.f - uv.x, sin_phi, cos_phi);
    sincos_2pi(0.5f * uv.y, sin_theta, cos_theta);
    auto w = normalize(mk3(sin_phi * sin_theta, cos_theta, cos_phi * sin_theta));
    r.L = env_radiance(tb, env, uv) * env.scale;
    r.pdf = env_directional_pdf(p, theta);
    r.wi = normalize(mul3(env.env_to_world, w));
    return r;
}

// Combined nodes nested in each other (combined.cpp composes freely; device code has no recursion): the tree is walked with an
// explicit stack, in the reference's operation order (post-order: a.L * scale_a + b.L * scale_b, the pdfs interpolated), by out-of-line
// functions that only the variants which make real calls anyway hold (lrhip.hip picks one for such a scene) -- the lean kernels' register
// allocation never sees them.  At most LR_ENV_MAX_COMBINED_DEPTH Combined nodes on a path (lr_scene.h; checked at upload).
#if defined(LR_VARIANT) && ((LR_VARIANT) & (96 | 256))
#define LR_ENV_TREE 1
__device__ __noinline__ EnvEval env_evaluate_tree(EnvTables tb, const DEnvironment *root, f3 wi) {// CombinedInstance::evaluate, combined.cpp:57-78
    const DEnvironment *node[LR_ENV_MAX_COMBINED_DEPTH];
    f3 local[LR_ENV_MAX_COMBINED_DEPTH];
    EnvEval first[LR_ENV_MAX_COMBINED_DEPTH];
    bool second[LR_ENV_MAX_COMBINED_DEPTH];
    auto sp = 0;
    auto cur = root;
    for (;;) {
        while (cur->kind == kEnvCombined && sp < LR_ENV_MAX_COMBINED_DEPTH) {// down the first children
            node[sp] = cur, second[sp] = false;
            wi = normalize(mul3(cur->world_to_env, wi));
            local[sp] = wi;
            cur = cur->child[0], sp++;
        }
        auto r = env_evaluate_one(tb, cur, wi);
        for (;;) {// up: a finished first child starts the second one, a finished second child finishes the node
            if (sp == 0) { return r; }
            const auto top = sp - 1;
            if (!second[top]) {
                first[top] = r, second[top] = true;
                cur = node[top]->child[1], wi = local[top];
                break;
            }
            const auto sa = node[top]->child_scale[0], sb = node[top]->child_scale[1];
            r = EnvEval{first[top].L * sa + r.L * sb, lerp(first[top].pdf, r.pdf, sb / (sa + sb))};
            sp = top;
        }
    }
}
__device__ __noinline__ EnvSample env_sample_tree(EnvTables tb, const DEnvironment *root, f2 u) {// CombinedInstance::sample, combined.cpp:80-111
    const DEnvironment *node[LR_ENV_MAX_COMBINED_DEPTH];
    bool took_a[LR_ENV_MAX_COMBINED_DEPTH];
    auto sp = 0;
    auto cur = root;
    while (cur->kind == kEnvCombined && sp < LR_ENV_MAX_COMBINED_DEPTH) {// choose a child by u.x, all the way down
        const auto sa = cur->child_scale[0], sb = cur->child_scale[1];
        const auto weight_a = sa / (sa + sb);
        const auto a = u.x < weight_a;
        u.x = a ? u.x / weight_a : (u.x - weight_a) / (1.f - weight_a);
        node[sp] = cur, took_a[sp] = a;
        cur = cur->child[a ? 0 : 1], sp++;
    }
    auto s = env_sample_one(tb, cur, u);
    while (sp > 0) {// up: the other child evaluated in the sampled direction, the node's transform applied
        sp--;
        const auto n = node[sp];
        const auto sa = n->child_scale[0], sb = n->child_scale[1];
        const auto weight_a = sa / (sa + sb);
        const auto o = env_evaluate_tree(tb, n->child[took_a[sp] ? 1 : 0], s.wi);
        if (took_a[sp]) {
            s.L = s.L * sa + o.L * sb;
            s.pdf = lerp(s.pdf, o.pdf, 1.f - weight_a);
        } else {
            s.L = o.L * sa + s.L * sb;
            s.pdf = lerp(o.pdf, s.pdf, 1.f - weight_a);
        }
        s.wi = normalize(mul3(n->env_to_world, s.wi));
    }
    return s;
}
#endif

// root environment: one record, or CombinedInstance over two (combined.cpp:57-111)
LR_D void env_evaluate(const DScene &scene, f3 wi, f3 &L, float &pdf) {
    auto &env = *scene.env;
    EnvTables tb{scene.textures, scene.texels};
#ifdef LR_ENV_TREE
    if (env.kind == kEnvCombined && env.tree != 0u) {
        auto e = env_evaluate_tree(tb, scene.env, wi);
        L = e.L, pdf = e.pdf;
        return;
    }
#endif
    if (env.kind != kEnvCombined) {
        auto e = env_evaluate_one(tb, scene.env, wi);
        L = e.L, pdf = e.pdf;
        return;
    }
    auto wi_local = normalize(mul3(env.world_to_env, wi));
    auto a = env_evaluate_one(tb, env.child[0], wi_local);
    auto b = env_evaluate_one(tb, env.child[1], wi_local);
    auto sa = env.child_scale[0], sb = env.child_scale[1];
    L = a.L * sa + b.L * sb;
    pdf = lerp(a.pdf, b.pdf, sb / (sa + sb));
}
LR_D void env_sample(const DScene &scene, f2 u, f3 &wi, f3 &L, float &pdf) {
    auto &env = *scene.env;
    EnvTables tb{scene.textures, scene.texels};
#ifdef LR_ENV_TREE
    if (env.kind == kEnvCombined && env.tree != 0u) {
        auto s = env_sample_tree(tb, scene.env, u);
        wi = s.wi, L = s.L, pdf = s.pdf;
        return;
    }
#endif
    if (env.kind != kEnvCombined) {
        auto s = env_sample_one(tb, scene.env, u);
        wi = s.wi, L = s.L, pdf = s.pdf;
        return;
    }
    auto sa = env.child_scale[0], sb = env.child_scale[1];
    auto weight_a = sa / (sa + sb);
    auto first = u.x < weight_a;// sample a and evaluate b, or the other way round
    u.x = first ? u.x / weight_a : (u.x - weight_a) / (1.f - weight_a);
    auto s = env_sample_one(tb, env.child[first ? 0 : 1], u);
    auto o = env_evaluate_one(tb, env.child[first ? 1 : 0], s.wi);
    if (first) {
        L = s.L * sa + o.L * sb;
        pdf = lerp(s.pdf, o.pdf, 1.f - weight_a);
    } else {
        L = o.L * sa + s.L * sb;
        pdf = lerp(o.pdf, s.pdf, 1.f - weight_a);
    }
    wi = normalize(mul3(env.env_to_world, s.wi));
}

// ---------------------------------------------------------------- light sampling
// LightSampler::Instance::sample (light_sampler.cpp:57-63) with UniformLightSamplerInstance::select / _sample_area /
// _sample_environment (uniform.cpp:78-137): one light or the environment, its radiance towards `it`, the solid-angle
// pdf (x the selection probability) and the shadow ray (Interaction::spawn_ray / spawn_ray_to, interaction.cpp:21-30).
struct LightPick {
    Ray shadow;
    f3 L;
    float pdf;
};
// NULL_SHAPE: `it` is Interaction{p} of a medium point (interaction.h:77-78): its Shape::Handle is default-constructed, the
// intersection-offset factor is 0 and p_robust() returns p itself (pinned against the reference's own code, oracle/_ref).
template<bool ENV, bool NULL_SHAPE = false>
LR_D LightPick sample_one_light(const DScene &scene, const SurfacePoint &it, float u_light_selection, f2 u_light_surface) {
    LightPick out;
    out.L = mk3(0.f), out.pdf = 0.f;
    auto n = static_cast<float>(scene.light_count);
    auto is_env = false;
    auto tag = 0u;
    auto prob = 0.f;
    if (scene.env_prob == 1.f) {
        is_env = true, prob = 1.f;
    } else if (scene.env_prob == 0.f) {
        tag = static_cast<uint32_t>(clampf(u_light_selection * n, 0.f, n - 1.f)), prob = 1.f / n;
    } else {
        auto uu = (u_light_selection - scene.env_prob) / (1.f - scene.env_prob);
        tag = static_cast<uint32_t>(clampf(uu * n, 0.f, n - 1.f));
        is_env = u_light_selection < scene.env_prob;
        prob = is_env ? scene.env_prob : (1.f - scene.env_prob) / n;
    }
    if (is_env) {// _sample_environment, uniform.cpp:125-137
        f3 wi;
        if (ENV && scene.env_kind != kEnvConstant) {
            env_sample(scene, u_light_surface, wi, out.L, out.pdf);
            out.pdf *= prob;
        } else {// constant emission: uniform sphere, spherical.cpp:114-118,138
            auto z = 1.0f - 2.0f * u_light_surface.x;
            auto r = sqrtf(fmaxf(1.0f - z * z, 0.0f));
            float sn, cs;
            sincos_2pi(u_light_surface.y, sn, cs);// phi = 2 pi u
            auto w = mk3(r * cs, r * sn, z);
            auto e = scene.env_to_world;
            wi = normalize(mk3(e[0], e[1], e[2]) * w.x + mk3(e[3], e[4], e[5]) * w.y + mk3(e[6], e[7], e[8]) * w.z);
            out.L = mk3(scene.env_L[0], scene.env_L[1], scene.env_L[2]);
            out.pdf = (kInvPi * 0.25f) * prob;
        }
        out.shadow.o = NULL_SHAPE ? it.p : robust_origin(it, wi);
        out.shadow.d = wi;
        out.shadow.t_min = 0.f, out.shadow.t_max = kFloatMax;
    } else {// _sample_area, uniform.cpp:107-123
        auto handle = scene.light_instances[tag];
        auto lh = reinterpret_cast<const uint4 *>(scene.instances + handle.instance_id)[0];
        auto l_tri_offset = scene.instances[handle.instance_id].triangle_offset;
        float u_remapped;
        auto slot = alias_slot(u_light_surface.x, lh.z, u_remapped);
        auto entry = scene.tri_alias[l_tri_offset + slot];
        auto pick = alias_pick(entry.prob, entry.alias, slot, u_remapped);
        f2 ut{pick.u, u_light_surface.y};// sample_uniform_triangle, sampling.cpp:89-98
        f2 uvt = ut.x < ut.y ? f2{0.5f * ut.x, -0.5f * ut.x + ut.y} : f2{-0.5f * ut.y + ut.x, 0.5f * ut.y};
        SurfacePoint lp;
        reconstruct<false>(scene, handle.instance_id, pick.index, mk3(uvt.x, uvt.y, 1.0f - uvt.x - uvt.y), lp);
        lp.back_facing = dot(lp.ng, it.p - lp.p) < 0.f;
        light_evaluate(scene, lp, pick.index, it.p, out.L, out.pdf);
        out.pdf *= prob;
        auto p_from = NULL_SHAPE ? it.p : robust_origin(it, lp.p - it.p);// spawn_ray_to, interaction.cpp:25-30
        auto Lv = lp.p - p_from;
        auto dist = length(Lv);
        out.shadow.o = p_from;
        out.shadow.d = Lv * (1.f / dist);
        out.shadow.t_min = 0.f, out.shadow.t_max = dist * .9999f;
    }
    return out;
}

}// namespace lrd
