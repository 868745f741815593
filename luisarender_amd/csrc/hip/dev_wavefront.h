// dev_wavefront.h — what the lean megakernel's wavefront variants (megapath_kernel.h: kFeatWf / kFeatCont) and the heavy-closure
// kernel (heavy_kernel.h) share: the film adds, the MIS weight and the layout of the two kinds of path records in HBM
// (dev_scene.h: WfArgs).
#pragma once
#include "dev_heavy.h"

namespace lrd {

// feature bits of the kernel variants (megapath_kernel.h describes them)
enum : uint32_t {
    kFeatCount = 1u, kFeatGeneric = 2u, kFeatEnv = 4u, kFeatAlpha = 8u, kFeatDisney = 16u, kFeatMix = 32u, kFeatLayered = 64u,
    kFeatAux = 128u,
    kFeatVpt = 256u,// the volumetric megakernel (megavpt_kernel.h): a different kernel, same launch interface
    kFeatNest = 512u,
    // wavefront mode (dev_scene.h: WfArgs): a LEAN megakernel that parks the paths hitting a Disney / Mix / Layered surface in HBM
    // queues for the heavy-closure kernel (heavy_kernel.h) instead of shading them; kFeatCont = its continuation pass, whose work
    // items are the records the heavy kernel wrote (next ray + shadow ray) instead of camera samples
    kFeatWf = 1024u, kFeatCont = 2048u,
    // round 4: the path-pool scheduler (megapool_kernel.h) instead of one path per lane; exists for the lean masks and the wavefront passes
    kFeatPool = 4096u,
    // round 6: the lean kernel decodes 8-bit texels (dev_shade.h: texel_at, LR_BYTE_TEXELS).  The decode's branch in the texture lookup cost
    // every lean kernel its allocation -- C2, whose scene holds no image, 2.8 %, C3 1.7 %, C5 1.3 % (profiles/r06k_byte_texel_bit.txt) -- so it
    // is compiled into the lean kernels of this bit only, which lrhip_render takes when the uploaded scene holds packed texels; the variants
    // that make real calls (Mix / Layered / auxiliary / volumetric) and the heavy kernels keep it in their one out-of-line lookup
    kFeatByteTex = 8192u,
    // round 6: the generic sampler's kind is PaddedSobol AT COMPILE TIME (the translation unit defines LR_ONLY_SAMPLER to match, megapath_variant.hip):
    // no three-way dispatch at the draws, and the pool kernel keeps nothing of the stream but (sample index, pixel) -- the dimension is a
    // function of the depth (megapool_kernel.h: PADDED).  PaddedSobol is what the reference's scene converter writes for every README scene
    // (tools/tungsten2luisa.py:373-412).  Compiled for the lean pool kernels (variants.h: LR_PADDED_LIST); lrhip_render takes one where it
    // exists and the scene's sampler is PaddedSobol, the run-time generic kernel of the same mask otherwise.  C2: 953 -> 975 Msamples/s at
    // 256 spp, films bit-identical (profiles/r06y_padded_sobol_kernels.txt)
    kFeatPadded = 16384u,
    kFeatSceneMask = kFeatEnv | kFeatAlpha | kFeatDisney | kFeatMix | kFeatLayered
};

// ColorFilmInstance::_accumulate (color.cpp:107-130, effective_spp = 1) into the wave's LDS copy of its tile.
LR_D void film_accumulate(float4 *pixel, f3 rgb, float clamp) {
    if (!(any_nan(rgb) || any_inf(rgb))) {
        auto threshold = clamp * fmaxf(1.f, 1.f);
        auto strength = fmaxf(fmaxf(fmaxf(fabsf(rgb.x), fabsf(rgb.y)), fabsf(rgb.z)), 0.f);
        auto c = rgb * (threshold / fmaxf(strength, threshold));
        if (c.x != 0.f || c.y != 0.f || c.z != 0.f) {
            atomicAdd(&pixel->x, c.x), atomicAdd(&pixel->y, c.y), atomicAdd(&pixel->z, c.z);
        }
        atomicAdd(&pixel->w, 1.f);
    }
}

// Radiance -> 64-bit FIXED POINT (two's complement in an unsigned word: sums wrap exactly, so negative samples -- the Mitchell and
// Lanczos filters have negative lobes, filter.cpp:49-64 weight = f / pdf -- add like any other; round 3 clamped them to zero).
// `scale` is a power of two (WfArgs::accum_scale): the product is exact, the conversion rounds to nearest.
LR_D unsigned long long radiance_to_fixed(float v, float scale) {
    // (the product is exact -- a power of two -- and rintf leaves an integer-valued float: the conversion truncates nothing.  In fp32
    // throughout: the fp64 form cost the pool kernel 24 spilled VGPRs in the middle of its shading block)
    return static_cast<unsigned long long>(static_cast<long long>(rintf(v * scale)));
}

// The same for a path that finishes OUTSIDE the wave that owns its tile (a path that was parked at a heavy hit: it ends in the
// continuation pass or in the heavy kernel, in whatever wave picked its record up; round 4: a straggler of a work item its wave has
// already left, megapool_kernel.h).  The order of these adds is a race between waves, so they are made order-independent: the
// radiance goes into 64-bit FIXED-POINT sums (integer atomics: associative, the film stays bit-reproducible run to run and under
// any tile sharding) and the sample count into the film's own w (an integer below 2^24: exact in fp32 in any order).
// lrhip_render adds accum / scale to the film once, after the last round.
// WfArgs::count_at_flush (the pool kernels): the wave that owned the sample's work item has already counted it when it left the
// item; a REJECTED sample (NaN / Inf, color.cpp:110-113) takes its count back instead.
LR_D void wf_film_accumulate(const DScene &scene, float4 *film, uint32_t pixel_index, f3 rgb, float clamp) {
    if (!(any_nan(rgb) || any_inf(rgb))) {
        auto threshold = clamp * fmaxf(1.f, 1.f);
        auto strength = fmaxf(fmaxf(fmaxf(fabsf(rgb.x), fabsf(rgb.y)), fabsf(rgb.z)), 0.f);
        auto c = rgb * (threshold / fmaxf(strength, threshold));
        auto acc = scene.wf.accum + static_cast<size_t>(pixel_index) * 3u;
        if (c.x != 0.f) { atomicAdd(acc + 0, radiance_to_fixed(c.x, scene.wf.accum_scale)); }
        if (c.y != 0.f) { atomicAdd(acc + 1, radiance_to_fixed(c.y, scene.wf.accum_scale)); }
        if (c.z != 0.f) { atomicAdd(acc + 2, radiance_to_fixed(c.z, scene.wf.accum_scale)); }
        if (scene.wf.count_at_flush == 0u) { atomicAdd(&film[pixel_index].w, 1.f); }
    } else if (scene.wf.count_at_flush != 0u) {
        atomicAdd(&film[pixel_index].w, -1.f);
    }
}

LR_D float balance(float f_pdf, float g_pdf) {// balance_heuristic, sampling.cpp:133-140
    auto sum = f_pdf + g_pdf;
    return sum == 0.0f ? 0.0f : f_pdf / sum;
}

// ---- path records (field-major: word f of record s at base[f * capacity + s])
// parked at a heavy hit:  0-2 ray direction  3 baked triangle  4-5 hit u v  6-8 beta  9-11 Li  12 pixel index  13 depth  14.. sampler
// continuation:           0-2 next ray o  3-5 next ray d  6-8 shadow o  9-11 shadow d  12 shadow t_max  13-15 nee  16-18 beta  19-21 Li
//                         22 pdf_bsdf  23 pixel index  24 depth | want_shadow << 16 | want_closest << 17  25.. sampler
struct WfQueue {
    uint32_t *base;
    uint32_t capacity;
    // field-major columns: a column's address is wave-uniform (SGPRs), the slot a 32-bit byte offset on top of it (capacity <= 2^30,
    // lrhip.hip) -- one global access with scalar base per field instead of 64-bit address arithmetic per field and lane
    LR_D uint32_t *at(uint32_t slot, uint32_t field) const {
        return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base + static_cast<size_t>(field) * capacity) + (slot << 2u));
    }
    LR_D void put(uint32_t slot, uint32_t field, uint32_t v) const { *at(slot, field) = v; }
    LR_D void put(uint32_t slot, uint32_t field, float v) const { put(slot, field, __float_as_uint(v)); }
    LR_D void put3(uint32_t slot, uint32_t field, f3 v) const { put(slot, field, v.x), put(slot, field + 1u, v.y), put(slot, field + 2u, v.z); }
    LR_D uint32_t get(uint32_t slot, uint32_t field) const { return *at(slot, field); }
    LR_D float getf(uint32_t slot, uint32_t field) const { return __uint_as_float(get(slot, field)); }
    LR_D f3 get3(uint32_t slot, uint32_t field) const { return mk3(getf(slot, field), getf(slot, field + 1u), getf(slot, field + 2u)); }
};
template<uint32_t SAMPLER_WORDS>
LR_D WfQueue wf_heavy_queue(const DScene &scene, uint32_t kind) {
    return WfQueue{scene.wf.heavy + static_cast<size_t>(kind) * (kWfHeavyWords + SAMPLER_WORDS) * scene.wf.capacity, scene.wf.capacity};
}
LR_D WfQueue wf_cont_queue(const DScene &scene) { return WfQueue{scene.wf.cont, scene.wf.capacity}; }

// slots for the lanes of `mask` behind one atomic of the wave's first such lane; wave-uniform control flow around it
LR_D uint32_t wf_reserve(uint32_t *counter, unsigned long long mask, uint32_t lane) {
    const auto leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(mask))) - 1u;
    uint32_t base = 0u;
    if (lane == leader) { base = atomicAdd(counter, static_cast<uint32_t>(__popcll(mask))); }
    base = static_cast<uint32_t>(__shfl(static_cast<int>(base), static_cast<int>(leader)));
    return base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32u), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Work distribution: ONE atomic counter over the item space (tiles x sample-chunks, tile-major).  The 4096 resident waves then work
// on a moving front of ~300 neighbouring tiles, so every XCD's L2 already holds the front's BVH lines; per-XCD item ranges were
// measured in round 2 and lost 1 % (eight fronts = eight tails; profiles/archive/r02b_ab_xcd_waves.txt).
LR_D uint32_t next_item(uint32_t *counter, uint32_t item_count, uint32_t lane) {
    uint32_t item = 0u;
    if (lane == 0u) { item = atomicAdd(counter, 1u); }
    item = __shfl(item, 0);
    return item < item_count ? item : kInvalid;
}


}// namespace lrd
