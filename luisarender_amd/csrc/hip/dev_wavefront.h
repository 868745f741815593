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

// The same for a path that finishes OUTSIDE the wave that owns its tile (a path that was parked at a heavy hit: it ends in the
// continuation pass or in the heavy kernel, in whatever wave picked its record up).  The order of these adds is a race between
// waves, so they are made order-independent: the radiance goes into 64-bit FIXED-POINT sums (integer atomics: associative, the
// film stays bit-reproducible run to run and under any tile sharding) and the sample count into the film's own w (an integer below
// 2^24: exact in fp32 in any order).  lrhip_render adds accum / scale to the film once, after the last round.
LR_D void wf_film_accumulate(const DScene &scene, float4 *film, uint32_t pixel_index, f3 rgb, float clamp) {
    if (!(any_nan(rgb) || any_inf(rgb))) {
        auto threshold = clamp * fmaxf(1.f, 1.f);
        auto strength = fmaxf(fmaxf(fmaxf(fabsf(rgb.x), fabsf(rgb.y)), fabsf(rgb.z)), 0.f);
        auto c = rgb * (threshold / fmaxf(strength, threshold));
        auto acc = scene.wf.accum + static_cast<size_t>(pixel_index) * 3u;
        auto fixed = [&](float v) { return static_cast<unsigned long long>(static_cast<double>(fmaxf(v, 0.f)) * static_cast<double>(scene.wf.accum_scale) + 0.5); };
        if (c.x != 0.f) { atomicAdd(acc + 0, fixed(c.x)); }
        if (c.y != 0.f) { atomicAdd(acc + 1, fixed(c.y)); }
        if (c.z != 0.f) { atomicAdd(acc + 2, fixed(c.z)); }
        atomicAdd(&film[pixel_index].w, 1.f);
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
    LR_D void put(uint32_t slot, uint32_t field, uint32_t v) const { base[static_cast<size_t>(field) * capacity + slot] = v; }
    LR_D void put(uint32_t slot, uint32_t field, float v) const { put(slot, field, __float_as_uint(v)); }
    LR_D void put3(uint32_t slot, uint32_t field, f3 v) const { put(slot, field, v.x), put(slot, field + 1u, v.y), put(slot, field + 2u, v.z); }
    LR_D uint32_t get(uint32_t slot, uint32_t field) const { return base[static_cast<size_t>(field) * capacity + slot]; }
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

}// namespace lrd
