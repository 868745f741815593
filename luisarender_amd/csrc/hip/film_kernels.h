// film_kernels.h — the small kernels around the megakernel: chunk resolve and the Color film's convert
// (src/films/color.cpp:87-93).  Included by lrhip.hip only (the megakernel variants are separate objects).
#pragma once
#include "dev_scene.h"

namespace lrd {

// film += sum over chunks of the per-chunk partial sums, in chunk order (deterministic)
__global__ void resolve_partial_kernel(float4 *film, const float4 *partial, uint32_t pixel_count, uint32_t chunk_count,
                                       uint32_t width, uint32_t tiles_x, uint32_t tile_begin, uint32_t tile_end,
                                       uint32_t tile_stride) {
    auto i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixel_count) { return; }
    auto px = i % width, py = i / width;
    auto ty = py / 8u, tcol = px / 8u;
    auto tile = ty * tiles_x + (tcol + tiles_x - ty % tiles_x) % tiles_x;// the tile NUMBER of column tcol in the rotated row (lrhip.h)
    if (tile < tile_begin || tile >= tile_end || (tile - tile_begin) % tile_stride != 0u) { return; }
    auto v = film[i];
    for (auto c = 0u; c < chunk_count; c++) {
        auto p = partial[static_cast<size_t>(c) * pixel_count + i];
        v.x += p.x, v.y += p.y, v.z += p.z, v.w += p.w;
    }
    film[i] = v;
}

// wavefront mode: the fixed-point radiance sums of the paths that finished outside their tile's wave (dev_wavefront.h:
// wf_film_accumulate) join the film once per lrhip_render, and the sums are cleared for the next call
__global__ void wf_resolve_kernel(float4 *film, unsigned long long *accum, uint32_t pixel_count, double inv_scale) {
    auto i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixel_count) { return; }
    auto a = accum + static_cast<size_t>(i) * 3u;
    if ((a[0] | a[1] | a[2]) == 0ull) { return; }
    auto v = film[i];
    // (signed sums: dev_wavefront.h radiance_to_fixed)
    v.x += static_cast<float>(static_cast<double>(static_cast<long long>(a[0])) * inv_scale);
    v.y += static_cast<float>(static_cast<double>(static_cast<long long>(a[1])) * inv_scale);
    v.z += static_cast<float>(static_cast<double>(static_cast<long long>(a[2])) * inv_scale);
    film[i] = v;
    a[0] = 0ull, a[1] = 0ull, a[2] = 0ull;
}

// convert kernel of the Color film, color.cpp:87-93
__global__ void film_convert_kernel(const float4 *film, float4 *out, uint32_t pixel_count, float sx, float sy, float sz) {
    auto i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pixel_count) { return; }
    auto c = film[i];
    auto inv = 1.f / fmaxf(c.w, 1.f);
    out[i] = make_float4((inv * sx) * c.x, (inv * sy) * c.y, (inv * sz) * c.z, 1.f);
}

}// namespace lrd
