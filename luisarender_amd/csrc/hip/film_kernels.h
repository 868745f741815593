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

// Wavefront mode, round 6: THE TAIL OF A SLICE IS HANDED OVER.  A slice's rounds { heavy kernels -> continuation pass } thin out fast -- on the kitchen
// class the third round moves a twelfth of the first one's paths -- and from then on a round takes what ONE batch's latency takes, whatever it moves:
// 13 of a slice's 16 rounds were 24 ms of its ~400 (profiles/r06_final_wf_trace_c5_2048spp.txt, the per-round table).  Paths are independent
// of their slice (a parked record holds pixel, depth and stream position; finished paths add to the frame's fixed-point sums, which are
// order-independent), so after `rounds` rounds the paths still parked simply wait in their queues for the NEXT slice's first round and ride along with
// its thousand times more numerous ones.  No host round trip: mode 0 (after a round from the hand-over round on) moves the three queues' counts
// aside if they are at most `margin` (the slots the queues hold beyond a slice's own paths) -- the rounds still launched find empty queues and
// cost microseconds; mode 1 (ahead of the next slice's camera pass) puts them back, and the camera pass parks behind them.  The frame's LAST
// slice runs all its rounds.  Films bit-identical with and without (tests/test_gpu_pool.py).
__global__ void wf_carry_kernel(uint32_t *counts, uint32_t margin, uint32_t mode) {
    if (threadIdx.x != 0u || blockIdx.x != 0u) { return; }
    const auto carried = counts[kWfCarry] + counts[kWfCarry + 1u] + counts[kWfCarry + 2u];
    if (mode == 0u) {
        const auto parked = static_cast<unsigned long long>(counts[kWfCountHeavy]) + counts[kWfCountHeavy + 1u] + counts[kWfCountHeavy + 2u];
        if (carried == 0u && parked != 0ull && parked <= margin) {
            for (auto k = 0u; k < kWfKinds; k++) { counts[kWfCarry + k] = counts[kWfCountHeavy + k], counts[kWfCountHeavy + k] = 0u; }
        }
    } else if (carried != 0u) {
        for (auto k = 0u; k < kWfKinds; k++) { counts[kWfCountHeavy + k] += counts[kWfCarry + k], counts[kWfCarry + k] = 0u; }
    }
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
