// heavy_variant.hip — ONE instantiation of the heavy-closure kernel of wavefront mode (heavy_kernel.h) and its launch / occupancy
// entry points for lrhip.hip.  LR_HVARIANT: bit 0 diagnostics counters, bit 1 generic sampler, bits 2-3 the closure kind (0 Disney,
// 4 Mix, 8 Layered), bit 9 (512) Mix / Layered nested in each other (Mix / Layered kernels only).  One object per mask, built in
// parallel like the megakernel variants.
#include <hip/hip_runtime.h>

#ifndef LR_HVARIANT
#error "compile with -DLR_HVARIANT=<mask>"
#endif
#if (LR_HVARIANT) & 512
#define LR_NEST 1
#else
#define LR_NEST 0
#endif
// texture lookup and environment evaluate / sample as real calls (one copy each), like the megakernel variants that held these closures
#define LR_CALL __device__ __noinline__
#include "heavy_kernel.h"
#define LR_CAT2(a, b) a##b
#define LR_CAT(a, b) LR_CAT2(a, b)

#define LR_HFEATURES ((LR_HVARIANT) & ~12u)
#define LR_HKIND (((LR_HVARIANT) >> 2) & 3u)
namespace lrd {
template __global__ void heavy_kernel<LR_HFEATURES, LR_HKIND>(DScenePtr, RenderArgs);
}

extern "C" hipError_t LR_CAT(lrhip_heavy_launch_, LR_HVARIANT)(unsigned blocks, hipStream_t stream, const lrd::DScene *device_scene,
                                                              const lrd::RenderArgs *args) {
    hipLaunchKernelGGL((lrd::heavy_kernel<LR_HFEATURES, LR_HKIND>), dim3(blocks), dim3(lrd::kBlockThreads), 0, stream, (lrd::DScenePtr)device_scene, *args);
    return hipGetLastError();
}

extern "C" hipError_t LR_CAT(lrhip_heavy_occupancy_, LR_HVARIANT)(int *blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lrd::heavy_kernel<LR_HFEATURES, LR_HKIND>, lrd::kBlockThreads, 0);
}
