// megapath_variant.hip — ONE instantiation of the megakernel (feature mask LR_VARIANT, see megapath_kernel.h) and its
// launch / occupancy entry points for lrhip.hip.  Compiled once per mask of variants.h into its own object so that
// the variants build in parallel.
#include <hip/hip_runtime.h>

#ifndef LR_VARIANT
#error "compile with -DLR_VARIANT=<feature mask>"
#endif
#if ((LR_VARIANT) & 16384) && !defined(LR_ONLY_SAMPLER)// kFeatPadded: the generic sampler is PaddedSobol, at compile time (dev_shade.h: LR_SAMPLER_KIND_OF)
#define LR_ONLY_SAMPLER LR_SAMPLER_PADDED_SOBOL
#endif
#if (LR_VARIANT) & 256// kFeatVpt: the volumetric megakernel
#include "megavpt_kernel.h"
#define LR_KERNEL megavpt_kernel
#elif (LR_VARIANT) & 4096// kFeatPool: the path-pool scheduler (round 4)
#if defined(LR_POOL_PARK_ON_STACK) && LR_POOL_PARK_ON_STACK == 0 && !defined(LR_STACK_LDS)
#define LR_STACK_LDS 11// (such pool kernels give five of the sixteen LDS stack entries per lane to what a lane keeps across the shading block, megapool_kernel.h)
#endif
#include "megapath_kernel.h"
#include "megapool_kernel.h"
#define LR_KERNEL megapool_kernel
#else
#include "megapath_kernel.h"
#define LR_KERNEL megapath_kernel
#endif
#define LR_CAT2(a, b) a##b
#define LR_CAT(a, b) LR_CAT2(a, b)

namespace lrd {
template __global__ void LR_KERNEL<LR_VARIANT>(DScenePtr, RenderArgs);
}

// `device_scene`: the lrd::DScene record in device memory (lrhip_render copies it there ahead of every launch)
extern "C" hipError_t LR_CAT(lrhip_variant_launch_, LR_VARIANT)(unsigned blocks, hipStream_t stream, const lrd::DScene *device_scene,
                                                               const lrd::RenderArgs *args) {
    hipLaunchKernelGGL(lrd::LR_KERNEL<LR_VARIANT>, dim3(blocks), dim3(lrd::kBlockThreads), 0, stream,
                       (lrd::DScenePtr)device_scene, *args);
    return hipGetLastError();
}

extern "C" hipError_t LR_CAT(lrhip_variant_occupancy_, LR_VARIANT)(int *blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lrd::LR_KERNEL<LR_VARIANT>, lrd::kBlockThreads, 0);
}
