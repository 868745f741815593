/* lrhip.h — thin C ABI of the MI355X (gfx950) megakernel path tracer (liblrhip.so).
 *
 * This is the drop-in boundary of the hot path.  The reference has no such C interface: its
 * MegaPath integrator is a C++ plugin (`create`/`destroy`, src/base/scene_node.h:58-67) whose
 * Instance::render JIT-compiles and launches a LuisaCompute kernel.  Each entry point below
 * replaces one step of that path; the reference-side binding a maintainer would add lives in
 * INTEGRATION.md and luisarender_amd/csrc/host/plugin_megapath.cpp.
 *
 *   lrhip_create / lrhip_destroy   Context::create_device + Stream       src/apps/cli.cpp:166-172,181
 *   lrhip_upload_scene             Pipeline::create uploads               src/base/pipeline.cpp:44-99,
 *                                  Geometry::build                        src/base/geometry.cpp:12-27
 *   lrhip_update_scene             Pipeline::update / Geometry::update    src/base/pipeline.cpp:101-113, geometry.cpp:194-216
 *   lrhip_film_clear               ColorFilmInstance::prepare/clear       src/films/color.cpp:132-144
 *   lrhip_render                   _render_one_camera's spp loop of       src/base/integrator.cpp:86-107
 *                                  render(sample_id, time, weight).dispatch(resolution), i.e.
 *                                  Li() + film accumulate                 src/integrators/mega_path.cpp:49-156
 *   lrhip_film_download            ColorFilmInstance::download            src/films/color.cpp:99-105
 *   lrhip_film_reduce              (no reference equivalent: the one collective of the multi-GPU path)
 *   lrhip_get_counters             (no reference equivalent; roofline accounting)
 *
 * Conventions: 0 = OK, negative = error (text via lrhip_last_error, thread-local); nothing
 * throws or aborts across the boundary.  One context per GPU; a context is not thread-safe;
 * different contexts may be driven from different host threads / processes.  All buffers are
 * POD, little-endian, laid out as in lr_scene.h.  The library copies what it needs during
 * lrhip_upload_scene; the caller keeps ownership of the scene tables.
 */
#ifndef LRHIP_H
#define LRHIP_H

#include "lr_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrhip_ctx lrhip_ctx;

#define LRHIP_OK 0
#define LRHIP_ERROR_INVALID (-1)  /* bad argument / state        */
#define LRHIP_ERROR_DEVICE (-2)   /* HIP runtime failure         */
#define LRHIP_ERROR_UNSUPPORTED (-3)

/* Work of one lrhip_render call: samples [spp_begin, spp_end) of every pixel of the screen tiles
 * {tile_begin + k * tile_stride : tile_begin + k * tile_stride < tile_end}.  Tiles are 8x8 pixels; tile number
 * t = ty * tiles_x + j names the tile in row ty whose column is (j + ty) mod tiles_x (tiles_x = ceil(W/8)): every row is
 * rotated by its index, so that a strided shard (rank, tile_count, world) is a set of DIAGONALS of the frame, never a set
 * of columns (tiles_x is a multiple of the world size for every power-of-two frame).  Every tile has exactly one number,
 * so any partition of [0, tile_count) partitions the frame.  (0, tile_count, 1) renders the whole frame; (rank, tile_count,
 * world) is the interleaved screen-tile shard of one GPU; tile_begin >= tile_end is an empty shard (no error). */
typedef struct lrhip_render_params {
    uint32_t spp_begin, spp_end;
    uint32_t tile_begin, tile_end, tile_stride;
    uint32_t flags;          /* LRHIP_RENDER_* */
    /* Work-item sizing hint: the number of shards the frame is split into (0 = 1).  A frame is cut into
     * (tile, sample-chunk) items whose size balances the drain at the end of every item against the tail of the
     * launch; a 1/W shard has W times fewer tiles per GPU and wants smaller items (item size ~ sqrt(tiles per shard)).  The chunking —
     * and with it the fp32 summation order of the film — is a function of (resolution, spp, balance_shards) ONLY:
     * renders that pass the same value are bit-identical under any tile sharding and on any device.            */
    uint32_t balance_shards;
    /* Camera::ShutterSample weight of these samples (src/base/integrator.cpp:74,91-95: film()->accumulate(pixel,
     * shutter_weight * L)); read only when flags has LRHIP_RENDER_SHUTTER_WEIGHT, otherwise 1 */
    float shutter_weight;
} lrhip_render_params;

#define LRHIP_RENDER_COUNTERS 1u /* gather per-ray node/triangle counters (slower kernel variant) */
#define LRHIP_RENDER_SHUTTER_WEIGHT 2u /* lrhip_render_params.shutter_weight is valid */

typedef struct lrhip_counters {
    uint64_t paths, closest_rays, shadow_rays;
    uint64_t nodes_visited;  /* quantised BVH4 packets fetched (64 B each)   */
    uint64_t tris_tested;    /* triangle tests (48 B each)                   */
    uint64_t surface_hits, nee_samples, path_length_sum;
    /* SIMD-occupancy diagnostics: lane-iterations of the traversal loop (all lanes of every wave) and the
     * ones in which the lane had a ray in flight; shading blocks executed per lane / with a hit to shade */
    uint64_t trace_steps, trace_steps_busy, shade_calls, shade_busy;
    uint64_t trace_steps_starved; /* lane-steps idle because the lane had no sample left to start */
    /* wave-cycle diagnostics (s_memtime, summed over waves): inside the shading block (A), inside the traversal
     * loop (B), and from a wave's first to its last instruction */
    uint64_t shade_cycles, trace_cycles, wave_cycles;
    uint64_t nodes_empty;    /* node visits in which no child was hit (popped after the ray had already shortened, or plain misses) */
    /* wave cycles of three sections of the shading block: hit reconstruction + emission + light sample, closure evaluate + sample +
     * Russian roulette, path regeneration (camera rays); the remainder of shade_cycles is queue bookkeeping and ray launch */
    uint64_t shade_light_cycles, shade_closure_cycles, shade_regen_cycles;
    /* round 6: section cycles of the traversal loop and the shading block from the stall-probe build of the counting kernels
     * (make hip-variant DEFS=-DLR_STALL_PROBE, tools/stall_probe.py; slot names there); zero in the shipped library */
    uint64_t probe[16];
} lrhip_counters;

int lrhip_create(int device_ordinal, lrhip_ctx **out);
void lrhip_destroy(lrhip_ctx *ctx);

/* optional: launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int lrhip_set_stream(lrhip_ctx *ctx, void *hip_stream);

/* scene->accel must be built (lrhost_scene_build_accel) */
int lrhip_upload_scene(lrhip_ctx *ctx, const lr_scene *scene);

/* Pipeline::update (src/base/pipeline.cpp:101-113) for the next shutter sample of a motion-blurred frame: `scene` must be the
 * uploaded scene moved to another time (lrhost_scene_set_time).  Only what moves is copied again, over the same device buffers
 * and in stream order — instance matrices, the re-baked triangles and their shading records, the refitted BVH packets, camera and
 * environment transforms — while textures, environment tables, materials, the film, its binding and the counters stay.  Table
 * sizes and BVH topology are checked against the upload; anything else that differs is the caller's error.                      */
int lrhip_update_scene(lrhip_ctx *ctx, const lr_scene *scene);

/* optional: accumulate into a caller-owned device buffer float4[W*H] (e.g. a torch tensor that
 * RCCL reduces afterwards); NULL = library-owned film (default) */
int lrhip_bind_film(lrhip_ctx *ctx, void *device_float4_film);
int lrhip_film_clear(lrhip_ctx *ctx);

/* asynchronous on the context's stream */
int lrhip_render(lrhip_ctx *ctx, const lrhip_render_params *params);
int lrhip_synchronize(lrhip_ctx *ctx);

/* converted != 0: (sum.rgb / max(sum.w, 1)) * 2^exposure, alpha 1 (color.cpp:87-93);
 * converted == 0: the raw (sum r, sum g, sum b, n) film.  Synchronises.              */
int lrhip_film_download(lrhip_ctx *ctx, float *rgba, int converted);

/* The path's only collective: sum-reduce of the per-rank films to rank `root` over RCCL / xGMI, in place on the film
 * this context accumulates into, in stream order behind the renders.  `nccl_comm` is the caller's ncclComm_t (one per process /
 * GPU, created by the caller: ncclCommInitRank); librccl.so is loaded on first use, so the library has no link-time dependency
 * on it.  Every pixel is owned by exactly one rank under tile sharding and the others hold exact zeros there, so the reduced
 * film is bit-identical to the 1-GPU film.  The reference has no multi-device path (src/apps/cli.cpp:172,181).              */
int lrhip_film_reduce(lrhip_ctx *ctx, void *nccl_comm, int root);
/* the same for several contexts driven by ONE host thread (one process, one context + communicator per GPU): a single RCCL group */
int lrhip_film_reduce_group(int count, lrhip_ctx *const *ctxs, void *const *comms, int root);

/* Communicators for lrhip_film_reduce, so that a host needs no RCCL headers of its own (librccl.so is loaded on first use):
 *   one process per GPU:  rank 0 calls lrhip_comm_unique_id, ships the 128 bytes to the other ranks by its own means (MPI, a
 *                         file, torch.distributed), every rank calls lrhip_comm_init_rank on its context         (ncclCommInitRank)
 *   one process, N GPUs:  lrhip_comm_init_all(N, device ordinals, comms)                                        (ncclCommInitAll)
 * lrhip_comm_destroy releases one communicator.  lrhip_device_count = hipGetDeviceCount.                                      */
#define LRHIP_COMM_ID_BYTES 128
int lrhip_device_count(int *count);
int lrhip_comm_unique_id(unsigned char id[LRHIP_COMM_ID_BYTES]);
int lrhip_comm_init_rank(lrhip_ctx *ctx, int world, int rank, const unsigned char id[LRHIP_COMM_ID_BYTES], void **comm);
int lrhip_comm_init_all(int count, const int *devices, void **comms);
int lrhip_comm_destroy(void *comm);
/* what a communicator actually spans: out = { ranks (ncclCommCount), this rank (ncclCommUserRank), its HIP device (ncclCommCuDevice) } --
 * so that a host (bench.py's multi_gpu block) can say from its own output whether RCCL saw the N ranks it was launched with */
int lrhip_comm_info(void *comm, int out[3]);

int lrhip_get_counters(lrhip_ctx *ctx, lrhip_counters *out); /* summed since upload; synchronises */
/* HIP-event time of the megakernel launches of the last lrhip_render call, in ms; synchronises */
double lrhip_last_render_ms(lrhip_ctx *ctx);
/* Feature mask of the precompiled megakernel variant the last lrhip_render launched (the reference JIT-compiles one
 * kernel per scene, src/base/integrator.cpp:56-77; here the smallest precompiled superset of the scene's needs is
 * picked): bit 0 counters, 1 generic sampler, 2 image/directional environment, 3 alpha-tested traversal, 4 Disney,
 * 5 Mix, 6 Layered.  The kernel's symbol is lrd::megapath_kernel<mask>.                                          */
#define LRHIP_FEAT_COUNT 1u
#define LRHIP_FEAT_GENERIC_SAMPLER 2u
#define LRHIP_FEAT_ENVIRONMENT 4u
#define LRHIP_FEAT_ALPHA 8u
#define LRHIP_FEAT_DISNEY 16u
#define LRHIP_FEAT_BYTE_TEXELS 8192u /* a lean kernel that decodes 8-bit texels (lrhip_set_texture_storage) */
#define LRHIP_FEAT_PADDED_SOBOL 16384u /* (with LRHIP_FEAT_GENERIC_SAMPLER | LRHIP_FEAT_POOL) a pool kernel compiled for the PaddedSobol sampler */
#define LRHIP_FEAT_MIX 32u
#define LRHIP_FEAT_LAYERED 64u
#define LRHIP_FEAT_AUX_INTEGRATORS 128u
#define LRHIP_FEAT_VOLUMETRIC 256u
#define LRHIP_FEAT_NESTED 512u /* Mix trees with Layered leaves / Layered surfaces with Mix interfaces */
uint32_t lrhip_last_variant(lrhip_ctx *ctx);

/* Diagnostics of ONE context, for tests and tools (the product path never calls it; the library reads no environment variable):
 *   force_features  scene-feature bits (LRHIP_FEAT_ENVIRONMENT .. LRHIP_FEAT_LAYERED) OR-ed into what the uploaded scene needs, so
 *                   that a larger precompiled variant renders a scene that does not need it (A/B of variants, twin tests); 0 = none
 *   item_scale      factor on the loss model's constant behind the work-item size of lrhip_render (sweeps); 0 or 1 = default; a
 *                   negative value = its magnitude with UNIFORM work items (rounds 1-2) instead of the tapered ones of round 3.
 *                   A value other than the default changes the order of a pixel's float adds, i.e. the film's last bits.        */
int lrhip_set_diagnostics(lrhip_ctx *ctx, uint32_t force_features, double item_scale);

/* The work-item partition lrhip_render uses for `spp` samples per pixel of a frame cut into `balance_shards` shards (no device
 * needed): out = { chunks per tile, how many of them are big, samples per pixel of a big chunk, of a small chunk }.  Chunk k of a
 * tile covers samples [k * big, ..) for k < big count and [big count * big + (k - big count) * small, ..) after that, clipped to spp;
 * all big items of a launch are handed out before the first small one (items taper towards the end of the launch).              */
int lrhip_work_items(uint32_t width, uint32_t height, uint32_t spp, uint32_t balance_shards, uint32_t out[4]);

/* Wavefront mode (round 3): a scene with Mix or Layered surfaces under the MegaPath integrator is rendered by a lean megakernel that
 * parks the paths reaching a Disney / Mix / Layered surface in HBM queues, a heavy-closure kernel that shades those vertices in full
 * waves of one closure kind, and a continuation pass of the megakernel -- alternating until the queues are empty.  The sample range
 * is cut into slices of `slice_paths` paths of one nominal shard of the frame (tile_count / balance_shards tiles): a function of the
 * frame and the caller's hint only, like the work items, so that films stay bit-identical under sharding and whatever memory is
 * free; a call over more tiles than the queues can hold takes them group after group, which changes no bit.
 *   mode         0 = automatic (default), 1 = never: the all-in-one megakernel variants (A/B, tests),
 *                2 = automatic with queues of eight tiles (tests: the tile groups a GPU short of memory would use)
 *                Round 6: a slice runs ONE round and HANDS the paths still parked OVER to the next slice's first round (they are
 *                independent of their slice and add to the frame's order-independent fixed-point sums; the call's last slice runs all its
 *                rounds) -- its remaining rounds moved a few thousand paths each at one batch's latency, 6 % of a kitchen-class frame.
 *                mode | rounds << 8 overrides the one (tests, A/B); mode | 65535 << 8: never, every slice runs all its rounds.
 *   slice_paths  paths per slice (0 = default 2^28: 86 .. 100 GB of queues when a whole slice is in flight, the hand-over margin included)
 * lrhip_last_variant reports LRHIP_FEAT_WAVEFRONT | the lean kernel's bits | the closure bits the heavy kernel served.          */
#define LRHIP_FEAT_WAVEFRONT 1024u
int lrhip_set_wavefront(lrhip_ctx *ctx, uint32_t mode, uint32_t slice_paths);

/* Scheduling of the lean megakernels (round 4).  The path-pool kernels (csrc/hip/megapool_kernel.h) give every lane TWO path contexts:
 * while one context's rays are traced the other waits -- for the shading block with its hit, or for the lane with the rays of its
 * next job -- so that a lane whose job ends goes on with its other context inside the traversal loop (rays and hits in registers,
 * the 64 bytes of path state in a per-thread record, the ray in flight parked in the LDS while the lane shades).  Work items overlap
 * inside a wave and the film is summed in 64-bit fixed point: bit-reproducible under ANY sharding, grid size and work-item
 * partition.  Measured on MI355X (profiles/r04_final_*): lane utilisation of the traversal loop 0.56 -> 0.91, of the shading block
 * 0.47 -> 0.63, films equal to the one-path-per-lane kernels' to 8e-8, and 5 .. 18 % faster on every scene but the smallest ones
 * (a Cornell box: 13 % slower -- a ray is a handful of steps there and the pool's costlier shading block is not paid back).  They
 * serve every scene the lean kernels serve (basic closures and Disney inline, the wavefront-mode passes).
 * The fixed-point film -- the pool kernels' and wavefront mode's alike -- holds film clamp x spp up to 2^37 per call: a call beyond that
 * (a clamp of 1e7 at 65536 spp) is rendered in sample sub-ranges that fit, one after the other (round 5; the film is the same film: a
 * sub-range is what a progressive caller would have passed).  Only a clamp that cannot hold ONE sample (clamping switched off: 1e20,
 * inf) takes the float-accumulating kernels of rounds 1-3 instead -- one path per lane, Mix / Layered scenes on the all-in-one
 * variants (about half the speed of wavefront mode on the kitchen class) -- as do paths deeper than 65535 and the variants with
 * out-of-line closures / sibling integrators / media.  lrhip_last_variant tells which family rendered (LRHIP_FEAT_POOL, LRHIP_FEAT_WAVEFRONT).
 *   mode  0 = automatic: the pool kernels where they are the faster ones -- from 98304 BVH triangles up, from twice that for paths of
 *             depth <= 6, from half of it for scenes of fewer than 64 spp (tools/sched_sweep.py: triangles x depth x spp; the choice is
 *             never more than 1.4 % off the better kernel there) -- one path per lane below;
 *         1 = one path per lane; 2 = the pool kernels wherever one exists for the scene
 * lrhip_last_variant reports LRHIP_FEAT_POOL when the pool kernels rendered.                                                         */
#define LRHIP_FEAT_POOL 4096u
int lrhip_set_scheduler(lrhip_ctx *ctx, uint32_t mode);
/* Texel storage of the NEXT lrhip_upload_scene (round 5).  The host hands every image over as float RGBA texels (lr_scene.texels: what the
 * reference's textures hold).  An image whose every texel is an 8-bit code's float -- decoded from a PNG / JPEG / BMP / TGA file -- can stay
 * 8 bits per channel on the device: a quarter of the footprint in HBM and in the caches, decoded per lookup to exactly the floats the host
 * made (both of the host readers' conversions, b * (1 / 255.f) and b / 255.f, bit for bit: tests/test_gpu_parity.py).
 *   mode  1 = automatic (default): where the scene's image texels exceed 192 MB as floats (below that the caches hold them and the
 *             decode's few instructions per texel are not paid back: kitchen class -1.5 %, camera class +4 %);
 *         0 = never; 2 = every image that qualifies (A/B, tests).
 * Round 6: the decode is compiled into the kernels that need it only (lean kernels of the LRHIP_FEAT_BYTE_TEXELS bit -- the Disney feature
 * sets of both schedulers -- and every variant that makes real calls); a MegaPath scene with alpha-tested, Mix or Layered surfaces or
 * nested Combined environments renders on kernels without it and keeps float texels under mode 1; mode 2 on such a scene is
 * LRHIP_ERROR_UNSUPPORTED at upload.  The float texels of a packed image are not uploaded as well.                                    */
int lrhip_set_texture_storage(lrhip_ctx *ctx, uint32_t mode);
uint64_t lrhip_packed_texels(lrhip_ctx *ctx); /* texels of the uploaded scene held as 8-bit codes (4 bytes each instead of 16) */

/* The automatic rule's threshold (no device needed): the number of BVH triangles from which a scene whose integrator allows paths of
 * `max_depth` vertices and whose description asks for `scene_spp` samples per pixel (0 = unknown) renders on the pool kernels.       */
uint32_t lrhip_pool_auto_triangles(uint32_t max_depth, uint32_t scene_spp);

const char *lrhip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LRHIP_H */
