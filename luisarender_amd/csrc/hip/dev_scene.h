// dev_scene.h — HBM-resident scene layout of the gfx950 megakernel.
//
// Built by lrhip_upload_scene from the lr_scene tables (include/lr_scene.h).  Layout rules:
// every record a lane gathers per hit is 16-byte aligned and read with dwordx4 loads; records
// that are always read together share one 128-byte line (instance = handle + matrix + normal
// matrix; BVH node = 4 child boxes + refs).  See DESIGN.md "Data layout in HBM".
#pragma once
#include "../../../include/lr_scene.h"
#include "dev_math.h"

namespace lrd {

// One instance = one 128-byte line: handle (reference uint4, src/base/shape.cpp:46-70),
// object->world columns, transpose(inverse(M3x3)) columns (src/base/geometry.cpp:378), and the
// mesh slice offsets that stand in for the reference's bindless buffer ids.
struct alignas(16) DInstance {
    uint32_t handle[4];
    float c0[3]; uint32_t vertex_offset;
    float c1[3]; uint32_t triangle_offset;
    float c2[3]; uint32_t pad0;
    float t[3];  uint32_t pad1;
    float n0[3]; uint32_t pad2;
    float n1[3]; uint32_t pad3;
    float n2[3]; uint32_t pad4;
};
static_assert(sizeof(DInstance) == 128, "DInstance must be one cache line");

// Closure parameters with every constant texture folded in (what the reference's JIT does by
// inlining ConstantTexture values, src/textures/constant.cpp:73-79).  128 bytes = one line; the five simple
// closures only read the first 64.
//   MATTE   c0 = Kd                 s0 = sigma (degrees)
//   MIRROR  c0 = color              alpha
//   GLASS   c0 = Kr  c1 = Kt        s0 = eta_i  s1 = eta_t  s2 = Kr_ratio   alpha
//   PLASTIC c0 = Kd' c1 = sigma_a   s0 = Kd_weight  s1 = eta                alpha
//   METAL   c0 = n   c1 = k  c2 = Kd tint  s0 = eta_i                        alpha
//   DISNEY  c0 = color  s0 = color_lum   e[] = DisneyContext scalars (see kDisney* indices), x = lobes | flags
//   MIX     s0 = ratio  x[0], x[1] = surface tags of a, b
struct alignas(16) DClosure {
    uint32_t kind;
    uint32_t dynamic;   // 1: some parameter is a non-constant texture or a normal map -> resolve per hit
    float alpha_x, alpha_y;
    float c0[3]; float s0;
    float c1[3]; float s1;
    float c2[3]; float s2;
    float e[12];
    uint32_t x[4];
};
static_assert(sizeof(DClosure) == 128, "DClosure is one cache line");
enum : uint32_t {// DClosure::e slots of a Disney closure (DisneyContext, disney.cpp:304-321)
    kDisneyMetallic = 0, kDisneyEtaI, kDisneyEtaT, kDisneyRoughness, kDisneySpecularTint, kDisneyAnisotropic,
    kDisneySheen, kDisneySheenTint, kDisneyClearcoat, kDisneyClearcoatGloss, kDisneySpecularTrans, kDisneyFlatness
};// diffuse_trans lives in s1; x[0] = lobe mask, x[1] = transmissive, x[2] = thin

// One surface as the device reads it for a DYNAMIC closure (some parameter an image / checkerboard texture, or a normal map): the host's
// record and, beside it, what every texture SLOT of the record evaluates to where its texture is constant (round 6).  Per-hit closure
// resolution used to ask the texture table about every slot through one out-of-line lookup each -- a Disney surface with two image maps
// made thirteen dependent round trips of ~2.3 k cycles for its eleven constants, one surface kind after the other (the stall probe:
// 92 k of the camera class's 185 k cycles per shading batch, profiles/r06f_stalls1_c4.txt).  Now the constants are plain loads from this
// record -- independent of each other, issued together -- and only the slots in `dynamic_mask` are looked up (dev_heavy.h: load_lobe).
constexpr uint32_t kSurfaceSlots = 13u;// texture slots a surface kind uses (lr_scene.h: Disney's 0 .. 12)
struct alignas(16) DSurface {
    float value[kSurfaceSlots][4]; // Texture::evaluate of the slot's texture where it is constant (lr_texture::v); 16-byte aligned: read as float4
    uint32_t dynamic_mask;         // slots whose texture is not constant: evaluated per hit
    uint32_t channels[2];          // lr_texture::channels of the slot's texture, 4 bits per slot
    int32_t first_lookup;          // the texture a hit on this surface looks up FIRST: its normal map, else its first slot in dynamic_mask; -1: none (load_lobe, LR_LOBE_FORM 3)
    lr_surface raw;                // the host's record (136 B)
    uint32_t pad2[6];
};
static_assert(sizeof(lr_surface) == 136 && sizeof(DSurface) == 384, "DSurface: 13 slot values + masks + the host's record");

struct alignas(16) DLight {
    float L[3];          // emission * scale for constant emission
    int32_t emission_tex;// >= 0 and dynamic: evaluate per hit
    float scale;
    uint32_t two_sided;
    uint32_t dynamic;
    uint32_t pad;
};
static_assert(sizeof(DLight) == 32, "DLight");

struct DCamera {
    uint32_t kind, width, height, pad;
    float c2w[16];
    float tan_half_fov, focus_distance, lens_radius, projected_pixel_size;
    float ortho_scale, clip_near, clip_far, pad2;
};

// Quantised BVH4 packet, 64 bytes = 4 x dwordx4 (dev_trace.h).  Child planes are 8-bit offsets from the
// packet's box origin in units of `scale` per axis; byte i of each plane word belongs to child i.
struct alignas(16) DNodeQ {
    float origin[3]; float scale_x;
    uint32_t lo_x, lo_y, lo_z, hi_x;
    uint32_t hi_y, hi_z; float scale_y, scale_z;
    uint32_t child[4];
};
static_assert(sizeof(DNodeQ) == 64, "DNodeQ is half a cache line");

// Non-constant environments (image-based Spherical with importance tables, Directional), read through a pointer so
// that the kernel-argument block of the common case stays small.
struct DEnvironment {
    float world_to_env[9], env_to_world[9];// column-major 3x3
    int32_t emission_tex;
    float scale;
    uint32_t map_width, map_height;        // 2048 x 1024 (spherical.cpp:22)
    const lr_alias_entry *alias;           // [h] marginal + [h][w] conditional
    const float *pdf;                      // [h][w]
    float direction[3];
    float cos_half_angle;
    uint32_t visible, constant_emission;
    uint32_t kind;                         // kEnv* of this record
    float child_scale[2];                  // kEnvCombined (combined.cpp): scales and records of children a, b
    uint32_t tree;                         // kEnvCombined: a child is a Combined node itself (dev_shade.h: env_evaluate_tree)
    const DEnvironment *child[2];
};

enum : uint32_t { kEnvNone = 0u, kEnvConstant = 1u, kEnvImage = 2u, kEnvDirectional = 3u, kEnvCombined = 4u };

// Shading record of one BAKED (world-space) BVH triangle, indexed by the triangle index the traversal ends on: everything
// Geometry::shading_point (geometry.cpp:345-389) gathers through instance -> triangle indices -> three vertices — three
// DEPENDENT round trips of 16 + 64 + 12 + 96 B — in ONE 128-byte line.  600 k triangles = 77 MB; 288 GB of HBM make the
// replication free, the two saved round trips per hit are paid back in every shading block.
struct DShadeTri {// 8 x float4
    float p0[3];  uint32_t flags;        // world-space vertex 0 | shape property flags (handle.x & 1023)
    float e1[3];  uint32_t tags;         // p1 - p0 | handle.y (light / surface / medium tags)
    float e2[3];  uint32_t offset_bits;  // p2 - p0 | handle.w
    float n0[3];  float uv0x;            // vertex normals through transpose(inverse(M)) (not normalised: shading_point
    float n1[3];  float uv0y;            //   normalises the interpolated sum, which is linear in them)
    float n2[3];  float uv1x;
    float uv1y, uv2x, uv2y, tri_pdf;     // | pdf table entry of the primitive (lights)
    uint32_t inst, prim, tri_offset, pad;
};
static_assert(sizeof(DShadeTri) == 128, "one line per triangle");

// ---- wavefront mode (round 3): scenes with Mix / Layered surfaces.  The out-of-line closures do not live in the megakernel any
// more: a LEAN megakernel (kFeatWf variants) PARKS a path that hits a Disney / Mix / Layered surface -- its state goes into a queue
// in HBM, one queue per closure kind, and the lane takes the next sample -- a separate kernel (heavy_kernel.h) shades the parked
// vertices in full waves of ONE closure kind with a register allocation of its own and writes CONTINUATION records (shadow ray,
// next ray, throughput), which the lean megakernel's continuation pass (kFeatCont) traces and carries on like any other path.
// Rounds alternate until the queues are empty (at most max_depth of them).  All queues are field-major [field][slot] so that a
// wave's pushes and pops are coalesced.  The reference's own design for the same idea: src/integrators/wave_path_v2.cpp:419-440.
constexpr uint32_t kWfKinds = 3u;            // Disney, Mix, Layered (LR_SURFACE_DISNEY .. LR_SURFACE_LAYERED)
constexpr uint32_t kWfHeavyWords = 14u;      // + sampler words: d(3) tri u v beta(3) Li(3) pixel depth
constexpr uint32_t kWfContWords = 25u;       // + sampler words: ray o d (6) shadow o d tmax (7) nee(3) beta(3) Li(3) pdf pixel depth|flags
constexpr uint32_t kWfSamplerWordsMax = 4u;
#ifndef LR_WF_ITEM
#define LR_WF_ITEM 1024// (256 / 512 / 1024: 433 / 442 / 444 Msamples/s on C5 at 512 spp)
#endif
constexpr uint32_t kWfItemRecords = LR_WF_ITEM;    // continuation records per work item of the continuation pass, at most (megapath_kernel.h)
// device-side counters (uint32 each): [0..2] parked paths per kind, [3] continuation records, [4] work counter of the continuation
// pass, [5..7] work counters of the heavy kernels (one per kind)
enum : uint32_t { kWfCountHeavy = 0u, kWfCountCont = 3u, kWfWorkCont = 4u, kWfWorkHeavy = 5u, kWfCounterWords = 8u,
                  // round 6: behind the round's counters, the parked paths a slice HANDS OVER to the next one (film_kernels.h: wf_carry_kernel)
                  kWfCarry = 8u, kWfCounterBufferWords = 12u };
struct WfArgs {
    uint32_t *heavy;             // [kWfKinds][kWfHeavyWords + sampler words][capacity]
    uint32_t *cont;              // [kWfContWords + sampler words][capacity]
    uint32_t *counts;            // kWfCounterWords counters of THIS round
    uint32_t capacity;           // records per queue (>= paths of one slice: a path is parked at most once per round)
    unsigned long long *accum;   // fixed-point radiance sums [pixel][3] of the paths that finish outside their tile's wave
    float accum_scale;           // 2^k: radiance -> fixed point (the film gets accum / 2^k once per lrhip_render)
    uint32_t count_at_flush;     // pool kernels (megapool_kernel.h): samples are counted when their wave leaves the work item; 0 = when they finish
};

struct DScene {
    // acceleration structure
    const DNodeQ *nodes;
    const lr_bvh_triangle *bvh_tris;
    // geometry tables
    const DInstance *instances;
    const lr_vertex *vertices;
    const lr_triangle *triangles;
    const lr_alias_entry *tri_alias;
    const float *tri_pdf;
    const lr_light_handle *light_instances;
    // shading tables
    const DClosure *closures;
    const DSurface *surfaces;// the host's records + their constant texture slots (dynamic closures, alpha / opacity wrappers)
    const DLight *lights;
    const lr_texture *textures;
    const float *texels;
    const lr_filter *filter;
    DCamera camera;
    // environment: kEnv*; kEnvConstant is served from env_L / env_to_world, the others from *env (FULL kernels)
    uint32_t env_kind;
    float env_L[3];
    float env_to_world[9];
    // integrator / sampler / film
    uint32_t max_depth, rr_depth;
    const DShadeTri *shade_tris;// [accel.triangle_count], same order as bvh_tris
    const lr_medium *media;   // Pipeline::_media (the volumetric megakernel only)
    uint32_t env_medium_tag;  // Pipeline::environment_medium_tag, LR_INVALID_ID = none
    uint32_t integrator_kind, integrator_flags;// lr_integrator_kind / LR_DIRECT_* / LR_NORMAL_* (kFeatAux variants only)
    float rr_threshold, env_prob;
    uint32_t light_count;   // distinct Light nodes (uniform.cpp:82)
    uint32_t has_lights;
    uint32_t sampler_kind, seed;
    float film_clamp;
    uint32_t sampler_spp;          // PaddedSobol permutation length
    uint32_t sobol_scale;          // global Sobol pixel grid
    float shutter_weight;          // radiance scale of the current lrhip_render call (integrator.cpp:74: film()->accumulate(pixel, shutter_weight * L))
    uint32_t sampler_tile;         // TileShared wrapper (tile_shared.cpp): tile width | height << 16, 0 = none
    uint32_t sampler_tile_jitter;
    const uint32_t *sobol_matrices;// [1024][52]
    const uint32_t *sobol_bytes;   // [1024][7][256]: the matrices' products with every value of every index byte (Sobol sampler only)
    const uint64_t *vdc_sobol, *vdc_sobol_inv;// [52] rows for log2(sobol_scale)
    const uint64_t *vdc_bytes, *vdc_inv_bytes;// [7][256]: their products with every value of every byte (Sobol sampler, scale > 1)
    const DEnvironment *env;
    WfArgs wf;// wavefront mode only (behind the scene pointer like everything else: scalar loads where a field is used)
};

// The scene record reaches the kernels through a pointer into constant memory, not by value: as a 400-byte kernel argument
// the compiler loaded all of it into SGPRs up front and then spilled a hundred of them into VGPR lanes for the length of
// the kernel (lean variant: 99 -> 7 spilled SGPRs, 122 -> 105 spilled VGPRs, 320 -> 256 B scratch; everything-variant: 163 ->
// 75, 258 -> 98, 1168 -> 1008 B; C2 +0.8 %, C5 +2.7 %); now fields are fetched with scalar loads where they are used.
typedef const DScene __attribute__((address_space(4))) *DScenePtr;

struct DCounters {
    unsigned long long paths, closest_rays, shadow_rays, nodes_visited, tris_tested, surface_hits, nee_samples,
        path_length_sum, trace_steps, trace_steps_busy, shade_calls, shade_busy, trace_steps_starved, shade_cycles, trace_cycles, wave_cycles, nodes_empty,
        shade_light_cycles, shade_closure_cycles, shade_regen_cycles;// sections of the shading block (round 3)
    unsigned long long probe[16];// -DLR_STALL_PROBE builds only (dev_trace.h: THE STALL PROBE): section cycles summed over waves; zero otherwise
};

struct RenderArgs {
    float4 *film;          // (sum r, sum g, sum b, n) per pixel, row-major
    uint32_t spp_begin, spp_end;
    uint32_t tile_begin, tile_end, tile_stride;
    uint32_t tiles_x, tiles_y;
    uint32_t chunk_count;  // spp range split into `chunk_count` contiguous chunks per tile
    uint32_t item_count;   // tiles_in_range * chunk_count
    uint32_t *work_counter;
    float4 *partial;       // chunk_count > 1: per-chunk partial sums [chunk][pixel]
    uint32_t *spill;       // traversal stack overflow area [kSpillEntries][total_threads]
    uint32_t total_threads;
    DCounters *counters;
    // Work items shrink towards the end of the launch (round 3): the first chunk_big_count chunks of a tile's sample range hold
    // chunk_big samples per pixel each, the rest chunk_small; all big items are handed out before the first small one.  Big items
    // keep the drain at the end of an item (its last paths finish with most lanes idle) rare, small ones at the end keep the tail
    // of the launch (waves out of items while the last ones finish) short.  Uniform chunks: chunk_big_count = chunk_count.
    uint32_t chunk_big_count, chunk_big, chunk_small;
    float4 *pool;          // pool kernels (megapool_kernel.h): path state of every thread's two contexts, [context][quad][thread], Infinity-Cache resident
};

// item number -> (tile of the range, chunk, sample range) under the chunking above
struct ItemRange {
    uint32_t tile_index, chunk, s_begin, s_end;
};
__device__ __forceinline__ ItemRange item_range(const RenderArgs &args, uint32_t item) {
    const auto tiles = args.item_count / args.chunk_count;
    const auto big_items = tiles * args.chunk_big_count;
    ItemRange r;
    if (item < big_items) {
        r.tile_index = item / args.chunk_big_count;
        r.chunk = item - r.tile_index * args.chunk_big_count;
        r.s_begin = args.spp_begin + r.chunk * args.chunk_big;
        r.s_end = r.s_begin + args.chunk_big;
    } else {
        const auto small_count = args.chunk_count - args.chunk_big_count;
        const auto j = item - big_items;
        r.tile_index = j / small_count;
        const auto c = j - r.tile_index * small_count;
        r.chunk = args.chunk_big_count + c;
        r.s_begin = args.spp_begin + args.chunk_big_count * args.chunk_big + c * args.chunk_small;
        r.s_end = r.s_begin + args.chunk_small;
    }
    r.s_begin = r.s_begin < args.spp_end ? r.s_begin : args.spp_end;
    r.s_end = r.s_end < args.spp_end ? r.s_end : args.spp_end;
    return r;
}

}// namespace lrd
