/* oracle.h — C API of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  A CPU restatement of the reference's megakernel path tracer
 * (src/integrators/mega_path.cpp:49-156 driven by src/base/integrator.cpp:51-113) on the
 * flattened tables of include/lr_scene.h.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; it is the checker, never the product.
 *
 * PARITY PINNED TO THE REFERENCE (round 2).  The reference's own binary still cannot be built (LuisaCompute, LLVM, Embree are
 * absent), but its own RENDER CODE can: oracle/Makefile.ref compiles /root/reference/src/{util,sdl,base}/*.cpp and its plugins
 * where they lie, against a scalar stand-in for the LuisaCompute DSL (oracle/ref_shim), into oracle/_ref/libref.so.
 * tests/test_oracle_vs_ref.py holds this oracle against it -- hashes, generators, alias tables, filter tables, instance handles,
 * every closure's evaluate / sample, the per-sample Li of MegaPath / Direct / Normal / MegaVPTNaive over a scene corpus and whole
 * frames: bit for bit (Layered: to 2e-6) -- and tests/golden/ref_*.npz are frames the reference's code rendered, which
 * tests/test_ref_golden.py compares with this oracle (bit for bit, CPU) and with the HIP path (GPU box).  What libref cannot
 * pin is what LuisaCompute itself supplies (builtins, the ray-tracing unit, texture filtering): stated in oracle/ref_shim.
 * The older pins stay: bit-exact KATs against independent implementations (python xxhash, published PCG32 vectors), alias-table
 * frequency tests, BSDF energy / reciprocity tests and closed-form furnace scenes -- all under tests/.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/lr_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

typedef struct oracle_counters {
    uint64_t paths;          /* samples traced                                  */
    uint64_t closest_rays;   /* Geometry::trace_closest calls                   */
    uint64_t shadow_rays;    /* Geometry::trace_any calls                       */
    uint64_t nodes_visited;  /* canonical BVH2 nodes fetched (64 B each)        */
    uint64_t tris_tested;    /* triangle tests (48 B each)                      */
    uint64_t surface_hits;   /* interactions reconstructed (188 B each)         */
    uint64_t nee_samples;    /* light samples drawn (208 B each)                */
    uint64_t path_length_sum;/* sum over paths of surface bounces taken          */
} oracle_counters;

/* builds the canonical two-level BVH2 over `scene`; the scene tables must outlive the ctx */
oracle_ctx *oracle_create(const lr_scene *scene);
void oracle_destroy(oracle_ctx *ctx);
/* Motion blur: the reference renders one Camera::ShutterSample after the other (src/base/integrator.cpp:91-95), the scene moved
 * to the sample's time and the radiance scaled by the sample's weight (:74).  The caller moves the scene (lrhost_scene_set_time),
 * creates a ctx over the moved tables and sets the weight (default 1) before oracle_render of that sample's spp range. */
void oracle_set_shutter_weight(oracle_ctx *ctx, float weight);

/* Accumulates samples [spp_begin, spp_end) of every pixel of the rectangle
 * [x0, x1) x [y0, y1) into `film` (float4[W*H] = (sum r, sum g, sum b, n), the reference's
 * film layout src/films/color.cpp:107-123), in sample order per pixel, with `threads`
 * std::threads over rows.  Counters are added to *counters when non-NULL.           */
int oracle_render(oracle_ctx *ctx, uint32_t spp_begin, uint32_t spp_end,
                  uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                  int threads, float *film, oracle_counters *counters);

/* convert kernel of the Color film (src/films/color.cpp:87-93) */
void oracle_film_convert(const lr_scene *scene, const float *film, float *rgba_out);

/* radiance of one sample (no film clamp), for pixel-level debugging and tests */
void oracle_li(oracle_ctx *ctx, uint32_t px, uint32_t py, uint32_t sample_index, float rgb_out[3]);

/* closest hit of an explicit ray: out = {inst, prim, bary.u, bary.v, t}; inst = ~0u on miss */
void oracle_trace_closest(oracle_ctx *ctx, const float origin[3], const float direction[3],
                          float t_min, float t_max, uint32_t out_ids[2], float out_bary_t[3]);
/* primary camera ray of (pixel, sample): origin[3], direction[3], weight */
void oracle_camera_ray(oracle_ctx *ctx, uint32_t px, uint32_t py, uint32_t sample_index, float out[7]);

/* first `n` generate_1d() draws of the scene's sampler for (pixel, sample): out[0..1] = generate_pixel_2d(),
 * out[2 .. 2+n) = the following generate_1d() values */
void oracle_sampler_stream(const lr_scene *scene, uint32_t px, uint32_t py, uint32_t sample_index, uint32_t n, float *out);

/* ---- unit-level hooks used by the KAT tests */
uint32_t oracle_xxhash32_1(uint32_t x);
uint32_t oracle_xxhash32_2(uint32_t x, uint32_t y);
uint32_t oracle_xxhash32_3(uint32_t x, uint32_t y, uint32_t z);
uint32_t oracle_xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w);
float oracle_lcg(uint32_t *state);
uint32_t oracle_pcg32_next(uint64_t *state, uint64_t *inc);
void oracle_pcg32_seed(uint64_t seq_index, uint64_t *state, uint64_t *inc);
void oracle_create_alias_table(const float *values, uint32_t n, lr_alias_entry *table, float *pdf);
void oracle_sample_alias_table(const lr_alias_entry *table, uint32_t n, float u, uint32_t *index, float *u_remapped);
void oracle_filter_sample(const lr_filter *filter, float ux, float uy, float out_offset_weight[3]);
void oracle_encode_handle(uint32_t buffer_base, uint32_t flags, uint32_t surface_tag, uint32_t light_tag,
                          uint32_t medium_tag, uint32_t tri_count, float shadow_term, float isect_offset,
                          uint32_t out[4]);
void oracle_offset_ray_origin(const float p[3], const float n[3], float out[3]);
/* BSDF evaluation of surface `surface_tag` on a flat patch with geometric normal +z and the given
 * shading normal; directions in world space.  out = {f.r, f.g, f.b, pdf} */
void oracle_surface_evaluate(const lr_scene *scene, uint32_t surface_tag, const float ns[3],
                             const float wo[3], const float wi[3], float out[4]);
/* out = {f.r, f.g, f.b, pdf, wi.x, wi.y, wi.z, event} */
void oracle_surface_sample(const lr_scene *scene, uint32_t surface_tag, const float ns[3],
                           const float wo[3], float u_lobe, float ux, float uy, float out[8]);

#ifdef __cplusplus
}
#endif
#endif /* ORACLE_H */
