/* lr_scene.h — flattened, POD scene description ("SceneBlob") shared by the host
 * scene builder, the HIP megakernel path tracer (include/lrhip.h) and the CPU oracle.
 *
 * Every table mirrors a device-side table of the reference renderer; citations are
 * relative to the LuisaRender tree (reference, read-only):
 *   lr_vertex / lr_triangle   src/util/vertex.h:37-56, compute::Triangle (3 x u32)
 *   lr_alias_entry            src/util/sampling.h:29-32
 *   lr_instance_handle        src/base/shape.cpp:46-94 (uint4 bit packing)
 *   lr_light_handle           src/base/light.h:26-29
 *   lr_filter                 src/base/filter.h:30-33, src/base/filter.cpp:24-47
 *   lr_camera                 src/cameras/pinhole.cpp:12-15, src/cameras/thin_lens.cpp:44-50
 *   lr_film                   src/films/color.cpp:25-42
 *   lr_integrator             src/integrators/mega_path.cpp:21-25
 *
 * All structs are plain C, little-endian, 4-byte scalars.  Matrices are column-major
 * (m[col*4+row]) like luisa::float4x4.
 */
#ifndef LR_SCENE_H
#define LR_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_INVALID_ID 0xffffffffu

typedef struct lr_vertex { float px, py, pz, nx, ny, nz, u, v; } lr_vertex; /* 32 B */
typedef struct lr_triangle { uint32_t i0, i1, i2; } lr_triangle;           /* 12 B */
typedef struct lr_alias_entry { float prob; uint32_t alias; } lr_alias_entry;
typedef struct lr_uint4 { uint32_t x, y, z, w; } lr_uint4;
typedef struct lr_light_handle { uint32_t instance_id, light_tag; } lr_light_handle;

/* A unique mesh (deduplicated by content, src/base/geometry.cpp:53-57).  The reference's
 * "bindless buffer base" of a mesh (vertices, triangles, alias, pdf at base+0..3,
 * src/base/shape.h:139-142) is the mesh index here; the four buffers are the slices
 * below of the global arrays. */
typedef struct lr_mesh {
    uint32_t vertex_offset, vertex_count;     /* into lr_scene.vertices            */
    uint32_t triangle_offset, triangle_count; /* into triangles / tri_alias / tri_pdf */
} lr_mesh;

/* Shape property flags, src/base/shape.h:34-39 */
enum {
    LR_SHAPE_HAS_VERTEX_NORMAL = 1u << 0,
    LR_SHAPE_HAS_VERTEX_UV = 1u << 1,
    LR_SHAPE_HAS_SURFACE = 1u << 2,
    LR_SHAPE_HAS_LIGHT = 1u << 3,
    LR_SHAPE_HAS_MEDIUM = 1u << 4,
    LR_SHAPE_MAYBE_NON_OPAQUE = 1u << 5
};

/* One TLAS instance: the packed handle of the reference plus the object->world matrix
 * the reference keeps with the acceleration structure (src/base/geometry.cpp:130,307). */
typedef struct lr_instance {
    lr_uint4 handle;       /* Shape::Handle::encode, mesh index as buffer base */
    float object_to_world[16];
    uint32_t visible;      /* TLAS visibility mask bit (geometry.cpp:130)        */
    uint32_t pad[3];
} lr_instance;

/* ---- textures (src/base/texture.cpp, src/textures/{constant,image,checkerboard}.cpp) */
enum { LR_TEX_CONSTANT = 0, LR_TEX_IMAGE = 1, LR_TEX_CHECKERBOARD = 2 };
enum { LR_TEX_ADDR_REPEAT = 0, LR_TEX_ADDR_EDGE = 1, LR_TEX_ADDR_MIRROR = 2, LR_TEX_ADDR_ZERO = 3 };
enum { LR_TEX_FILTER_POINT = 0, LR_TEX_FILTER_BILINEAR = 1 };
enum { LR_TEX_ENC_LINEAR = 0, LR_TEX_ENC_SRGB = 1, LR_TEX_ENC_GAMMA = 2 };
typedef struct lr_texture {
    uint32_t kind;
    uint32_t channels;     /* 1..4 (Texture::channels)                        */
    float v[4];            /* constant value (already multiplied by `scale`)   */
    /* image: float RGBA texels at lr_scene.texels[texel_offset ...]            */
    uint32_t width, height;
    uint64_t texel_offset; /* in float4 units                                  */
    uint32_t address, filter, encoding;
    float gamma[3];
    float uv_scale[2], uv_offset[2];
    float scale[4];        /* image.cpp `scale` (per channel)                  */
    /* checkerboard: on/off child texture ids and uv scale                      */
    int32_t child[2];
    float checker_scale;
    uint32_t pad;          /* ignored on input.  (The DEVICE copy of an image's record uses it for the texel storage: 0 = float texels;
                              else bits 0-1 the host's 8-bit conversion, bits 4-7 channels that hold one value, kept in `v`, and
                              texel_offset counts 32-bit words -- lrhip_set_texture_storage, lrhip.hip: pack_byte_textures)        */
} lr_texture;

/* ---- surfaces: one record per registered Surface node (pipeline surface tag) */
enum {
    LR_SURFACE_NULL = 0,
    LR_SURFACE_MATTE = 1,   /* src/surfaces/matte.cpp   */
    LR_SURFACE_MIRROR = 2,  /* src/surfaces/mirror.cpp  */
    LR_SURFACE_GLASS = 3,   /* src/surfaces/glass.cpp   */
    LR_SURFACE_PLASTIC = 4, /* src/surfaces/plastic.cpp (alias Substrate) */
    LR_SURFACE_METAL = 5,   /* src/surfaces/metal.cpp   */
    LR_SURFACE_DISNEY = 6,  /* src/surfaces/disney.cpp  */
    LR_SURFACE_MIX = 7,     /* src/surfaces/mix.cpp     */
    LR_SURFACE_LAYERED = 8  /* src/surfaces/layered.cpp: tex[0] thickness, tex[1] g, tex[2] albedo; u[0] top tag, u[1] bottom tag,
                             * u[2] max_depth, u[3] samples */
};
/* Composition (mix.cpp:82-212 and layered.cpp:195-500 hold arbitrary child closures): a Mix tree is at most LR_MIX_MAX_DEPTH Mix
 * levels deep below its root (u[2] = that depth; a Layered child counts as a leaf), and at most LR_LAYERED_MAX_LEVELS Layered surfaces
 * lie on any path through the interfaces (Layered inside Layered: one level; Mix trees may sit in between). */
enum { LR_MIX_MAX_DEPTH = 7, LR_LAYERED_MAX_LEVELS = 2 };
enum {
    LR_SURFACE_FLAG_REMAP_ROUGHNESS = 1u << 0,
    LR_SURFACE_FLAG_THIN = 1u << 1,
    LR_SURFACE_FLAG_TWO_SIDED = 1u << 2
};
/* texture slot meaning per kind (id < 0 = property absent -> reference default):
 *  MATTE   0 Kd, 1 sigma
 *  MIRROR  0 color, 1 roughness
 *  GLASS   0 Kr, 1 Kt, 2 roughness, 3 eta
 *  PLASTIC 0 Kd, 1 roughness, 2 sigma_a, 3 eta, 4 thickness
 *  METAL   0 Kd, 1 roughness            f[0..2] = n(R,G,B)  f[3..5] = k(R,G,B)
 *  DISNEY  0 color, 1 metallic, 2 eta, 3 roughness, 4 specular_tint, 5 anisotropic,
 *          6 sheen, 7 sheen_tint, 8 clearcoat, 9 clearcoat_gloss, 10 specular_trans,
 *          11 flatness, 12 diffuse_trans
 *  MIX     0 ratio                      u[0], u[1] = surface tags of a, b          */
typedef struct lr_surface {
    uint32_t kind;
    uint32_t flags;
    int32_t tex[16];
    float f[8];
    uint32_t u[4];
    int32_t alpha_tex;      /* OpacitySurfaceWrapper, src/base/surface.h:160-230 */
    int32_t normal_tex;     /* NormalMapWrapper, src/base/surface.h:232-310      */
    float normal_strength;
    uint32_t pad;
} lr_surface;

/* ---- lights: one record per registered Light node (pipeline light tag) */
enum { LR_LIGHT_NULL = 0, LR_LIGHT_DIFFUSE = 1 }; /* src/lights/diffuse.cpp */
typedef struct lr_light {
    uint32_t kind;
    int32_t emission_tex;
    float scale;
    uint32_t two_sided;
} lr_light;

/* ---- environment (src/environments/{spherical,directional}.cpp) */
enum { LR_ENV_NONE = 0, LR_ENV_SPHERICAL = 1, LR_ENV_DIRECTIONAL = 2,
       LR_ENV_COMBINED = 3 /* src/environments/combined.cpp: children in lr_scene.environment_children */ };
typedef struct lr_environment {
    uint32_t kind;
    int32_t emission_tex;
    float scale;
    uint32_t compensate_mis;
    float world_to_env[9];   /* 3x3, column-major: transpose of env transform's 3x3 */
    float env_to_world[9];
    /* spherical importance tables (src/environments/spherical.cpp:193-228); empty for
     * constant-emission environments */
    uint32_t map_width, map_height;          /* 2048 x 1024 */
    const lr_alias_entry *alias;             /* [h + h*w] marginal rows first */
    const float *pdf;                        /* [h*w]                          */
    /* directional */
    float direction[3];
    float cos_half_angle;
    uint32_t visible;
    /* combined: scales of children a and b (both > 0; a Combined with one live child is flattened by the host) and their
     * records, indices into lr_scene.environment_children.  A child may be a Combined node itself (combined.cpp:23-111 composes
     * freely): children precede their parents in the array (child[k] < own index for a node IN the array), at most
     * LR_ENV_MAX_COMBINED_DEPTH Combined nodes on a root-to-leaf path. */
    float child_scale[2];
    uint32_t child[2];
    uint32_t pad;
} lr_environment;
enum { LR_ENV_MAX_COMBINED_DEPTH = 4 };

/* ---- camera / filter / film / sampler / integrator */
enum { LR_CAMERA_PINHOLE = 0, LR_CAMERA_THIN_LENS = 1, LR_CAMERA_ORTHO = 2 };
typedef struct lr_camera {
    uint32_t kind;
    uint32_t width, height;
    uint32_t spp;
    float camera_to_world[16];
    float tan_half_fov;                 /* pinhole.cpp:44-46                        */
    /* thin lens (thin_lens.cpp:71-87)                                              */
    float focus_distance, lens_radius, projected_pixel_size;
    float ortho_scale;                  /* ortho.cpp                                */
    float clip_near, clip_far;          /* ClipPlaneCameraWrapper, camera.h:116-157 */
    uint32_t pad;
} lr_camera;

enum { LR_FILTER_LUT_SIZE = 64 };
typedef struct lr_filter {
    float radius;
    float shift[2];
    float pad;
    float lut[LR_FILTER_LUT_SIZE];
    float pdf[LR_FILTER_LUT_SIZE - 1];
    float alias_prob[LR_FILTER_LUT_SIZE - 1];
    uint32_t alias_index[LR_FILTER_LUT_SIZE - 1];
    uint32_t pad2[3];
} lr_filter;

typedef struct lr_film {
    float scale[3];   /* 2^exposure, color.cpp:39-41 */
    float clamp;      /* color.cpp:42                */
} lr_film;

enum { LR_SAMPLER_INDEPENDENT = 0, LR_SAMPLER_SOBOL = 1, LR_SAMPLER_PADDED_SOBOL = 2, LR_SAMPLER_PCG32 = 3 };
enum { LR_SOBOL_DIMENSIONS = 1024, LR_SOBOL_MATRIX_SIZE = 52 }; /* src/util/sobolmatrices.h:13-14 */
typedef struct lr_sampler {
    uint32_t kind;
    uint32_t seed;    /* src/base/sampler.cpp:9-11, default 19980810 */
    uint32_t spp;     /* Sampler::Instance::reset(.., spp): PaddedSobol permutation length (padded_sobol.cpp:112) */
    uint32_t scale;   /* next_pow2(max(W, H)): global Sobol pixel grid (sobol.cpp:119-120) */
    /* tables of the Sobol samplers (NULL for the others): generator matrices [1024][52] and the two
     * van-der-Corput rows [52] for m = log2(scale) (sobol.cpp:121-129); derived by tools/gen_sobol_tables.py */
    const uint32_t *sobol_matrices;
    const uint64_t *vdc_sobol;
    const uint64_t *vdc_sobol_inv;
    /* TileShared wrapper (src/samplers/tile_shared.cpp:44-62): tile_size[0] != 0 -> the base sampler above is started with the TILE
     * of the pixel, (pixel [+ jitter]) / tile_size, so all pixels of a tile share one sample sequence; `scale` is then that of the
     * tile grid (the wrapper resets its base with the tile count as the resolution).  tile_jitter: the pixel is first moved by
     * uint2(float2(h >> 16, h & 0xffff) * 2^-16 * resolution) % resolution, h = xxhash32(sample index) (:52-56). */
    uint32_t tile_size[2];
    uint32_t tile_jitter;
    uint32_t tile_pad;
} lr_sampler;

typedef enum lr_integrator_kind {
    LR_INTEGRATOR_MEGAPATH = 0, /* src/integrators/mega_path.cpp (the hot path)                                  */
    LR_INTEGRATOR_DIRECT = 1,   /* src/integrators/direct.cpp:66-200: one bounce, light / surface / both sampling  */
    LR_INTEGRATOR_NORMAL = 2,   /* src/integrators/normal.cpp:36-70: geometric / shading normal visualiser         */
    LR_INTEGRATOR_VPT_NAIVE = 3 /* src/integrators/mega_vpt_naive.cpp:68-483: volumetric megakernel  */
} lr_integrator_kind;
enum {
    LR_DIRECT_SAMPLE_LIGHTS = 1u,   /* importance_sampling "light" | "both"   (direct.cpp:27-42) */
    LR_DIRECT_SAMPLE_SURFACES = 2u, /* importance_sampling "surface" | "both"                    */
    LR_NORMAL_REMAP = 1u,           /* normal.cpp:21: ns * .5 + .5                                */
    LR_NORMAL_SHADING = 2u          /* normal.cpp:22: shading normal of the closure instead of ng */
};

typedef struct lr_integrator {
    uint32_t max_depth;   /* mega_path.cpp:23 */
    uint32_t rr_depth;    /* :24 */
    float rr_threshold;   /* :25 */
    float env_prob;       /* UniformLightSampler::_env_prob, uniform.cpp:39-48 */
    uint32_t light_count; /* pipeline.lights().size(): number of distinct Light nodes, uniform.cpp:82 */
    uint32_t kind;        /* lr_integrator_kind */
    uint32_t flags;       /* LR_DIRECT_* / LR_NORMAL_* */
    uint32_t environment_medium_tag; /* Pipeline::environment_medium_tag (pipeline.cpp:77-79); LR_INVALID_ID = none */
} lr_integrator;

/* Participating media (src/base/medium.h, src/media/{homogeneous,vacuum,null}.cpp) with their phase function
 * (src/phasefunctions/henyey_greenstein.cpp).  Reached from the VPT integrator only; a Null medium registers nothing
 * (geometry.cpp:139).  Coefficients are the textures' constant values (homogeneous.cpp:196-199 requires constants),
 * decoded as unbounded sRGB spectra = the values themselves.                                                          */
typedef enum lr_medium_kind { LR_MEDIUM_VACUUM = 0, LR_MEDIUM_HOMOGENEOUS = 1 } lr_medium_kind;
#define LR_MEDIUM_VACUUM_PRIORITY 0xffffffffu /* Medium::VACUUM_PRIORITY, medium.h:27 */
typedef struct lr_medium {  /* 64 B */
    uint32_t kind;
    uint32_t priority;      /* medium.cpp:13 (`priority`, default 0); Vacuum: VACUUM_PRIORITY (vacuum.cpp:69) */
    float eta;              /* homogeneous.cpp:192 */
    float g;                /* HenyeyGreenstein::_g clamped to [-1, 1], henyey_greenstein.cpp:64 */
    float sigma_a[3], sigma_s[3], le[3];
    uint32_t pad[3];
} lr_medium;

/* ---- wide BVH for the HIP traversal kernel (built by the host library; the CPU oracle
 * ignores it and builds its own canonical BVH2).  See DESIGN.md "BVH layout". */
typedef struct lr_bvh4_node {   /* 128 B, 8 x float4: SoA over the 4 children */
    float lo_x[4], lo_y[4], lo_z[4];
    float hi_x[4], hi_y[4], hi_z[4];
    /* child reference: bit31 = leaf; inner: node index; leaf: triangle index (27 bits), bits 27..30 =
     * count-1 which MUST be 0 — the kernel's leaf step tests exactly one triangle (lrhip_upload_scene rejects
     * anything else); 0xffffffff = empty slot */
    uint32_t child[4];
    uint32_t pad[4];
} lr_bvh4_node;

typedef struct lr_bvh_triangle { /* 48 B: world-space, pre-transformed */
    float v0[3];
    uint32_t inst;
    float e1[3];
    uint32_t prim;
    float e2[3];
    uint32_t flags;   /* bit0: instance visible (camera+shadow rays), bit1: opaque */
} lr_bvh_triangle;

typedef struct lr_accel {
    const lr_bvh4_node *nodes;
    uint32_t node_count;
    const lr_bvh_triangle *triangles;
    uint32_t triangle_count;
    float world_min[3], world_max[3];
} lr_accel;

/* ---- the blob: pointers + counts; the producer owns all memory */
typedef struct lr_scene {
    const lr_vertex *vertices;         uint64_t vertex_count;
    const lr_triangle *triangles;      uint64_t triangle_count;
    const lr_alias_entry *tri_alias;   /* [triangle_count], per-mesh tables, geometry.cpp:78-79 */
    const float *tri_pdf;              /* [triangle_count]                                        */
    const lr_mesh *meshes;             uint32_t mesh_count;
    const lr_instance *instances;      uint32_t instance_count;
    const lr_light_handle *light_instances; uint32_t light_instance_count; /* geometry.cpp:149-153 */
    const lr_surface *surfaces;        uint32_t surface_count;
    const lr_light *lights;            uint32_t light_count;
    const lr_texture *textures;        uint32_t texture_count;
    const float *texels;               uint64_t texel_count;   /* float4 units */
    lr_environment environment;
    lr_camera camera;
    lr_filter filter;
    lr_film film;
    lr_sampler sampler;
    lr_integrator integrator;
    lr_accel accel;                    /* nodes == NULL when not built */
    uint32_t any_non_opaque;           /* Geometry::_any_non_opaque, geometry.cpp:124 */
    uint32_t environment_child_count;  /* 2 when environment.kind == LR_ENV_COMBINED, else 0 */
    const lr_environment *environment_children; /* records below a Combined root (leaves with their own tables, nested Combined nodes) */
    const lr_medium *media;            /* Pipeline::_media in registration order (tag = index) */
    uint32_t medium_count;
    uint32_t pad_media;
} lr_scene;

#ifdef __cplusplus
}
#endif
#endif /* LR_SCENE_H */
