// lrhip.hip — C ABI (include/lrhip.h) of the gfx950 megakernel path tracer: context, scene
// upload into HBM, launches, film read-back.  Device code lives in the dev_*.h headers and
// megapath_kernel.h.  Written for gfx950 only; no host fallback exists — without a HIP device
// every entry point fails with LRHIP_ERROR_DEVICE.
#include "../../../include/lrhip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <vector>
#include <functional>

#include "film_kernels.h"
#include "megapath_kernel.h"
#include "megapool_kernel.h"
#include "variants.h"

// one translation unit per precompiled megakernel variant (megapath_variant.hip, -DLR_VARIANT=<mask>); weak, so that
// experimental builds may leave variants out (tools/ab.sh) — a missing one is reported, never silently replaced
#define LR_DECLARE_VARIANT(mask)                                                                                       \
    extern "C" __attribute__((weak)) hipError_t lrhip_variant_launch_##mask(unsigned, hipStream_t, const lrd::DScene *, const lrd::RenderArgs *); \
    extern "C" __attribute__((weak)) hipError_t lrhip_variant_occupancy_##mask(int *);
LR_VARIANT_LIST(LR_DECLARE_VARIANT)
LR_PADDED_LIST(LR_DECLARE_VARIANT)
#undef LR_DECLARE_VARIANT
// the heavy-closure kernels of wavefront mode, one translation unit each (heavy_variant.hip, -DLR_HVARIANT=<mask>)
#define LR_HEAVY_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(516) X(517) X(518) X(519) X(520) X(521) X(522) X(523)
#define LR_HEAVY_DECL(mask)                                                                                                   \
    extern "C" __attribute__((weak)) hipError_t lrhip_heavy_launch_##mask(unsigned, hipStream_t, const lrd::DScene *, const lrd::RenderArgs *); \
    extern "C" __attribute__((weak)) hipError_t lrhip_heavy_occupancy_##mask(int *);
LR_HEAVY_LIST(LR_HEAVY_DECL)
#undef LR_HEAVY_DECL

namespace {

struct VariantEntry {
    uint32_t mask;
    hipError_t (*launch)(unsigned, hipStream_t, const lrd::DScene *, const lrd::RenderArgs *);
    hipError_t (*occupancy)(int *);
};
#define LR_VARIANT_ENTRY(mask) VariantEntry{mask##u, lrhip_variant_launch_##mask, lrhip_variant_occupancy_##mask},
const VariantEntry kVariants[] = {LR_VARIANT_LIST(LR_VARIANT_ENTRY)};
// the lean pool kernels compiled for the PaddedSobol sampler (kFeatPadded, variants.h), looked up by mask
const VariantEntry kPaddedVariants[] = {LR_PADDED_LIST(LR_VARIANT_ENTRY)};
constexpr size_t kPaddedVariantCount = sizeof(kPaddedVariants) / sizeof(kPaddedVariants[0]);
#undef LR_VARIANT_ENTRY
static_assert(sizeof(kVariants) / sizeof(kVariants[0]) == lrd::kSceneVariantCount * 4u, "variants.h and kSceneVariants disagree");

// (mask: bit 0 counters, bit 1 generic sampler, bits 2-3 closure kind (0 Disney, 4 Mix, 8 Layered), 512 nested Mix / Layered)
#define LR_HEAVY_ENTRY(mask) VariantEntry{mask##u, lrhip_heavy_launch_##mask, lrhip_heavy_occupancy_##mask},
const VariantEntry kHeavyVariants[] = {LR_HEAVY_LIST(LR_HEAVY_ENTRY)};
#undef LR_HEAVY_ENTRY
int find_variant(const VariantEntry *table, size_t n, uint32_t mask) {
    for (size_t k = 0; k < n; k++) {
        if (table[k].mask == mask) { return static_cast<int>(k); }
    }
    return -1;
}

// smallest precompiled superset of the scene's feature bits (+ the count / generic-sampler bits, which are exact) among the
// variants of one scheduler: `pool` = the path-pool kernels of round 4 (megapool_kernel.h), otherwise the one-path-per-lane kernels
// `byte_texels`: the scene holds packed 8-bit texels -- only a kernel that decodes them will do (the lean ones of the kFeatByteTex bit, and every
// variant that makes real calls: dev_wavefront.h); a scene without them never takes a kFeatByteTex kernel
bool decodes_byte_texels(uint32_t mask) { return (mask & (lrd::kFeatByteTex | lrd::kFeatMix | lrd::kFeatLayered | lrd::kFeatVpt)) != 0u; }
int pick_variant(uint32_t scene_features, bool count, bool generic, bool pool = false, bool byte_texels = false) {
    for (uint32_t i = 0u; i < lrd::kSceneVariantCount; i++) {
        if (((lrd::kSceneVariants[i] & lrd::kFeatPool) != 0u) != pool) { continue; }
        if (byte_texels ? !decodes_byte_texels(lrd::kSceneVariants[i]) : (lrd::kSceneVariants[i] & lrd::kFeatByteTex) != 0u) { continue; }
        if ((lrd::kSceneVariants[i] & scene_features) == scene_features) {
            auto mask = lrd::kSceneVariants[i] | (count ? lrd::kFeatCount : 0u) | (generic ? lrd::kFeatGeneric : 0u);
            for (uint32_t k = 0u; k < lrd::kSceneVariantCount * 4u; k++) {
                if (kVariants[k].mask == mask) { return static_cast<int>(k); }
            }
        }
    }
    return -1;
}

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define LR_HIP_CHECK(expr)                                                                                   \
    do {                                                                                                     \
        auto err_ = (expr);                                                                                  \
        if (err_ != hipSuccess) {                                                                            \
            return fail(LRHIP_ERROR_DEVICE, std::string{#expr} + ": " + hipGetErrorString(err_));            \
        }                                                                                                    \
    } while (0)

struct DeviceBuffer {
    void *ptr{nullptr};
    size_t bytes{0};
    void release() {
        if (ptr != nullptr) { (void)hipFree(ptr); }
        ptr = nullptr, bytes = 0;
    }
};

// persistent-grid sizing inputs that must not depend on the device actually present, so that the
// chunking (and therefore the fp32 summation order of the film) is identical on every GPU
constexpr double kNominalWaves = 4096.0;// 256 CUs x 4 SIMDs x 4 waves
constexpr uint32_t kMaxChunks = 64u;     // partial planes: chunk_count x 16 B per pixel
constexpr uint32_t kWfCarryRounds = 1u;// wavefront mode: rounds of a slice before its parked paths wait for the next slice (film_kernels.h: wf_carry_kernel)
constexpr uint64_t kWfQueueBudget = 104ull << 30u;// bytes the queues of wavefront mode may take (of 288 GB; round 6: 104 GB -- the default slice with its hand-over margin takes 86 - 100)
#ifndef LR_MAX_BLOCKS_PER_CU
#define LR_MAX_BLOCKS_PER_CU 8
#endif
constexpr uint32_t kMaxBlocksPerCu = LR_MAX_BLOCKS_PER_CU; // resident 256-thread blocks per CU the persistent grid may use (-D: A/B builds)

}// namespace

struct lrhip_ctx {
    int device{0};
    hipStream_t stream{nullptr};
    bool own_stream{true};
    hipEvent_t ev_begin{nullptr}, ev_end{nullptr};
    bool timed{false};
    uint32_t byte_textures{1u};  // 8-bit images as 8-bit texels on the device (lrhip_upload_scene): lrhip_set_texture_storage: 0 never, 1 = where the scene's float texels exceed kByteTextureFloatBytes, 2 always
    uint64_t packed_texel_words{0u};// texels of the uploaded scene held as 8-bit codes (lrhip_packed_texels)
    uint64_t texel_bytes{0u};       // bytes of the texel table on the device (float texels of the images that stay float + the packed words)
    bool in_split{false};        // lrhip_render is rendering a call in sample sub-ranges (below): the first sub-range's begin event stands for the call
    std::vector<DeviceBuffer> scene_buffers;
    lrd::DScene scene{};
    bool scene_ready{false};
    uint32_t width{0}, height{0};
    float film_scale[3]{1.f, 1.f, 1.f};
    DeviceBuffer film_own, converted, partial, spill, counters, work_counter;
    DeviceBuffer scene_record;// lrd::DScene in device memory: the kernels read it through scalar loads (dev_scene.h: DScenePtr)
    float4 *film{nullptr};// bound film (own or external)
    float4 *film_external{nullptr};// lrhip_bind_film's buffer; kept across uploads of the same resolution
    uint32_t film_external_w{0}, film_external_h{0};
    uint32_t grid_blocks{0};
    uint32_t cu_count{0};
    uint32_t bvh_depth{0};
    uint32_t update_counts[7]{};// table sizes of the uploaded scene: what lrhip_update_scene checks its argument against
    uint32_t last_variant{0u};// feature mask of the kernel the last lrhip_render launched
    uint32_t features{0u};// lrd::kFeat* bits the uploaded scene needs (environment, alpha test, Disney / Mix / Layered)
    bool env_tree{false};// Combined environments nested in each other: only the call-making variants walk them (dev_shade.h)
    int variant_blocks[lrd::kSceneVariantCount * 4u];// resident blocks per CU of each precompiled variant (-1: not asked yet)
    int padded_blocks[32];// ... and of the kFeatPadded kernels (kPaddedVariants)
    uint32_t diag_force_features{0u};// lrhip_set_diagnostics (tests / tools)
    double diag_item_scale{0.};
    // wavefront mode (dev_scene.h: WfArgs): queues, counters and the fixed-point radiance sums; sized on first use
    DeviceBuffer wf_heavy, wf_cont, wf_counts, wf_accum;
    int heavy_blocks[sizeof(kHeavyVariants) / sizeof(kHeavyVariants[0])];// resident blocks per CU of each heavy-kernel variant (-1: not asked yet)
    uint32_t wf_mode{0u};        // lrhip_set_wavefront: 0 = automatic (scenes with Mix / Layered surfaces), 1 = never, 2 = automatic with tiny tile groups (tests)
    uint32_t wf_slice_paths{0u}; // paths per slice (queue capacity); 0 = default
    uint32_t diag_wf_carry_rounds{0u};// lrhip_set_diagnostics: rounds before a slice hands its parked paths over (0 = kWfCarryRounds; 65535 = never: every slice drains)
    // round 4: the path-pool scheduler (megapool_kernel.h): slot records of every resident wave; lrhip_set_scheduler
    DeviceBuffer pool;
    uint32_t scheduler{0u};      // lrhip_set_scheduler: 0 = automatic (wants_pool below), 1 = one path per lane, 2 = the pool kernels where one exists for the scene
};

namespace {

template<typename T>
int upload(lrhip_ctx *ctx, const T *host, size_t count, const T **device) {
    DeviceBuffer b;
    b.bytes = std::max<size_t>(count * sizeof(T), 16u);
    LR_HIP_CHECK(hipMalloc(&b.ptr, b.bytes));
    ctx->scene_buffers.emplace_back(b);
    if (count != 0u) { LR_HIP_CHECK(hipMemcpy(b.ptr, host, count * sizeof(T), hipMemcpyHostToDevice)); }
    *device = static_cast<const T *>(b.ptr);
    return LRHIP_OK;
}

int ensure(DeviceBuffer &b, size_t bytes) {
    if (b.bytes >= bytes) { return LRHIP_OK; }
    b.release();
    LR_HIP_CHECK(hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return LRHIP_OK;
}

void release_scene(lrhip_ctx *ctx) {
    for (auto &b : ctx->scene_buffers) { b.release(); }
    ctx->scene_buffers.clear();
    ctx->scene_ready = false;
    // the queues of wavefront mode and the pool kernels' state records are sized for the scene (and the memory free at the time): a new
    // scene starts without them (76-89 GB kept until lrhip_destroy starved later allocations on the same GPU)
    ctx->wf_heavy.release(), ctx->wf_cont.release(), ctx->pool.release();
}

// 3x3 inverse-transpose, same arithmetic as luisa::inverse(float3x3) + transpose (geometry.cpp:378)
void normal_matrix(const float *m, float out[9]) {
    float a[3][3];// a[c][r]
    for (auto c = 0; c < 3; c++) {
        for (auto r = 0; r < 3; r++) { a[c][r] = m[c * 4 + r]; }
    }
    auto one_over_det = 1.0f / (a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) -
                                a[1][0] * (a[0][1] * a[2][2] - a[2][1] * a[0][2]) +
                                a[2][0] * (a[0][1] * a[1][2] - a[1][1] * a[0][2]));
    float inv[3][3];// inv[c][r]
    inv[0][0] = (a[1][1] * a[2][2] - a[2][1] * a[1][2]) * one_over_det;
    inv[0][1] = (a[2][1] * a[0][2] - a[0][1] * a[2][2]) * one_over_det;
    inv[0][2] = (a[0][1] * a[1][2] - a[1][1] * a[0][2]) * one_over_det;
    inv[1][0] = (a[2][0] * a[1][2] - a[1][0] * a[2][2]) * one_over_det;
    inv[1][1] = (a[0][0] * a[2][2] - a[2][0] * a[0][2]) * one_over_det;
    inv[1][2] = (a[1][0] * a[0][2] - a[0][0] * a[1][2]) * one_over_det;
    inv[2][0] = (a[1][0] * a[2][1] - a[2][0] * a[1][1]) * one_over_det;
    inv[2][1] = (a[2][0] * a[0][1] - a[0][0] * a[2][1]) * one_over_det;
    inv[2][2] = (a[0][0] * a[1][1] - a[1][0] * a[0][1]) * one_over_det;
    for (auto c = 0; c < 3; c++) {// transpose: column c of the result = row c of inv
        for (auto r = 0; r < 3; r++) { out[c * 3 + r] = inv[r][c]; }
    }
}

// fp32 child boxes -> 64-byte quantised packet; conservative: decoded lo <= lo, decoded hi >= hi in the
// same fp32 fma the kernel uses (dev_trace.h)
// An empty slot gets inverted planes (lo 255, hi 0) AND the reference of the sentinel leaf (`empty_ref`: a triangle nothing hits, behind
// the last baked triangle): the kernel tests no child word, an empty slot misses wherever the node has an extent and costs one
// wasted triangle test where it has none.
lrd::DNodeQ quantise_node(const lr_bvh4_node &n, uint32_t empty_ref) {
    lrd::DNodeQ q{};
    const float *lo[3] = {n.lo_x, n.lo_y, n.lo_z};
    const float *hi[3] = {n.hi_x, n.hi_y, n.hi_z};
    float origin[3], scale[3];
    uint32_t plo[3] = {0u, 0u, 0u}, phi[3] = {0u, 0u, 0u};
    for (auto a = 0; a < 3; a++) {
        auto mn = std::numeric_limits<float>::max(), mx = -std::numeric_limits<float>::max();
        for (auto c = 0; c < 4; c++) {
            if (n.child[c] == LR_INVALID_ID) { continue; }
            mn = std::min(mn, lo[a][c]), mx = std::max(mx, hi[a][c]);
        }
        if (mn > mx) { mn = mx = 0.f; }
        origin[a] = mn;
        auto sc = (mx - mn) / 255.f;
        while (sc > 0.f && std::fmaf(255.f, sc, mn) < mx) { sc = std::nextafter(sc, std::numeric_limits<float>::max()); }
        scale[a] = sc;
        for (auto c = 0; c < 4; c++) {
            uint32_t ql = 255u, qh = 0u;// empty slot: inverted
            if (n.child[c] != LR_INVALID_ID) {
                if (sc > 0.f) {
                    auto fl = std::floor((static_cast<double>(lo[a][c]) - mn) / sc), fh = std::ceil((static_cast<double>(hi[a][c]) - mn) / sc);
                    ql = static_cast<uint32_t>(std::clamp(fl, 0.0, 255.0)), qh = static_cast<uint32_t>(std::clamp(fh, 0.0, 255.0));
                    while (ql > 0u && std::fmaf(static_cast<float>(ql), sc, mn) > lo[a][c]) { ql--; }
                    while (qh < 255u && std::fmaf(static_cast<float>(qh), sc, mn) < hi[a][c]) { qh++; }
                } else {
                    ql = qh = 0u;
                }
            }
            plo[a] |= ql << (8u * static_cast<uint32_t>(c));
            phi[a] |= qh << (8u * static_cast<uint32_t>(c));
        }
    }
    q.origin[0] = origin[0], q.origin[1] = origin[1], q.origin[2] = origin[2];
    q.scale_x = scale[0], q.scale_y = scale[1], q.scale_z = scale[2];
    q.lo_x = plo[0], q.lo_y = plo[1], q.lo_z = plo[2];
    q.hi_x = phi[0], q.hi_y = phi[1], q.hi_z = phi[2];
    for (auto c = 0; c < 4; c++) { q.child[c] = n.child[c] == LR_INVALID_ID ? empty_ref : n.child[c]; }
    return q;
}

// depth of the tree; 0 if a leaf holds more than one triangle (the kernel's leaf step tests exactly one)
uint32_t bvh_depth(const lr_accel &accel) {
    std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 1u}};
    auto depth = 0u;
    while (!stack.empty()) {
        auto [node, d] = stack.back();
        stack.pop_back();
        depth = std::max(depth, d);
        for (auto c : accel.nodes[node].child) {
            if (c == LR_INVALID_ID) { continue; }
            if (!(c & 0x80000000u)) { stack.emplace_back(c, d + 1u); }
            else if (((c >> 27u) & 15u) != 0u) { return 0u; }
        }
    }
    return depth;
}

// ---- the tables that depend on the scene time (Pipeline::update / Geometry::update): built once per upload and again per
// lrhip_update_scene, which copies them over the existing device buffers
std::vector<lrd::DNodeQ> build_packed_nodes(const lr_scene *s) {
    std::vector<lrd::DNodeQ> packed(s->accel.node_count);
    const auto empty_ref = lrd::kLeafFlag | s->accel.triangle_count;// the sentinel of build_padded_triangles
    for (uint32_t i = 0; i < s->accel.node_count; i++) { packed[i] = quantise_node(s->accel.nodes[i], empty_ref); }
    return packed;
}

std::vector<uint8_t> build_padded_triangles(const lr_scene *s) {// the baked triangles + the all-zero sentinel the empty node slots name (flags 0: never hit)
    std::vector<uint8_t> out((static_cast<size_t>(s->accel.triangle_count) + 1u) * sizeof(lr_bvh_triangle), 0u);
    std::memcpy(out.data(), s->accel.triangles, static_cast<size_t>(s->accel.triangle_count) * sizeof(lr_bvh_triangle));
    return out;
}

std::vector<lrd::DInstance> build_instances(const lr_scene *s) {// one 128-byte line each
    std::vector<lrd::DInstance> instances(s->instance_count);
    for (uint32_t i = 0; i < s->instance_count; i++) {
        auto &src = s->instances[i];
        auto &dst = instances[i];
        std::memset(&dst, 0, sizeof(dst));
        dst.handle[0] = src.handle.x, dst.handle[1] = src.handle.y, dst.handle[2] = src.handle.z, dst.handle[3] = src.handle.w;
        auto m = src.object_to_world;
        for (auto r = 0; r < 3; r++) { dst.c0[r] = m[r], dst.c1[r] = m[4 + r], dst.c2[r] = m[8 + r], dst.t[r] = m[12 + r]; }
        float nm[9];
        normal_matrix(m, nm);
        for (auto r = 0; r < 3; r++) { dst.n0[r] = nm[r], dst.n1[r] = nm[3 + r], dst.n2[r] = nm[6 + r]; }
        auto &mesh = s->meshes[src.handle.x >> 10u];
        dst.vertex_offset = mesh.vertex_offset;
        dst.triangle_offset = mesh.triangle_offset;
    }
    return instances;
}

// shading records of the baked triangles (dev_scene.h: DShadeTri), in BVH triangle order; empty + error text on bad references
std::vector<lrd::DShadeTri> build_shade_tris(const lr_scene *s, const std::vector<lrd::DInstance> &instances, std::string &error) {
    std::vector<lrd::DShadeTri> shade(s->accel.triangle_count);
    for (uint32_t i = 0; i < s->accel.triangle_count; i++) {
        auto &bt = s->accel.triangles[i];
        if (bt.inst >= s->instance_count) { error = "BVH triangle references an unknown instance"; return {}; }
        auto &inst = s->instances[bt.inst];
        auto &di = instances[bt.inst];
        auto &mesh = s->meshes[inst.handle.x >> 10u];
        if (bt.prim >= mesh.triangle_count) { error = "BVH triangle references an unknown primitive"; return {}; }
        auto tri = s->triangles[mesh.triangle_offset + bt.prim];
        const lr_vertex *v[3] = {s->vertices + mesh.vertex_offset + tri.i0, s->vertices + mesh.vertex_offset + tri.i1, s->vertices + mesh.vertex_offset + tri.i2};
        auto &r = shade[i];
        std::memset(&r, 0, sizeof(r));
        for (auto c = 0; c < 3; c++) { r.p0[c] = bt.v0[c], r.e1[c] = bt.e1[c], r.e2[c] = bt.e2[c]; }
        float *n[3] = {r.n0, r.n1, r.n2};
        for (auto k = 0; k < 3; k++) {
            for (auto c = 0; c < 3; c++) { n[k][c] = di.n0[c] * v[k]->nx + di.n1[c] * v[k]->ny + di.n2[c] * v[k]->nz; }
        }
        r.uv0x = v[0]->u, r.uv0y = v[0]->v, r.uv1x = v[1]->u, r.uv1y = v[1]->v, r.uv2x = v[2]->u, r.uv2y = v[2]->v;
        r.flags = inst.handle.x & 1023u, r.tags = inst.handle.y, r.offset_bits = inst.handle.w;
        // (round 6) bits 10-11: 1 + the heavy-closure kind of the triangle's surface (Disney 1, Mix 2, Layered 3; 0: a basic closure) -- what the
        // lean passes of wavefront mode park a hit by, read with the record instead of through a dependent gather of the closure table
        if ((r.flags & LR_SHAPE_HAS_SURFACE) != 0u) {
            const auto tag = (inst.handle.y >> 12u) & 4095u;
            if (tag < s->surface_count && s->surfaces[tag].kind >= LR_SURFACE_DISNEY) { r.flags |= (s->surfaces[tag].kind - LR_SURFACE_DISNEY + 1u) << 10u; }
        }
        r.tri_pdf = s->tri_pdf[mesh.triangle_offset + bt.prim];
        r.inst = bt.inst, r.prim = bt.prim, r.tri_offset = mesh.triangle_offset;
    }
    return shade;
}

void set_camera(lrd::DScene &d, const lr_scene *s) {
    auto &cam = d.camera;
    cam.kind = s->camera.kind, cam.width = s->camera.width, cam.height = s->camera.height;
    std::memcpy(cam.c2w, s->camera.camera_to_world, sizeof(cam.c2w));
    cam.tan_half_fov = s->camera.tan_half_fov, cam.focus_distance = s->camera.focus_distance;
    cam.lens_radius = s->camera.lens_radius, cam.projected_pixel_size = s->camera.projected_pixel_size;
    cam.ortho_scale = s->camera.ortho_scale, cam.clip_near = s->camera.clip_near, cam.clip_far = s->camera.clip_far;
}

}// namespace

extern "C" {

const char *lrhip_last_error(void) { return g_last_error.c_str(); }

int lrhip_create(int device_ordinal, lrhip_ctx **out) {
    if (out == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_create: out is NULL"); }
    int count = 0;
    LR_HIP_CHECK(hipGetDeviceCount(&count));
    if (device_ordinal < 0 || device_ordinal >= count) {
        return fail(LRHIP_ERROR_INVALID, "lrhip_create: device ordinal " + std::to_string(device_ordinal) +
                                             " out of range (" + std::to_string(count) + " HIP devices)");
    }
    LR_HIP_CHECK(hipSetDevice(device_ordinal));
    auto ctx = new lrhip_ctx{};
    ctx->device = device_ordinal;
    hipDeviceProp_t prop{};
    if (auto e = hipGetDeviceProperties(&prop, device_ordinal); e != hipSuccess) {
        delete ctx;
        return fail(LRHIP_ERROR_DEVICE, std::string{"hipGetDeviceProperties: "} + hipGetErrorString(e));
    }
    ctx->cu_count = static_cast<uint32_t>(prop.multiProcessorCount);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) {
        delete ctx;
        return fail(LRHIP_ERROR_DEVICE, "lrhip_create: failed to create stream/events");
    }
    *out = ctx;
    return LRHIP_OK;
}

void lrhip_destroy(lrhip_ctx *ctx) {
    if (ctx == nullptr) { return; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    release_scene(ctx);
    ctx->film_own.release(), ctx->converted.release(), ctx->partial.release();
    ctx->spill.release(), ctx->wf_heavy.release(), ctx->wf_cont.release(), ctx->wf_counts.release(), ctx->wf_accum.release(), ctx->pool.release(), ctx->counters.release(), ctx->work_counter.release(), ctx->scene_record.release();
    if (ctx->ev_begin) { (void)hipEventDestroy(ctx->ev_begin); }
    if (ctx->ev_end) { (void)hipEventDestroy(ctx->ev_end); }
    if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}

int lrhip_set_stream(lrhip_ctx *ctx, void *hip_stream) {
    if (ctx == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_set_stream: ctx is NULL"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); }
    if (hip_stream == nullptr) {
        LR_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    } else {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    }
    return LRHIP_OK;
}


int lrhip_update_scene(lrhip_ctx *ctx, const lr_scene *s) {
    if (ctx == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_update_scene: no scene uploaded"); }
    if (s == nullptr || s->camera.width != ctx->width || s->camera.height != ctx->height) {
        return fail(LRHIP_ERROR_INVALID, "lrhip_update_scene: the film resolution must not change");
    }
    // Only what Pipeline::update / Geometry::update move is copied again (instance matrices, the baked triangles and their
    // shading records, the refitted BVH packets, camera and environment transforms), over the SAME device buffers: a frame with
    // 256 shutter samples must not push its textures and environment tables through PCIe 256 times.  Everything else must be
    // the scene that was uploaded; the table sizes are the part of that contract that can be checked.
    if (s->accel.nodes == nullptr || s->accel.node_count != ctx->update_counts[0] || s->accel.triangle_count != ctx->update_counts[1] ||
        s->instance_count != ctx->update_counts[2] || s->triangle_count != ctx->update_counts[3] || s->texture_count != ctx->update_counts[4] ||
        s->surface_count != ctx->update_counts[5] || s->environment.kind != ctx->update_counts[6]) {
        return fail(LRHIP_ERROR_INVALID, "lrhip_update_scene: not the uploaded scene at another time (table sizes differ); use lrhip_upload_scene");
    }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    if (bvh_depth(s->accel) != ctx->bvh_depth) { return fail(LRHIP_ERROR_INVALID, "lrhip_update_scene: the BVH topology changed; use lrhip_upload_scene"); }
    auto &d = ctx->scene;
    auto copy = [&](const void *device, const void *host, size_t bytes) {// stream-ordered behind the renders that still read the old tables
        return bytes == 0u ? hipSuccess : hipMemcpyAsync(const_cast<void *>(device), host, bytes, hipMemcpyHostToDevice, ctx->stream);
    };
    auto packed = build_packed_nodes(s);
    auto instances = build_instances(s);
    std::string error;
    auto shade = build_shade_tris(s, instances, error);
    if (!error.empty()) { return fail(LRHIP_ERROR_INVALID, "lrhip_update_scene: " + error); }
    LR_HIP_CHECK(copy(d.nodes, packed.data(), packed.size() * sizeof(packed[0])));
    auto padded = build_padded_triangles(s);
    LR_HIP_CHECK(copy(d.bvh_tris, padded.data(), padded.size()));
    LR_HIP_CHECK(copy(d.instances, instances.data(), instances.size() * sizeof(instances[0])));
    LR_HIP_CHECK(copy(d.shade_tris, shade.data(), shade.size() * sizeof(shade[0])));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));// the host vectors above go out of scope
    set_camera(d, s);
    if (d.env_kind == lrd::kEnvConstant) {
        std::memcpy(d.env_to_world, s->environment.env_to_world, sizeof(d.env_to_world));
    } else if (d.env != nullptr && s->environment.kind != LR_ENV_COMBINED) {// (animated Combined environments are rejected by the host loader)
        // the record's two matrices lead the struct: overwrite them in place
        float m[18];
        std::memcpy(m, s->environment.world_to_env, sizeof(float) * 9u);
        std::memcpy(m + 9, s->environment.env_to_world, sizeof(float) * 9u);
        static_assert(offsetof(lrd::DEnvironment, world_to_env) == 0u && offsetof(lrd::DEnvironment, env_to_world) == sizeof(float) * 9u, "DEnvironment layout");
        LR_HIP_CHECK(hipMemcpy(const_cast<lrd::DEnvironment *>(d.env), m, sizeof(m), hipMemcpyHostToDevice));
    }
    return LRHIP_OK;
}

// Images whose every texel is an 8-bit code's float are kept as 8-bit texels on the device (dev_shade.h: texel_at): one 32-bit word per
// texel, appended behind the float texels of the scene (offsets in 32-bit words from the same base pointer).  A channel qualifies if all
// its texels are b * (1 / 255.f) (form 1), all are b / 255.f (form 2), or all hold one value (a padded alpha); an image qualifies if
// all four channels do and the coded ones agree on the form.  The device's decode reproduces the host's floats bit for bit: form 1 is
// the same multiplication, form 2 is checked against the division for all 256 codes first.  Returns the packed words; `textures` (the
// copy that goes to the device) gets the new offsets, the form and the constant channels.
// Offsets of packed images come out relative to the packed area.
constexpr uint64_t kByteTextureFloatBytes = 192ull << 20u;
static std::vector<uint32_t> pack_byte_textures(const lr_scene *s, std::vector<lr_texture> &textures) {
    std::vector<uint32_t> packed;
    auto division_ok = true;
    for (auto b = 0u; b < 256u; b++) { division_ok = division_ok && lrd::byte_over_255(static_cast<float>(b)) == static_cast<float>(b) / 255.f; }
    for (auto &t : textures) {
        t.pad = 0u;
        const auto count = static_cast<uint64_t>(t.width) * t.height;
        if (t.kind != LR_TEX_IMAGE || count == 0u || t.texel_offset + count > s->texel_count) { continue; }
        const auto px = s->texels + t.texel_offset * 4u;
        bool same[4], product[4], quotient[4];
        for (auto c = 0u; c < 4u; c++) {
            same[c] = true, product[c] = true, quotient[c] = division_ok;
            for (uint64_t i = 0u; i < count && (same[c] || product[c] || quotient[c]); i++) {
                const auto v = px[i * 4u + c];
                same[c] = same[c] && v == px[c];
                const auto code = v >= 0.f && v <= 1.f ? std::floor(v * 255.f + .5f) : -1.f;
                product[c] = product[c] && code >= 0.f && code * (1.f / 255.f) == v;
                quotient[c] = quotient[c] && code >= 0.f && code / 255.f == v;
            }
        }
        auto use = 0u, constant = 0u;
        for (auto f = 1u; f <= 2u && use == 0u; f++) {// the form under which every channel is either coded or one value
            auto all = true;
            auto mask = 0u;
            for (auto c = 0u; c < 4u; c++) {
                const auto coded = f == 1u ? product[c] : quotient[c];
                if (!coded && same[c]) { mask |= 1u << c; }
                all = all && (coded || same[c]);
            }
            if (all && mask != 15u) { use = f, constant = mask; }
        }
        if (use == 0u) { continue; }
        t.pad = use | (constant << 4u);
        for (auto c = 0u; c < 4u; c++) { if ((constant >> c) & 1u) { t.v[c] = px[c]; } }
        t.texel_offset = packed.size();// (relative to the packed area: lrhip_upload_scene adds its base)
        for (uint64_t i = 0u; i < count; i++) {
            auto word = 0u;
            for (auto c = 0u; c < 4u; c++) {
                const auto v = px[i * 4u + c];
                const auto code = (constant >> c) & 1u ? 0u : static_cast<uint32_t>(std::floor(v * 255.f + .5f));
                word |= (code & 255u) << (8u * c);
            }
            packed.push_back(word);
        }
    }
    return packed;
}

// Every index one table holds into another, checked once: the caller may be a third party, and nothing may read out of bounds
// on either side of the boundary (lrhip.h: "nothing throws or aborts across the boundary").
static std::string validate_indices(const lr_scene *s) {
    auto tex_ok = [&](int32_t id) { return id < 0 || static_cast<uint32_t>(id) < s->texture_count; };
    for (uint32_t i = 0; i < s->texture_count; i++) {
        auto &t = s->textures[i];
        if (t.kind == LR_TEX_CHECKERBOARD && (!tex_ok(t.child[0]) || !tex_ok(t.child[1]))) { return "texture " + std::to_string(i) + ": child texture out of range"; }
    }
    for (uint32_t i = 0; i < s->surface_count; i++) {
        auto &f = s->surfaces[i];
        for (auto t : f.tex) { if (!tex_ok(t)) { return "surface " + std::to_string(i) + ": texture id out of range"; } }
        if (!tex_ok(f.normal_tex) || !tex_ok(f.alpha_tex)) { return "surface " + std::to_string(i) + ": normal / alpha texture id out of range"; }
        if ((f.kind == LR_SURFACE_MIX || f.kind == LR_SURFACE_LAYERED) && (f.u[0] >= s->surface_count || f.u[1] >= s->surface_count)) {
            return "surface " + std::to_string(i) + ": child surface out of range";
        }
    }
    for (uint32_t i = 0; i < s->light_count; i++) {
        auto e = s->lights[i].emission_tex;
        if (e < 0 || static_cast<uint32_t>(e) >= s->texture_count) { return "light " + std::to_string(i) + ": emission texture out of range"; }
    }
    for (uint32_t i = 0; i < s->instance_count; i++) {
        auto &h = s->instances[i].handle;
        if ((h.x >> 10u) >= s->mesh_count) { return "instance " + std::to_string(i) + ": mesh index out of range"; }
        auto flags = h.x & 1023u;
        if ((flags & LR_SHAPE_HAS_SURFACE) && ((h.y >> 12u) & 4095u) >= s->surface_count) { return "instance " + std::to_string(i) + ": surface tag out of range"; }
        if ((flags & LR_SHAPE_HAS_LIGHT) && (h.y & 4095u) >= s->light_count) { return "instance " + std::to_string(i) + ": light tag out of range"; }
    }
    for (uint32_t i = 0; i < s->light_instance_count; i++) {
        if (s->light_instances[i].instance_id >= s->instance_count) { return "light instance " + std::to_string(i) + ": instance id out of range"; }
    }
    if (s->environment.kind == LR_ENV_COMBINED) {// a tree of Combined nodes over Spherical / Directional leaves (lr_scene.h: children before parents)
        std::vector<uint32_t> depth(s->environment_child_count, 0u);// Combined nodes from the record down, itself included
        auto check_node = [&](const lr_environment &c, uint32_t limit, uint32_t &d) -> std::string {
            d = 0u;
            if (c.kind != LR_ENV_COMBINED) { return {}; }
            for (auto k = 0; k < 2; k++) {
                if (c.child[k] >= limit) { return "child index out of range (children precede their parents in environment_children)"; }
                if (!(c.child_scale[k] > 0.f)) { return "child scales must be positive (a Combined node with one live child is flattened by the host)"; }
                d = std::max(d, depth[c.child[k]]);
            }
            d += 1u;
            return {};
        };
        for (uint32_t i = 0; i < s->environment_child_count; i++) {
            auto &c = s->environment_children[i];
            if (c.kind != LR_ENV_SPHERICAL && c.kind != LR_ENV_DIRECTIONAL && c.kind != LR_ENV_COMBINED) { return "environment child " + std::to_string(i) + ": invalid kind"; }
            if (c.kind != LR_ENV_COMBINED && (c.emission_tex < 0 || static_cast<uint32_t>(c.emission_tex) >= s->texture_count)) {
                return "environment child " + std::to_string(i) + ": emission texture out of range";
            }
            if (auto bad = check_node(c, i, depth[i]); !bad.empty()) { return "environment child " + std::to_string(i) + ": " + bad; }
        }
        uint32_t root_depth = 0u;
        if (auto bad = check_node(s->environment, s->environment_child_count, root_depth); !bad.empty()) { return "environment: " + bad; }
        if (root_depth > static_cast<uint32_t>(LR_ENV_MAX_COMBINED_DEPTH)) { return "environment: Combined nodes nested deeper than LR_ENV_MAX_COMBINED_DEPTH"; }
    }
    return {};
}

int lrhip_upload_scene(lrhip_ctx *ctx, const lr_scene *s) {
    if (ctx == nullptr || s == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: NULL argument"); }
    if (s->accel.nodes == nullptr || s->accel.node_count == 0u) {
        return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: scene->accel is not built (lrhost_scene_build_accel)");
    }
    if (s->environment.kind > LR_ENV_COMBINED || (s->environment.kind == LR_ENV_COMBINED && (s->environment_child_count < 2u || s->environment_children == nullptr))) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: environment kind not supported");
    }
    if (auto bad = validate_indices(s); !bad.empty()) { return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: " + bad); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    release_scene(ctx);
    ctx->bvh_depth = bvh_depth(s->accel);
    if (ctx->bvh_depth == 0u) { return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: the BVH must have one-triangle leaves (lrhost_scene_build_accel builds them)"); }
    ctx->features = s->any_non_opaque != 0u ? lrd::kFeatAlpha : 0u;
    ctx->env_tree = false;
    // a leaf names its triangle in 27 bits (the sentinel of the empty slots is one more); the traversal loop addresses packets and leaf
    // triangles by 32-bit byte offsets from their table bases (64 B x 2^26 packets, 48 B x 89 478 484 triangles = 4 GiB)
    constexpr uint32_t kMaxBvhTriangles = static_cast<uint32_t>((1ull << 32u) / sizeof(lr_bvh_triangle)) - 2u;
    static_assert(sizeof(lr_bvh_triangle) == 48u && kMaxBvhTriangles < lrd::kLeafIndexMask, "dev_trace.h: trav_leaf_fetch multiplies the leaf's triangle index by 48 in 32 bits");
    if (s->accel.triangle_count >= kMaxBvhTriangles || s->accel.node_count >= (1u << 26u)) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: more than 89 478 482 BVH triangles or 2^26 - 1 BVH packets");
    }
    // a walk pushes at most three entries per level; a pool kernel parks five more words on top of a lane's stack across the shading block
    // (megapool_kernel.h).  From the constants of THIS build: the `make shallow` library keeps 4 entries in LDS, not 16
    if (ctx->bvh_depth * 3u + lrd::kPoolParkedWords > lrd::kStackLds + lrd::kSpillEntries) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: BVH depth " + std::to_string(ctx->bvh_depth) +
                                                 " exceeds the traversal stack capacity");
    }
    auto &d = ctx->scene;
    d = lrd::DScene{};
    int rc;
#define LR_UP(call)                                   \
    if ((rc = (call)) != LRHIP_OK) {                  \
        release_scene(ctx);                           \
        return rc;                                    \
    }
    {
        auto packed = build_packed_nodes(s);
        LR_UP(upload(ctx, packed.data(), packed.size(), &d.nodes));
    }
    {
        auto padded = build_padded_triangles(s);
        const uint8_t *dev = nullptr;
        LR_UP(upload(ctx, padded.data(), padded.size(), &dev));
        d.bvh_tris = reinterpret_cast<const lr_bvh_triangle *>(dev);
    }
    LR_UP(upload(ctx, s->vertices, s->vertex_count, &d.vertices));
    LR_UP(upload(ctx, s->triangles, s->triangle_count, &d.triangles));
    LR_UP(upload(ctx, s->tri_alias, s->triangle_count, &d.tri_alias));
    LR_UP(upload(ctx, s->tri_pdf, s->triangle_count, &d.tri_pdf));
    LR_UP(upload(ctx, s->light_instances, s->light_instance_count, &d.light_instances));
    {// textures: 8-bit images are uploaded as 8-bit texels (pack_byte_textures above; dev_shade.h: texel_at), behind the float texels
        std::vector<lr_texture> textures(s->textures, s->textures + s->texture_count);
        // ... where the float texels are more than the caches hold: the decode costs a few instructions per texel, which a scene whose
        // images sit in the L2 / Infinity Cache anyway does not get back (kitchen class, 36 MB of float texels: -1.5 % packed; camera
        // class, 512 MB: +4 %; the same frame at 128 MB of float texels: +-0 -- profiles/r05zd_byte_textures.txt, r05zc_byte_textures_always.txt, r05zb_c4_texture_size.txt)
        auto image_texels = static_cast<uint64_t>(0u);
        for (auto &t : textures) { if (t.kind == LR_TEX_IMAGE) { image_texels += static_cast<uint64_t>(t.width) * t.height; } }
        // ... and where the kernels the scene renders on decode them (round 6: the decode is compiled into the lean kernels of the kFeatByteTex
        // bit only, which exist for the Disney feature sets of both schedulers -- dev_wavefront.h): a scene with alpha-tested surfaces, Mix or
        // Layered surfaces (wavefront mode: lean passes without the decode) or nested Combined environments keeps float texels.  The sibling
        // integrators and the volumetric kernel run on variants that always decode.
        auto lean_decodes = s->any_non_opaque == 0u;
        for (uint32_t i = 0; i < s->surface_count && lean_decodes; i++) { lean_decodes = s->surfaces[i].kind != LR_SURFACE_MIX && s->surfaces[i].kind != LR_SURFACE_LAYERED; }
        if (s->environment.kind == LR_ENV_COMBINED) {
            for (uint32_t i = 0; i < s->environment_child_count; i++) { lean_decodes = lean_decodes && s->environment_children[i].kind != LR_ENV_COMBINED; }
        }
        const auto decodes = lean_decodes || s->integrator.kind != LR_INTEGRATOR_MEGAPATH;
        if (ctx->byte_textures == 2u && !decodes) {
            release_scene(ctx);
            return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: texture storage mode 2 (always 8-bit texels) on a scene whose kernels do not decode them "
                                                 "(alpha-tested, Mix or Layered surfaces, nested Combined environments under MegaPath)");
        }
        const auto pack = decodes && (ctx->byte_textures == 2u || (ctx->byte_textures == 1u && image_texels * 16u > kByteTextureFloatBytes));
        for (auto &t : textures) { t.pad = 0u; }
        const auto packed = pack ? pack_byte_textures(s, textures) : std::vector<uint32_t>{};
        // The float texels of an image that is held packed do NOT go to the device as well (the camera class kept 512 MB of dead
        // floats beside its 128 MB of codes): what is uploaded is the union of the texel ranges the float-kept images name, closed up, with
        // their offsets rebased (ranges that overlap -- two records over one image -- stay one range).
        std::vector<std::pair<uint64_t, uint64_t>> ranges;// [begin, end) in texels, of the images that stay float
        for (auto &t : textures) {
            const auto count = static_cast<uint64_t>(t.width) * t.height;
            if (t.kind == LR_TEX_IMAGE && t.pad == 0u && count != 0u && t.texel_offset + count <= s->texel_count) { ranges.emplace_back(t.texel_offset, t.texel_offset + count); }
        }
        std::sort(ranges.begin(), ranges.end());
        std::vector<std::array<uint64_t, 3>> kept;// merged ranges: begin, end, where it starts on the device
        uint64_t kept_texels = 0u;
        for (auto &r : ranges) {
            if (!kept.empty() && r.first <= kept.back()[1]) {
                if (r.second > kept.back()[1]) { kept_texels += r.second - kept.back()[1], kept.back()[1] = r.second; }
            } else {
                kept.push_back({r.first, r.second, kept_texels}), kept_texels += r.second - r.first;
            }
        }
        if (packed.empty()) { kept.assign(1u, {0u, s->texel_count, 0u}), kept_texels = s->texel_count; }// (nothing packed: the table as the host made it)
        for (auto &t : textures) {
            if (t.kind != LR_TEX_IMAGE) { continue; }
            if (t.pad != 0u) { t.texel_offset += kept_texels * 4u; continue; }// packed: 32-bit words from the same base, behind the floats
            for (auto &k : kept) {
                if (t.texel_offset >= k[0] && t.texel_offset < k[1]) { t.texel_offset = t.texel_offset - k[0] + k[2]; break; }
            }
        }
        LR_UP(upload(ctx, textures.data(), textures.size(), &d.textures));
        DeviceBuffer b;
        const auto float_bytes = static_cast<size_t>(kept_texels) * 4u * sizeof(float);
        b.bytes = std::max<size_t>(float_bytes + packed.size() * sizeof(uint32_t), 16u);
        LR_HIP_CHECK(hipMalloc(&b.ptr, b.bytes));
        ctx->scene_buffers.emplace_back(b);
        for (auto &k : kept) {
            if (k[1] > k[0]) { LR_HIP_CHECK(hipMemcpy(static_cast<char *>(b.ptr) + k[2] * 16u, s->texels + k[0] * 4u, (k[1] - k[0]) * 16u, hipMemcpyHostToDevice)); }
        }
        if (!packed.empty()) { LR_HIP_CHECK(hipMemcpy(static_cast<char *>(b.ptr) + float_bytes, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice)); }
        d.texels = static_cast<const float *>(b.ptr);
        ctx->packed_texel_words = packed.size();
        ctx->texel_bytes = b.bytes;
    }
    LR_UP(upload(ctx, &s->filter, 1u, &d.filter));
    auto instances = build_instances(s);
    LR_UP(upload(ctx, instances.data(), instances.size(), &d.instances));
    {
        std::string error;
        auto shade = build_shade_tris(s, instances, error);
        if (!error.empty()) { release_scene(ctx); return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: " + error); }
        LR_UP(upload(ctx, shade.data(), shade.size(), &d.shade_tris));
    }
    // closures: fold constant textures on the host (same arithmetic as the per-hit device path)
    auto is_constant = [&](int32_t id) { return id < 0 || s->textures[id].kind == LR_TEX_CONSTANT; };
    std::vector<lrd::DClosure> closures(s->surface_count);
    // the surfaces as the device reads them for dynamic closures: the host's record + every texture slot's value where its texture is
    // constant + which slots need a lookup per hit (dev_scene.h: DSurface)
    std::vector<lrd::DSurface> surfaces(s->surface_count);
    for (uint32_t i = 0; i < s->surface_count; i++) {
        auto &rec = surfaces[i];
        std::memset(&rec, 0, sizeof(rec));
        rec.raw = s->surfaces[i];
        for (auto slot = 0u; slot < lrd::kSurfaceSlots; slot++) {
            const auto id = rec.raw.tex[slot];
            if (id < 0) { continue; }
            auto &t = s->textures[id];
            if (t.kind == LR_TEX_CONSTANT) { std::memcpy(rec.value[slot], t.v, sizeof(t.v)); }
            else { rec.dynamic_mask |= 1u << slot; }
            rec.channels[slot >> 3u] |= (t.channels & 15u) << ((slot & 7u) * 4u);
        }
        rec.first_lookup = rec.raw.normal_tex >= 0 ? rec.raw.normal_tex : (rec.dynamic_mask != 0u ? rec.raw.tex[__builtin_ctz(rec.dynamic_mask)] : -1);
    }
    LR_UP(upload(ctx, surfaces.data(), surfaces.size(), &d.surfaces));
    for (uint32_t i = 0; i < s->surface_count; i++) {
        auto &surf = s->surfaces[i];
        if (surf.kind == LR_SURFACE_DISNEY) { ctx->features |= lrd::kFeatDisney; }
        if (surf.kind == LR_SURFACE_MIX) { ctx->features |= lrd::kFeatMix; }
        if (surf.kind == LR_SURFACE_LAYERED) { ctx->features |= lrd::kFeatLayered | lrd::kFeatDisney; }
        if (surf.kind == LR_SURFACE_MIX || surf.kind == LR_SURFACE_LAYERED) {// composition of the two (validate_indices checked the child indices)
            // The interpreters of dev_heavy.h assume what the C++ loader enforces (scene.cpp: depth_of, surface_layered_levels); a
            // C-ABI caller gets the same answer here instead of silently wrong shading: at most LR_LAYERED_MAX_LEVELS Layered surfaces on
            // a path through the interfaces, Mix trees at most kMixMaxDepth levels deep with u[2] = that depth, no cycles.
            std::string bad;
            // (memoised per (surface, Layered levels above it): subtrees shared between parents -- a DAG from a C-ABI caller -- are walked
            // once, not once per path to them; a surface met again while it is still being walked is a cycle)
            std::map<std::pair<uint32_t, uint32_t>, int> memo;
            std::set<std::pair<uint32_t, uint32_t>> walking;
            std::function<int(uint32_t, uint32_t, uint32_t)> depth_of_inner;
            std::function<int(uint32_t, uint32_t, uint32_t)> depth_of = [&](uint32_t tag, uint32_t budget, uint32_t layered_above) -> int {
                const auto key = std::make_pair(tag, layered_above);
                if (auto it = memo.find(key); it != memo.end()) { return it->second; }
                if (!walking.insert(key).second) { bad = "a Mix / Layered tree that is cyclic or deeper than the interpreter reaches"; return 0; }
                const auto d = depth_of_inner(tag, budget, layered_above);
                walking.erase(key);
                if (bad.empty()) { memo.emplace(key, d); }
                return d;
            };
            depth_of_inner = [&](uint32_t tag, uint32_t budget, uint32_t layered_above) -> int {
                auto &c = s->surfaces[tag];
                if (budget == 0u) { bad = "a Mix / Layered tree that is cyclic or deeper than the interpreter reaches"; return 0; }
                if (c.kind == LR_SURFACE_LAYERED) {
                    if (layered_above >= static_cast<uint32_t>(LR_LAYERED_MAX_LEVELS)) {
                        bad = "Layered surfaces nested more than " + std::to_string(LR_LAYERED_MAX_LEVELS) + " levels deep are not supported";
                        return 0;
                    }
                    depth_of(c.u[0], budget - 1u, layered_above + 1u), depth_of(c.u[1], budget - 1u, layered_above + 1u);
                    return 1;// a leaf of the tree that holds it (its own interfaces count from zero)
                }
                if (c.kind != LR_SURFACE_MIX) { return 0; }
                auto depth = std::max(depth_of(c.u[0], budget - 1u, layered_above), depth_of(c.u[1], budget - 1u, layered_above));
                if (bad.empty() && static_cast<int>(c.u[2]) != depth) { bad = "a Mix surface whose u[2] is not the depth of its tree"; }
                if (bad.empty() && depth > lrd::kMixMaxDepth) { bad = "Mix surfaces nested more than " + std::to_string(lrd::kMixMaxDepth) + " levels deep are not supported"; }
                return 1 + depth;
            };
            depth_of(i, (static_cast<uint32_t>(lrd::kMixMaxDepth) + 2u) * (static_cast<uint32_t>(LR_LAYERED_MAX_LEVELS) + 1u) + 2u, 0u);
            if (!bad.empty()) {
                release_scene(ctx);
                return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: surface " + std::to_string(i) + ": " + bad);
            }
            for (auto k = 0u; k < 2u; k++) {
                auto child = s->surfaces[surf.u[k]].kind;
                if ((surf.kind == LR_SURFACE_MIX && child == LR_SURFACE_LAYERED) || (surf.kind == LR_SURFACE_LAYERED && (child == LR_SURFACE_MIX || child == LR_SURFACE_LAYERED))) {
                    ctx->features |= lrd::kFeatNest;
                }
            }
        }
        auto dynamic = surf.normal_tex >= 0;
        for (auto t : surf.tex) { dynamic = dynamic || !is_constant(t); }
        lrd::DClosure c{};
        if (!dynamic) {
            c = lrd::resolve_closure(
                surf,
                [&](int slot) {
                    auto &t = s->textures[surf.tex[slot]];
                    return make_float4(t.v[0], t.v[1], t.v[2], t.v[3]);
                },
                [&](int slot) { return s->textures[surf.tex[slot]].channels; }, 1.f);
        } else {
            c.kind = surf.kind;
            c.x[0] = surf.u[0], c.x[1] = surf.u[1], c.x[2] = surf.u[2], c.x[3] = surf.u[3];// children / masks are needed before resolution
        }
        c.dynamic = dynamic ? 1u : 0u;
        closures[i] = c;
    }
    LR_UP(upload(ctx, closures.data(), closures.size(), &d.closures));
    std::vector<lrd::DLight> lights(s->light_count);
    for (uint32_t i = 0; i < s->light_count; i++) {
        auto &src = s->lights[i];
        lrd::DLight l{};
        l.emission_tex = src.emission_tex, l.scale = src.scale, l.two_sided = src.two_sided;
        auto &t = s->textures[src.emission_tex];
        l.dynamic = t.kind == LR_TEX_CONSTANT ? 0u : 1u;
        if (!l.dynamic) {// evaluate_illuminant_spectrum of a static texture, texture.cpp:49-53,77-79
            auto rgb = lrd::extend_rgb(make_float4(t.v[0], t.v[1], t.v[2], t.v[3]), t.channels);
            auto sv = lrd::max0(rgb) * src.scale;
            l.L[0] = sv.x, l.L[1] = sv.y, l.L[2] = sv.z;
        }
        lights[i] = l;
    }
    LR_UP(upload(ctx, lights.data(), lights.size(), &d.lights));
#undef LR_UP
    set_camera(d, s);
    d.env_kind = lrd::kEnvNone;
    d.env = nullptr;
    if (s->environment.kind != LR_ENV_NONE) {
        auto &e = s->environment;
        if (e.kind != LR_ENV_COMBINED && (e.emission_tex < 0 || static_cast<uint32_t>(e.emission_tex) >= s->texture_count)) {
            release_scene(ctx);
            return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: environment without an emission texture");
        }
        auto &t = s->textures[e.kind == LR_ENV_COMBINED ? 0 : e.emission_tex];
        if (e.kind == LR_ENV_SPHERICAL && t.kind == LR_TEX_CONSTANT) {
            d.env_kind = lrd::kEnvConstant;
            auto sv = lrd::max0(lrd::extend_rgb(make_float4(t.v[0], t.v[1], t.v[2], t.v[3]), t.channels)) * e.scale;
            d.env_L[0] = sv.x, d.env_L[1] = sv.y, d.env_L[2] = sv.z;
            std::memcpy(d.env_to_world, e.env_to_world, sizeof(d.env_to_world));
        } else {
            // one DEnvironment per record; a Combined root points at its two children
            auto make_record = [&](const lr_environment &r, lrd::DEnvironment &de) -> int {
                auto &rt = s->textures[r.emission_tex];
                de = lrd::DEnvironment{};
                std::memcpy(de.world_to_env, r.world_to_env, sizeof(de.world_to_env));
                std::memcpy(de.env_to_world, r.env_to_world, sizeof(de.env_to_world));
                de.emission_tex = r.emission_tex, de.scale = r.scale;
                de.constant_emission = rt.kind == LR_TEX_CONSTANT ? 1u : 0u;
                std::memcpy(de.direction, r.direction, sizeof(de.direction));
                de.cos_half_angle = r.cos_half_angle, de.visible = r.visible;
                if (r.kind == LR_ENV_DIRECTIONAL) { de.kind = lrd::kEnvDirectional; return LRHIP_OK; }
                if (r.kind != LR_ENV_SPHERICAL) { return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: invalid environment record"); }
                if (rt.kind == LR_TEX_CONSTANT) { de.kind = lrd::kEnvConstant; return LRHIP_OK; }
                if (r.alias == nullptr || r.pdf == nullptr || r.map_width == 0u || r.map_height == 0u) {
                    return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: image-based Spherical environment without importance tables");
                }
                de.kind = lrd::kEnvImage;
                de.map_width = r.map_width, de.map_height = r.map_height;
                auto texels = static_cast<size_t>(r.map_width) * r.map_height;
                if (auto rc2 = upload(ctx, r.alias, texels + r.map_height, &de.alias); rc2 != LRHIP_OK) { return rc2; }
                return upload(ctx, r.pdf, texels, &de.pdf);
            };
            ctx->features |= lrd::kFeatEnv;
            // a Combined node points at its two children (uploaded before it: lr_scene.h orders children before parents)
            std::vector<const lrd::DEnvironment *> uploaded(e.kind == LR_ENV_COMBINED ? s->environment_child_count : 0u, nullptr);
            auto make_node = [&](const lr_environment &r, lrd::DEnvironment &de) -> int {
                if (r.kind != LR_ENV_COMBINED) { return make_record(r, de); }
                de = lrd::DEnvironment{};
                std::memcpy(de.world_to_env, r.world_to_env, sizeof(de.world_to_env));
                std::memcpy(de.env_to_world, r.env_to_world, sizeof(de.env_to_world));
                de.kind = lrd::kEnvCombined;
                for (auto k = 0; k < 2; k++) {
                    de.child_scale[k] = r.child_scale[k], de.child[k] = uploaded[r.child[k]];
                    if (s->environment_children[r.child[k]].kind == LR_ENV_COMBINED) { de.tree = 1u; }
                }
                return LRHIP_OK;
            };
            lrd::DEnvironment root{};
            int rc2 = LRHIP_OK;
            for (uint32_t i = 0; i < uploaded.size() && rc2 == LRHIP_OK; i++) {
                lrd::DEnvironment child{};
                if ((rc2 = make_node(s->environment_children[i], child)) == LRHIP_OK) { rc2 = upload(ctx, &child, 1u, &uploaded[i]); }
            }
            if (rc2 == LRHIP_OK) { rc2 = make_node(e, root); }
            // a root over nested Combined nodes is walked by out-of-line code that only the call-making variants hold (dev_shade.h:
            // env_evaluate_tree); lrhip_render picks one of those
            ctx->env_tree = root.tree != 0u;
            if (rc2 == LRHIP_OK) { rc2 = upload(ctx, &root, 1u, &d.env); }
            if (rc2 != LRHIP_OK) {
                release_scene(ctx);
                return rc2;
            }
            d.env_kind = root.kind;
        }
    }
    d.max_depth = s->integrator.max_depth, d.rr_depth = s->integrator.rr_depth;
    d.integrator_kind = s->integrator.kind, d.integrator_flags = s->integrator.flags;
    if (s->integrator.kind > LR_INTEGRATOR_VPT_NAIVE) { release_scene(ctx); return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: unknown integrator kind"); }
    d.env_medium_tag = s->integrator.environment_medium_tag;
    const auto nested = (ctx->features & lrd::kFeatNest) != 0u;
    if (s->integrator.kind == LR_INTEGRATOR_VPT_NAIVE) {// the volumetric megakernel is one kernel with everything in it
        for (uint32_t i = 0; i < s->instance_count; i++) {
            if ((s->instances[i].handle.x & LR_SHAPE_HAS_MEDIUM) && (s->instances[i].handle.y >> 24u) >= s->medium_count) {
                release_scene(ctx);
                return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: an instance references a medium that is not in scene->media");
            }
        }
        if (d.env_medium_tag != LR_INVALID_ID && d.env_medium_tag >= s->medium_count) { release_scene(ctx); return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: invalid environment medium tag"); }
        if (auto r = upload(ctx, s->media, s->medium_count, &d.media); r != LRHIP_OK) { release_scene(ctx); return r; }
        ctx->features = lrd::kFeatVpt;
        if (nested) { release_scene(ctx); return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: Mix / Layered surfaces nested in each other are supported by the MegaPath integrator only"); }
    } else if (s->integrator.kind != LR_INTEGRATOR_MEGAPATH) {
        if (nested) { release_scene(ctx); return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_upload_scene: Mix / Layered surfaces nested in each other are supported by the MegaPath integrator only"); }
        // the sibling integrators (Direct / Normal) live in the all-features variant only
        ctx->features |= lrd::kFeatSceneMask | lrd::kFeatAux;
    }
    d.rr_threshold = s->integrator.rr_threshold, d.env_prob = s->integrator.env_prob;
    d.light_count = s->integrator.light_count;
    d.has_lights = s->light_count != 0u ? 1u : 0u;
    d.sampler_kind = s->sampler.kind, d.seed = s->sampler.seed;
    d.sampler_spp = s->sampler.spp, d.sobol_scale = s->sampler.scale;
    if (s->sampler.tile_size[0] > 0xffffu || s->sampler.tile_size[1] > 0xffffu || (s->sampler.tile_size[0] != 0u) != (s->sampler.tile_size[1] != 0u)) {
        release_scene(ctx);
        return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: invalid sampler tile size");
    }
    d.sampler_tile = s->sampler.tile_size[0] | (s->sampler.tile_size[1] << 16u), d.sampler_tile_jitter = s->sampler.tile_jitter;
    if (s->camera.width == 0u || s->camera.height == 0u || s->camera.width > 0xffffu || s->camera.height > 0xffffu) {// (the generic samplers keep a pixel as x | y << 16)
        release_scene(ctx);
        return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: the film must be 1 .. 65535 pixels wide and high");
    }
    if (d.sampler_kind == LR_SAMPLER_SOBOL || d.sampler_kind == LR_SAMPLER_PADDED_SOBOL) {
        if (s->sampler.sobol_matrices == nullptr || (d.sampler_kind == LR_SAMPLER_SOBOL && d.sobol_scale > 1u && (s->sampler.vdc_sobol == nullptr || s->sampler.vdc_sobol_inv == nullptr))) {
            release_scene(ctx);
            return fail(LRHIP_ERROR_INVALID, "lrhip_upload_scene: Sobol sampler without its tables");
        }
        int rc2;
        if ((rc2 = upload(ctx, s->sampler.sobol_matrices, static_cast<size_t>(LR_SOBOL_DIMENSIONS) * LR_SOBOL_MATRIX_SIZE, &d.sobol_matrices)) != LRHIP_OK ||
            (s->sampler.vdc_sobol != nullptr &&
             ((rc2 = upload(ctx, s->sampler.vdc_sobol, static_cast<size_t>(LR_SOBOL_MATRIX_SIZE), &d.vdc_sobol)) != LRHIP_OK ||
              (rc2 = upload(ctx, s->sampler.vdc_sobol_inv, static_cast<size_t>(LR_SOBOL_MATRIX_SIZE), &d.vdc_sobol_inv)) != LRHIP_OK))) {
            release_scene(ctx);
            return rc2;
        }
        if (d.sampler_kind == LR_SAMPLER_SOBOL) {
            // the global Sobol sampler multiplies a 52-bit index with the generator matrix of the draw's dimension (sobol.cpp:52-60: one
            // table word per set bit).  The product is linear over GF(2), so it is the XOR of one precomputed word per index BYTE:
            // [dimension][byte][256] words, 7.3 MB, built here from the same matrices (bit-identical results; dev_shade.h: sobol_bits)
            constexpr size_t kBytes = (LR_SOBOL_MATRIX_SIZE + 7) / 8;
            std::vector<uint32_t> table(static_cast<size_t>(LR_SOBOL_DIMENSIONS) * kBytes * 256u, 0u);
            for (size_t dim = 0; dim < LR_SOBOL_DIMENSIONS; dim++) {
                auto m = s->sampler.sobol_matrices + dim * LR_SOBOL_MATRIX_SIZE;
                for (size_t k = 0; k < kBytes; k++) {
                    auto t = table.data() + (dim * kBytes + k) * 256u;
                    for (uint32_t b = 1u; b < 256u; b++) {
                        auto bit = 8u * static_cast<uint32_t>(k) + static_cast<uint32_t>(__builtin_ctz(b));
                        t[b] = t[b & (b - 1u)] ^ (bit < static_cast<uint32_t>(LR_SOBOL_MATRIX_SIZE) ? m[bit] : 0u);
                    }
                }
            }
            if ((rc2 = upload(ctx, table.data(), table.size(), &d.sobol_bytes)) != LRHIP_OK) {
                release_scene(ctx);
                return rc2;
            }
            // the same for the two 64-bit products that turn (pixel, sample number) into the sample's index in the global sequence
            // (_sobol_interval_to_index, sobol.cpp:67-96): [byte][256] each
            d.vdc_bytes = nullptr, d.vdc_inv_bytes = nullptr;
            if (s->sampler.vdc_sobol != nullptr) {
                auto bytewise = [&](const uint64_t *rows) {
                    std::vector<uint64_t> t(kBytes * 256u, 0ull);
                    for (size_t k = 0; k < kBytes; k++) {
                        for (uint32_t b = 1u; b < 256u; b++) {
                            auto bit = 8u * static_cast<uint32_t>(k) + static_cast<uint32_t>(__builtin_ctz(b));
                            t[k * 256u + b] = t[k * 256u + (b & (b - 1u))] ^ (bit < static_cast<uint32_t>(LR_SOBOL_MATRIX_SIZE) ? rows[bit] : 0ull);
                        }
                    }
                    return t;
                };
                auto vdc = bytewise(s->sampler.vdc_sobol), inv = bytewise(s->sampler.vdc_sobol_inv);
                if ((rc2 = upload(ctx, vdc.data(), vdc.size(), &d.vdc_bytes)) != LRHIP_OK || (rc2 = upload(ctx, inv.data(), inv.size(), &d.vdc_inv_bytes)) != LRHIP_OK) {
                    release_scene(ctx);
                    return rc2;
                }
            }
        }
    }
    d.film_clamp = s->film.clamp;
    for (auto i = 0; i < 3; i++) { ctx->film_scale[i] = s->film.scale[i]; }
    ctx->width = s->camera.width, ctx->height = s->camera.height;
    auto film_bytes = static_cast<size_t>(ctx->width) * ctx->height * sizeof(float4);
    if (auto r = ensure(ctx->film_own, film_bytes); r != LRHIP_OK) { return r; }
    if (auto r = ensure(ctx->converted, film_bytes); r != LRHIP_OK) { return r; }
    if (auto r = ensure(ctx->counters, sizeof(lrd::DCounters)); r != LRHIP_OK) { return r; }
    if (auto r = ensure(ctx->work_counter, 1024u); r != LRHIP_OK) { return r; }
    LR_HIP_CHECK(hipMemset(ctx->counters.ptr, 0, sizeof(lrd::DCounters)));
    // a caller-owned film (lrhip_bind_film) stays bound across uploads of the same resolution (MegaPathRenderer.render_frame
    // uploads per shutter sample); a resolution change unbinds it -- its size is no longer the frame's
    if (ctx->film_external != nullptr && ctx->film_external_w == ctx->width && ctx->film_external_h == ctx->height) {
        ctx->film = ctx->film_external;
    } else {
        ctx->film_external = nullptr;
        ctx->film = static_cast<float4 *>(ctx->film_own.ptr);
    }
    LR_HIP_CHECK(hipMemset(ctx->film, 0, film_bytes));
    // persistent grid: as many blocks as are resident, asked per variant at its first launch (lrhip_render); the
    // traversal-stack overflow area is sized for the densest variant
    for (auto &b : ctx->variant_blocks) { b = -1; }
    for (auto &b : ctx->padded_blocks) { b = -1; }
    for (auto &b : ctx->heavy_blocks) { b = -1; }
    ctx->grid_blocks = ctx->cu_count * kMaxBlocksPerCu;
    auto total_threads = static_cast<size_t>(ctx->grid_blocks) * lrd::kBlockThreads;
    if (auto r = ensure(ctx->spill, total_threads * lrd::kSpillEntries * sizeof(uint32_t)); r != LRHIP_OK) { return r; }
    ctx->update_counts[0] = s->accel.node_count, ctx->update_counts[1] = s->accel.triangle_count, ctx->update_counts[2] = s->instance_count;
    ctx->update_counts[3] = static_cast<uint32_t>(s->triangle_count), ctx->update_counts[4] = s->texture_count, ctx->update_counts[5] = s->surface_count;
    ctx->update_counts[6] = s->environment.kind;
    ctx->scene_ready = true;
    return LRHIP_OK;
}

int lrhip_bind_film(lrhip_ctx *ctx, void *device_float4_film) {
    if (ctx == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_bind_film: no scene uploaded"); }
    ctx->film_external = static_cast<float4 *>(device_float4_film);
    ctx->film_external_w = ctx->width, ctx->film_external_h = ctx->height;
    ctx->film = device_float4_film != nullptr ? static_cast<float4 *>(device_float4_film) : static_cast<float4 *>(ctx->film_own.ptr);
    return LRHIP_OK;
}

int lrhip_film_clear(lrhip_ctx *ctx) {
    if (ctx == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_film_clear: no scene uploaded"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    LR_HIP_CHECK(hipMemsetAsync(ctx->film, 0, static_cast<size_t>(ctx->width) * ctx->height * sizeof(float4), ctx->stream));
    return LRHIP_OK;
}

// Work items of a launch over `spp` samples per pixel of a shard with `shard_tiles` tiles (dev_scene.h: RenderArgs / item_range).
// Two losses are balanced.  The drain at the end of every item -- its last paths finish with most lanes idle -- is a share of
// ~a / S of an item of S samples per pixel; the tail of the launch -- waves out of items while the last ones finish -- is
// ~S_last * waves / (2 * spp * tiles).  With uniform items both depend on the same S and the optimum is S = sqrt(2a * spp * tiles /
// waves) (a = 0.625 from sweeps, rounds 1-2).  Round 3: the items TAPER -- the first ~90 % of the samples go out in items 2.5x that
// size, the rest in items a third of it, and all big items are handed out before the first small one: the bulk drains rarely, the
// end of the launch is made of short items.  C2 at 1024 spp, full frame / 1-of-8 shard: uniform 1213.5 / 164.5 ms, tapered 1197.5 /
// 158.0 ms (kernel-level strong-scaling efficiency at 8 shards 0.922 -> 0.947); big factor 2 / 2.5 / 3.5 / 4, small divisor 2 / 3 /
// 4 / 5 and fractions 0.75 / 0.85 / 0.9 were swept (profiles/archive/r03p_taper_sweep.txt).  The partition is a function of (spp, shard_tiles, scale) only -- never of the device
// or of the tile range of the call -- so films stay bit-identical under any sharding with the same balance_shards.
#ifndef LR_TAPER_BIG
#define LR_TAPER_BIG 2.5
#endif
#ifndef LR_TAPER_SMALL
#define LR_TAPER_SMALL 3.0
#endif
#ifndef LR_TAPER_FRACTION
#define LR_TAPER_FRACTION 0.9
#endif
struct Chunking {
    uint32_t count, big_count, big, small;
};
static Chunking chunking_of(uint32_t spp, double shard_tiles, double item_scale, bool taper) {
    const auto s_opt = std::max(1.0, std::sqrt(item_scale * spp * shard_tiles / kNominalWaves));
    Chunking c{};
    if (taper && spp >= 16u) {
        // (the partial planes bound the chunk count: at most half of them for the big items, the rest for the small ones)
        const auto big_min = static_cast<long>((static_cast<uint64_t>(spp) * 2u + kMaxChunks - 1u) / kMaxChunks);
        c.big = static_cast<uint32_t>(std::clamp(std::max(std::lround(LR_TAPER_BIG * s_opt), big_min), 1l, static_cast<long>(spp)));
        c.small = static_cast<uint32_t>(std::clamp(std::lround(s_opt / LR_TAPER_SMALL), 1l, static_cast<long>(c.big)));
        c.big_count = static_cast<uint32_t>(std::floor(LR_TAPER_FRACTION * spp / c.big));
        const auto rest = spp - c.big_count * c.big;
        auto small_count = (rest + c.small - 1u) / c.small;
        if (c.big_count + small_count > kMaxChunks && c.big_count < kMaxChunks) {// (few tiles and many samples: what fits the partial planes)
            c.small = (rest + (kMaxChunks - c.big_count) - 1u) / (kMaxChunks - c.big_count);
            small_count = (rest + c.small - 1u) / c.small;
        }
        c.count = c.big_count + small_count;
        if (c.big_count >= 1u && small_count >= 1u && c.count <= kMaxChunks) { return c; }
    }
    auto count = static_cast<uint32_t>(std::lround(spp / s_opt));
    count = std::max(1u, std::min({count, spp, kMaxChunks}));
    c.count = c.big_count = count, c.big = (spp + count - 1u) / count, c.small = c.big;
    return c;
}

// ---- fixed-point film sums (dev_wavefront.h: radiance_to_fixed; the pool kernels of round 4 and the parked paths of wavefront
// mode).  One sample adds at most clamp x |shutter weight| per channel and a pixel takes at most `spp` of them in one lrhip_render; the
// scale is the largest power of two that keeps that sum below 2^61, at most 2^40 (1e-12 of absolute resolution).  Returns log2 of the
// scale, or -1 when fewer than kMinFixedBits fractional bits would be left -- a clamp used to switch clamping off (1e20, inf): such a
// film cannot be held in 64-bit fixed point, and lrhip_render takes the float-accumulating kernels of rounds 1-3 instead.
constexpr int kMinFixedBits = 24;
int fixed_point_bits(float film_clamp, float shutter_weight, uint32_t spp) {
    const auto bound = std::max(1.0, static_cast<double>(film_clamp) * std::max(1.0, std::fabs(static_cast<double>(shutter_weight)))) * std::max(1u, spp);
    if (!std::isfinite(bound)) { return -1; }
    const auto bits = std::min(40, 61 - static_cast<int>(std::ceil(std::log2(bound))));
    return bits >= kMinFixedBits ? bits : -1;
}
// the frame's fixed-point sums [pixel][rgb], zero between renders (wf_resolve_kernel clears what it adds to the film)
int ensure_accum(lrhip_ctx *ctx, uint32_t pixel_count) {
    const auto bytes = static_cast<size_t>(pixel_count) * 3u * sizeof(unsigned long long);
    if (ctx->wf_accum.bytes < bytes) {
        if (auto r = ensure(ctx->wf_accum, bytes); r != LRHIP_OK) { return r; }
        LR_HIP_CHECK(hipMemsetAsync(ctx->wf_accum.ptr, 0, bytes, ctx->stream));
    }
    return LRHIP_OK;
}
// Which scheduler a frame of this scene runs under (lrhip_set_scheduler).  Automatic: the pool kernels, except on scenes so small that
// a ray is a handful of traversal steps -- there the pool kernel's costlier shading block (path state through global memory, the
// current context through the LDS) is not paid back by fuller traversal steps.  Measured in round 4 at the bench's scenes (kernel
// time, one path per lane / pool, profiles/r04_final_schedulers.txt): Cornell box, 32 triangles, 0.88; C2 1.5 M triangles 1.08 (1024
// spp) ... 1.12 (256 spp); C3 1.18; C4 1.06; C5 (wavefront mode) 1.10.
// A room scene swept over its triangle count (profiles/archive/r04i_scheduler_crossover.txt): 0.88 at 2-5 thousand triangles, 0.93 at 12, 0.96 at
// 30, 1.06 at 100, 1.20 at 400 thousand.
// Round 5 (tools/sched_sweep.py, profiles/r05j_scheduler_sweep.txt: the room scene over its triangle count x path depth x spp, the Cornell
// box over depth x spp): what the pool buys grows with the LENGTH of the walks (triangles) and of the paths (depth) and with the share of
// a launch that is drain (few samples per pixel); the break-even moves from ~30 thousand triangles (depth 16, 16 spp) to ~100 thousand
// (depth 16, 256 spp), ~130 (depth 4, 16 spp) and ~180 (depth 4, 256 spp).  The rule follows it with the SCENE's own numbers -- its BVH
// triangles, its integrator's depth, the spp its description asks for (not the spp of this call: every call of a frame must take the
// same kernel family, whose film sums differ in their last bits) -- and is never more than 1.4 % off the better kernel in that sweep
// (the triangle count alone: 3.8 %).
constexpr uint32_t kPoolAutoTriangles = 98304u;
uint32_t pool_auto_triangles(uint32_t max_depth, uint32_t scene_spp) {
    auto t = kPoolAutoTriangles;
    if (max_depth <= 6u) { t *= 2u; }
    if (scene_spp != 0u && scene_spp < 64u) { t /= 2u; }
    return t;
}
bool wants_pool(const lrhip_ctx *ctx) {
    return ctx->scheduler == 2u || (ctx->scheduler == 0u && ctx->update_counts[1] >= pool_auto_triangles(ctx->scene.max_depth, ctx->scene.sampler_spp));
}
// path state of the pool kernels: two contexts per thread, 4 (Independent sampler) | 5 float4 each, [context][quad][thread] (megapool_kernel.h); sized for 5
int ensure_pool(lrhip_ctx *ctx, uint32_t resident_blocks) {
    return ensure(ctx->pool, static_cast<size_t>(resident_blocks) * lrd::kWavesPerBlock * lrd::kPoolSlots * lrd::pool_quads<true>() * sizeof(float4));
}

// ---- wavefront mode (dev_scene.h: WfArgs): a scene with Mix or Layered surfaces under the MegaPath integrator.  The frame is cut
// into SLICES of the sample range whose paths fit the queues (a path is parked at most once per round, so a queue never needs more
// slots than the slice has paths); per slice: the camera pass of the lean megakernel <.. | Wf> (its own work items, chunked by the
// same loss model as the plain megakernel), then up to max_depth ROUNDS of { heavy kernel -> continuation pass <.. | Wf | Cont> }.
// Nothing comes back to the host in between: the kernels read their record counts from device memory and the grids are the
// persistent ones (an empty round costs a few microseconds), so a slice is one uninterrupted stretch of the stream.
static int render_wavefront(lrhip_ctx *ctx, const lrhip_render_params *p, uint32_t tiles_x, uint32_t tiles_y, uint32_t tiles_in_range,
                            uint32_t tile_count, bool count, bool generic) {
    if (ctx->packed_texel_words != 0u) {// (lrhip_upload_scene packs no scene that renders in wavefront mode: its lean passes hold no 8-bit texel decode)
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_render: wavefront mode with 8-bit texels on the device (lrhip_set_texture_storage)");
    }
    const auto spp = p->spp_end - p->spp_begin;
    const auto pixel_count = ctx->width * ctx->height;
    const auto sampler_words = generic ? lrd::kWfSamplerWordsMax : 1u;
    // Slice size: every slice pays the latency of its last rounds (a handful of paths, one batch each), so slices are large -- C5 at
    // 512 spp: 356 / 404 / 442 / 472 Msamples/s with 2^25 / 2^26 / 2^27 / ~2^27.9 paths per slice.  A slice is 2^28 paths of ONE NOMINAL
    // SHARD of the frame (tile_count / balance_shards tiles, like the chunking): its length in samples is a function of the frame and
    // the caller's hint only -- never of the free memory or of the tile range of this call -- because the work items, and with
    // them the order of the film's float sums, are cut per slice: every shard of a frame, and the unsharded frame rendered with the
    // same hint, must cut them alike.  A path takes (3 queues x 15..18 words + 26..29 words) x 4 B = 284 .. 332 B: 2^28 of them are
    // 76 .. 89 GB of the 288.  What the memory does decide is how many TILES go through the queues at a time (tile groups, below):
    // a call over more tiles than fit -- the unsharded frame with a shard hint, a GPU with little memory left -- takes its tiles
    // group after group with the same slices, which regroups nothing (items are per tile; parked paths add in fixed point).
    const auto nominal_paths = ctx->wf_slice_paths != 0u ? static_cast<uint64_t>(ctx->wf_slice_paths) : (1ull << 28u);
    const auto nominal_tiles = std::max<uint64_t>(1u, static_cast<uint64_t>(static_cast<double>(tile_count) / std::max(p->balance_shards, 1u) + 0.5));
    const auto slice_spp = static_cast<uint32_t>(std::max<uint64_t>(1u, std::min<uint64_t>(spp, nominal_paths / (nominal_tiles * 64u))));
    const auto per_path = static_cast<uint64_t>(lrd::kWfKinds * (lrd::kWfHeavyWords + sampler_words) + lrd::kWfContWords + sampler_words) * sizeof(uint32_t);
    size_t free_bytes = 0u, total_bytes = 0u;
    LR_HIP_CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
    const auto have = free_bytes + ctx->wf_heavy.bytes + ctx->wf_cont.bytes;// (the queues of an earlier call count as free)
    // at most half of what is free, and at most kWfQueueBudget: the 2^28-path default slice needs 86-100 GB with its hand-over margin, more buys nothing
    auto fit_paths = std::min<uint64_t>(1ull << 30u,// (dev_wavefront.h: a slot's byte offset inside a queue column is 32 bits)
                                         std::max<uint64_t>(1ull << 16u, std::min<uint64_t>(have / 2u, kWfQueueBudget) / per_path));
    if (ctx->wf_mode == 2u) { fit_paths = 8ull * 64u * slice_spp; }// (tests: eight tiles at a time)
    // (round 6: the queues hold an eighth more than a slice's own paths -- room for what the slice before handed over, film_kernels.h: wf_carry_kernel)
    const auto group_tiles = static_cast<uint32_t>(std::max<uint64_t>(1u, std::min<uint64_t>(tiles_in_range, fit_paths * 8u / 9u / (64ull * slice_spp))));
    const auto slice_paths = std::min<uint64_t>(static_cast<uint64_t>(group_tiles) * 64u * slice_spp, (1ull << 30u) * 8u / 9u);
    const auto carry_margin = static_cast<uint32_t>(slice_paths / 8u);
    const auto capacity = static_cast<uint32_t>(slice_paths + carry_margin);
    const auto heavy_words = static_cast<size_t>(lrd::kWfKinds) * (lrd::kWfHeavyWords + sampler_words) * capacity;
    const auto cont_words = static_cast<size_t>(lrd::kWfContWords + sampler_words) * capacity;
    if (auto r = ensure(ctx->wf_heavy, heavy_words * sizeof(uint32_t)); r != LRHIP_OK) { return r; }
    if (auto r = ensure(ctx->wf_cont, cont_words * sizeof(uint32_t)); r != LRHIP_OK) { return r; }
    if (ctx->wf_counts.ptr == nullptr) {
        if (auto r = ensure(ctx->wf_counts, lrd::kWfCounterBufferWords * sizeof(uint32_t)); r != LRHIP_OK) { return r; }
    }
    if (auto r = ensure_accum(ctx, pixel_count); r != LRHIP_OK) { return r; }
    auto &scene = ctx->scene;
    scene.shutter_weight = (p->flags & LRHIP_RENDER_SHUTTER_WEIGHT) != 0u ? p->shutter_weight : 1.f;
    const auto scale_log2 = fixed_point_bits(scene.film_clamp, scene.shutter_weight, spp);// (>= kMinFixedBits: lrhip_render checked)
    const auto accum_scale = std::ldexp(1.0, scale_log2);
    scene.wf.heavy = static_cast<uint32_t *>(ctx->wf_heavy.ptr), scene.wf.cont = static_cast<uint32_t *>(ctx->wf_cont.ptr);
    scene.wf.counts = static_cast<uint32_t *>(ctx->wf_counts.ptr), scene.wf.capacity = capacity;
    scene.wf.accum = static_cast<unsigned long long *>(ctx->wf_accum.ptr), scene.wf.accum_scale = static_cast<float>(accum_scale);
    // kernels: the lean camera pass + continuation pass with the scene's environment / alpha needs, the heavy kernel with its nesting
    // (the alpha-tested traversal only where a surface may be non-opaque: the kitchen stand-in with its lace made opaque runs at 530.6
    // instead of 520.5 Msamples/s on the lean kernels without it, profiles/archive/r03ar_wavefront_without_alpha_ab.txt)
    auto lean = (ctx->features & (lrd::kFeatEnv | lrd::kFeatAlpha)) | lrd::kFeatWf | (count ? lrd::kFeatCount : 0u) | (generic ? lrd::kFeatGeneric : 0u);
    const auto n_variants = sizeof(kVariants) / sizeof(kVariants[0]);
    // round 4: both lean passes under the path-pool scheduler (megapool_kernel.h) where those kernels are in the library
    auto pool = false;
    if (wants_pool(ctx)) {
        const auto a = find_variant(kVariants, n_variants, lean | lrd::kFeatPool), b = find_variant(kVariants, n_variants, lean | lrd::kFeatPool | lrd::kFeatCont);
        pool = a >= 0 && b >= 0 && kVariants[a].launch != nullptr && kVariants[b].launch != nullptr;
    }
    if (pool) { lean |= lrd::kFeatPool; }
    scene.wf.count_at_flush = pool ? 1u : 0u;
    const auto pool_film = pool;// the camera pass sums its tiles into the frame's fixed-point sums: no partial planes
    const auto vi_camera = find_variant(kVariants, n_variants, lean), vi_cont = find_variant(kVariants, n_variants, lean | lrd::kFeatCont);
    const auto n_heavy = sizeof(kHeavyVariants) / sizeof(kHeavyVariants[0]);
    int hi[lrd::kWfKinds];// the heavy kernel of each closure kind (Disney has no nested form)
    auto heavy_ok = true;
    for (auto k = 0u; k < lrd::kWfKinds; k++) {
        hi[k] = find_variant(kHeavyVariants, n_heavy, (k << 2u) | (k != 0u && (ctx->features & lrd::kFeatNest) != 0u ? 512u : 0u) | (count ? 1u : 0u) | (generic ? 2u : 0u));
        heavy_ok = heavy_ok && hi[k] >= 0 && kHeavyVariants[hi[k]].launch != nullptr;
    }
    if (vi_camera < 0 || vi_cont < 0 || !heavy_ok || kVariants[vi_camera].launch == nullptr || kVariants[vi_cont].launch == nullptr) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_render: the wavefront kernels for feature mask " + std::to_string(ctx->features) + " were not compiled into this library");
    }
    // round 6: the two lean passes compiled for the PaddedSobol sampler, where the scene's sampler is that and both exist for the mask (variants.h: LR_PADDED_LIST)
    auto e_camera = &kVariants[vi_camera], e_cont = &kVariants[vi_cont];
    auto eb_camera = &ctx->variant_blocks[vi_camera], eb_cont = &ctx->variant_blocks[vi_cont];
    if (pool && generic && ctx->scene.sampler_kind == LR_SAMPLER_PADDED_SOBOL) {
        const auto pa = find_variant(kPaddedVariants, kPaddedVariantCount, kVariants[vi_camera].mask | lrd::kFeatPadded);
        const auto pb = find_variant(kPaddedVariants, kPaddedVariantCount, kVariants[vi_cont].mask | lrd::kFeatPadded);
        if (pa >= 0 && pb >= 0 && kPaddedVariants[pa].launch != nullptr && kPaddedVariants[pb].launch != nullptr) {
            e_camera = &kPaddedVariants[pa], e_cont = &kPaddedVariants[pb], eb_camera = &ctx->padded_blocks[pa], eb_cont = &ctx->padded_blocks[pb];
        }
    }
    auto blocks_of = [&](int &cache, const VariantEntry &v, int &out) -> int {
        if (cache < 0) {
            int per_cu = 0;
            LR_HIP_CHECK(v.occupancy(&per_cu));
            cache = std::max(1, std::min(per_cu, static_cast<int>(kMaxBlocksPerCu)));
        }
        out = cache;
        return LRHIP_OK;
    };
    int b_camera = 0, b_cont = 0, b_heavy[lrd::kWfKinds] = {0, 0, 0};
    if (auto r = blocks_of(*eb_camera, *e_camera, b_camera); r != LRHIP_OK) { return r; }
    if (auto r = blocks_of(*eb_cont, *e_cont, b_cont); r != LRHIP_OK) { return r; }
    for (auto k = 0u; k < lrd::kWfKinds; k++) {
        if (auto r = blocks_of(ctx->heavy_blocks[hi[k]], kHeavyVariants[hi[k]], b_heavy[k]); r != LRHIP_OK) { return r; }
    }
    // which closure kinds the scene holds at all (a Mix / Layered surface may reach a Disney child, which is shaded inside that kind's kernel)
    const bool has_kind[lrd::kWfKinds] = {(ctx->features & lrd::kFeatDisney) != 0u, (ctx->features & lrd::kFeatMix) != 0u, (ctx->features & lrd::kFeatLayered) != 0u};
    const auto resident = ctx->cu_count * static_cast<uint32_t>(std::max(b_camera, b_cont));
    if (auto r = ensure(ctx->spill, static_cast<size_t>(resident) * lrd::kBlockThreads * lrd::kSpillEntries * sizeof(uint32_t)); r != LRHIP_OK) { return r; }
    if (pool) {
        if (auto r = ensure_pool(ctx, resident); r != LRHIP_OK) { return r; }
    }
    if (auto r = ensure(ctx->scene_record, sizeof(lrd::DScene)); r != LRHIP_OK) { return r; }
    LR_HIP_CHECK(hipMemcpyAsync(ctx->scene_record.ptr, &scene, sizeof(lrd::DScene), hipMemcpyHostToDevice, ctx->stream));
    const auto device_scene = static_cast<const lrd::DScene *>(ctx->scene_record.ptr);
    lrd::RenderArgs args{};
    args.film = ctx->film;
    args.tile_begin = p->tile_begin, args.tile_end = p->tile_end, args.tile_stride = p->tile_stride;
    args.tiles_x = tiles_x, args.tiles_y = tiles_y;
    args.work_counter = static_cast<uint32_t *>(ctx->work_counter.ptr);
    args.spill = static_cast<uint32_t *>(ctx->spill.ptr);
    args.pool = static_cast<float4 *>(ctx->pool.ptr);
    args.counters = static_cast<lrd::DCounters *>(ctx->counters.ptr);
    const auto counts = static_cast<uint32_t *>(ctx->wf_counts.ptr);
    const auto shard_tiles = static_cast<double>(tile_count) / std::max(p->balance_shards, 1u);
    auto item_scale = 1.25;
    if (ctx->diag_item_scale != 0.) { item_scale *= std::max(0.01, std::fabs(ctx->diag_item_scale)); }
    if (!ctx->in_split) { LR_HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream)); }
    LR_HIP_CHECK(hipMemsetAsync(counts, 0, lrd::kWfCounterBufferWords * sizeof(uint32_t), ctx->stream));// (nothing handed over yet)
    // rounds a slice runs before it hands what is still parked over to the next one (the last slice of the call runs them all)
    const auto carry_rounds = ctx->diag_wf_carry_rounds != 0u ? ctx->diag_wf_carry_rounds : kWfCarryRounds;
    for (auto g0 = 0u; g0 < tiles_in_range; g0 += group_tiles) {// tile groups: what fits the queues at a time (see above)
    const auto group_count = std::min(group_tiles, tiles_in_range - g0);
    args.tile_begin = p->tile_begin + g0 * p->tile_stride;
    args.tile_end = std::min(p->tile_end, args.tile_begin + group_count * p->tile_stride);
    for (auto s0 = p->spp_begin; s0 < p->spp_end; s0 += slice_spp) {
        const auto s1 = std::min(p->spp_end, s0 + slice_spp);
        const auto n = s1 - s0;
        // ---- camera pass: samples [s0, s1) of every tile of the shard; heavy hits are parked
        const auto ck = chunking_of(n, shard_tiles, item_scale, ctx->diag_item_scale >= 0.);
        const auto chunk_count = ck.count;
        args.spp_begin = s0, args.spp_end = s1, args.chunk_count = chunk_count, args.item_count = group_count * chunk_count;
        args.chunk_big_count = ck.big_count, args.chunk_big = ck.big, args.chunk_small = ck.small;
        args.total_threads = ctx->cu_count * static_cast<uint32_t>(b_camera) * lrd::kBlockThreads;
        if (chunk_count > 1u && !pool_film) {// (the pool kernels add every item to the frame's fixed-point sums: no partial planes)
            if (auto r = ensure(ctx->partial, static_cast<size_t>(chunk_count) * pixel_count * sizeof(float4)); r != LRHIP_OK) { return r; }
            args.partial = static_cast<float4 *>(ctx->partial.ptr);
        }
        const auto last_slice = g0 + group_tiles >= tiles_in_range && s0 + slice_spp >= p->spp_end;
        LR_HIP_CHECK(hipMemsetAsync(ctx->work_counter.ptr, 0, 1024u, ctx->stream));
        LR_HIP_CHECK(hipMemsetAsync(counts, 0, lrd::kWfCounterWords * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(lrd::wf_carry_kernel, dim3(1), dim3(64), 0, ctx->stream, counts, carry_margin, 1u);// (the paths the slice before handed over)
        LR_HIP_CHECK(hipGetLastError());
        LR_HIP_CHECK(e_camera->launch(std::min(ctx->cu_count * static_cast<uint32_t>(b_camera), (args.item_count + 3u) / 4u), ctx->stream, device_scene, &args));
        if (chunk_count > 1u && !pool_film) {
            hipLaunchKernelGGL(lrd::resolve_partial_kernel, dim3((pixel_count + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->film,
                               args.partial, pixel_count, chunk_count, ctx->width, tiles_x, args.tile_begin, args.tile_end, p->tile_stride);
        }
        // ---- rounds: a path leaves a round either finished or parked again (one level deeper), so max_depth rounds empty the queues
        args.chunk_count = 1u, args.item_count = 0u;// (the continuation pass reads its item count from the device)
        args.chunk_big_count = 1u, args.chunk_big = 0u, args.chunk_small = 0u;
        for (auto round = 0u; round < std::max(scene.max_depth, 1u); round++) {
            for (auto k = 0u; k < lrd::kWfKinds; k++) {
                if (has_kind[k]) { LR_HIP_CHECK(kHeavyVariants[hi[k]].launch(ctx->cu_count * static_cast<uint32_t>(b_heavy[k]), ctx->stream, device_scene, &args)); }
            }
            // the heavy kernels have consumed the parked paths: their counters (and work counters) restart for the continuation pass
            LR_HIP_CHECK(hipMemsetAsync(counts + lrd::kWfCountHeavy, 0, 3u * sizeof(uint32_t), ctx->stream));
            LR_HIP_CHECK(hipMemsetAsync(counts + lrd::kWfWorkHeavy, 0, 3u * sizeof(uint32_t), ctx->stream));
            args.total_threads = ctx->cu_count * static_cast<uint32_t>(b_cont) * lrd::kBlockThreads;
            LR_HIP_CHECK(e_cont->launch(ctx->cu_count * static_cast<uint32_t>(b_cont), ctx->stream, device_scene, &args));
            LR_HIP_CHECK(hipMemsetAsync(counts + lrd::kWfCountCont, 0, 2u * sizeof(uint32_t), ctx->stream));// (+ its work counter, next to it)
            if (!last_slice && round + 1u >= carry_rounds && carry_rounds < 0xffffu) {
                hipLaunchKernelGGL(lrd::wf_carry_kernel, dim3(1), dim3(64), 0, ctx->stream, counts, carry_margin, 0u);
                LR_HIP_CHECK(hipGetLastError());
            }
        }
    }
    }
    hipLaunchKernelGGL(lrd::wf_resolve_kernel, dim3((pixel_count + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->film,
                       static_cast<unsigned long long *>(ctx->wf_accum.ptr), pixel_count, 1.0 / accum_scale);
    LR_HIP_CHECK(hipGetLastError());
    LR_HIP_CHECK(hipEventRecord(ctx->ev_end, ctx->stream));
    ctx->timed = true;
    // what rendered: the lean camera-pass kernel's mask + the closure bits the heavy kernel served
    ctx->last_variant = e_camera->mask | (ctx->features & (lrd::kFeatDisney | lrd::kFeatMix | lrd::kFeatLayered | lrd::kFeatNest));
    return LRHIP_OK;
}

// Which of the kernels that sum the film in 64-bit fixed point a call of this scene takes, PROVIDED the sums fit (fixed_point_bits):
// 1 wavefront mode (render_wavefront), 2 a pool kernel, 0 neither (the float-accumulating kernels of rounds 1-3).  One place for the
// conditions: lrhip_render decides with it whether a call beyond the fixed-point range is worth rendering in sample sub-ranges.
static int fixed_point_film_kind(const lrhip_ctx *ctx, bool count, bool generic) {
    if (ctx->wf_mode != 1u && ctx->diag_force_features == 0u && !ctx->env_tree && (ctx->features & (lrd::kFeatAux | lrd::kFeatVpt)) == 0u) {
        const auto plain = pick_variant(ctx->features, false, generic, false, ctx->packed_texel_words != 0u);
        if (plain >= 0 && (kVariants[plain].mask & (lrd::kFeatMix | lrd::kFeatLayered)) != 0u) { return 1; }
    }
    auto features = ctx->features | (ctx->diag_force_features & lrd::kFeatSceneMask);
    if (ctx->env_tree && (features & (lrd::kFeatAux | lrd::kFeatVpt)) == 0u) { features |= lrd::kFeatMix; }
    const auto byte_texels = ctx->packed_texel_words != 0u;
    const auto vi = pick_variant(features, count, generic, false, byte_texels);
    if (wants_pool(ctx) && ctx->scene.max_depth < 65536u && vi >= 0 && (kVariants[vi].mask & (lrd::kFeatMix | lrd::kFeatLayered | lrd::kFeatAux | lrd::kFeatVpt)) == 0u) {
        const auto vp = pick_variant(features, count, generic, true, byte_texels);
        if (vp >= 0 && kVariants[vp].launch != nullptr && kVariants[vp].occupancy != nullptr && (kVariants[vp].mask & lrd::kFeatWf) == 0u) { return 2; }
    }
    return 0;
}

int lrhip_render(lrhip_ctx *ctx, const lrhip_render_params *p) {
    if (ctx == nullptr || p == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_render: no scene uploaded"); }
    auto tiles_x = (ctx->width + 7u) / 8u, tiles_y = (ctx->height + 7u) / 8u;
    auto tile_count = tiles_x * tiles_y;
    if (p->spp_end < p->spp_begin || p->tile_stride == 0u || p->tile_end > tile_count) {
        return fail(LRHIP_ERROR_INVALID, "lrhip_render: invalid spp/tile range");
    }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->timed = false;
    if (p->spp_end == p->spp_begin || p->tile_begin >= p->tile_end) { return LRHIP_OK; }// (rank >= tile count: an empty shard)
    // MegakernelPathTracingInstance::_render_one_camera (mega_path.cpp:40-47): no lights -> nothing rendered
    // (the normal visualiser needs no light: normal.cpp has no such check)
    if (!ctx->scene.has_lights && ctx->scene.env_kind == lrd::kEnvNone && ctx->scene.integrator_kind != LR_INTEGRATOR_NORMAL) { return LRHIP_OK; }
    auto tiles_in_range = (p->tile_end - p->tile_begin + p->tile_stride - 1u) / p->tile_stride;
    auto spp = p->spp_end - p->spp_begin;
    // Wavefront mode (render_wavefront above) for every MegaPath scene that would otherwise land in an all-in-one variant with
    // out-of-line closures: Mix / Layered surfaces, and Disney together with an alpha test (no lean <Alpha | Disney> variant is
    // precompiled; such a scene ran at 433 Msamples/s on <60> where its Mix-holding sibling ran at 480 in wavefront mode).
    // (both need the film's fixed-point sums: a frame they cannot hold -- fixed_point_bits -- takes the float-accumulating kernels)
    const auto weight = (p->flags & LRHIP_RENDER_SHUTTER_WEIGHT) != 0u ? p->shutter_weight : 1.f;
    const auto fixed_bits = fixed_point_bits(ctx->scene.film_clamp, weight, spp);
    // A call whose sums do not fit the fixed-point film as a whole (clamp x spp beyond 2^37: a clamp of 1e7 at 65536 spp) is rendered in sample
    // sub-ranges that do, one resolve into the float film per range, instead of silently leaving wavefront mode and the pool kernels for the
    // all-in-one variants (the kitchen class runs at ~300 instead of ~560 Msamples/s there).  Only a clamp that does not even
    // hold ONE sample (switched off: 1e20, inf) still takes the float-accumulating kernels -- lrhip.h says so.
    // (Only where the frame WOULD be summed in fixed point -- wavefront mode or a pool kernel, the conditions of the two branches below:
    // the float-accumulating kernels render such a call in one launch as before, in their usual order of adds.  ADVICE r05.)
    const auto count_flag = (p->flags & LRHIP_RENDER_COUNTERS) != 0u;
    const auto generic_flag = ctx->scene.sampler_kind != LR_SAMPLER_INDEPENDENT;
    if (fixed_bits < 0 && !ctx->in_split && fixed_point_bits(ctx->scene.film_clamp, weight, 1u) >= 0 && fixed_point_film_kind(ctx, count_flag, generic_flag) != 0) {
        auto n = spp;
        while (n > 1u && fixed_point_bits(ctx->scene.film_clamp, weight, n) < 0) { n = (n + 1u) / 2u; }
        LR_HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream));// (before the flag: a failure here must not leave it set)
        struct SplitGuard {
            lrhip_ctx *c;
            ~SplitGuard() { c->in_split = false; }
        } guard{ctx};
        ctx->in_split = true;
        auto rc = LRHIP_OK;
        for (auto s0 = p->spp_begin; s0 < p->spp_end && rc == LRHIP_OK; s0 += n) {
            auto sub = *p;
            sub.spp_begin = s0, sub.spp_end = std::min(p->spp_end, s0 + n);
            rc = lrhip_render(ctx, &sub);
        }
        return rc;
    }
    if (fixed_bits >= 0 && fixed_point_film_kind(ctx, count_flag, generic_flag) == 1) {
        return render_wavefront(ctx, p, tiles_x, tiles_y, tiles_in_range, tile_count, count_flag, generic_flag);
    }
    // Chunking (chunking_of above: tapered items) is a function of the frame only (tile_count, spp, balance_shards), never of the device or the tile
    // range of this call: tile_count is that of ONE shard of the frame as the caller declares it (balance_shards), so that every shard
    // of a frame -- and the unsharded frame rendered with the same hint -- uses the same chunking.  (Uniform items, rounds 1-2, C2 at
    // 1024 spp: full frame 7 / 14 / 28 chunks -> 1992 / 1987 / 1987 ms; the 1/8 shard 14 / 28 / 56 / 64 chunks -> 305 / 271 / 266 / 265 ms.)
    auto shard_tiles = static_cast<double>(tile_count) / std::max(p->balance_shards, 1u);
    auto item_scale = 1.25;
    if (ctx->diag_item_scale != 0.) { item_scale *= std::max(0.01, std::fabs(ctx->diag_item_scale)); }// lrhip_set_diagnostics: sweep of the loss model's constant (< 0: uniform items)
    const auto ck = chunking_of(spp, shard_tiles, item_scale, ctx->diag_item_scale >= 0.);
    const auto chunk_count = ck.count;
    lrd::RenderArgs args{};
    args.film = ctx->film;
    ctx->scene.shutter_weight = (p->flags & LRHIP_RENDER_SHUTTER_WEIGHT) != 0u ? p->shutter_weight : 1.f;
    args.spp_begin = p->spp_begin, args.spp_end = p->spp_end;
    args.tile_begin = p->tile_begin, args.tile_end = p->tile_end, args.tile_stride = p->tile_stride;
    args.tiles_x = tiles_x, args.tiles_y = tiles_y;
    args.chunk_count = chunk_count;
    args.chunk_big_count = ck.big_count, args.chunk_big = ck.big, args.chunk_small = ck.small;
    args.item_count = tiles_in_range * chunk_count;
    args.work_counter = static_cast<uint32_t *>(ctx->work_counter.ptr);
    args.spill = static_cast<uint32_t *>(ctx->spill.ptr);
    args.total_threads = ctx->grid_blocks * lrd::kBlockThreads;
    args.counters = static_cast<lrd::DCounters *>(ctx->counters.ptr);
    auto pixel_count = ctx->width * ctx->height;
    LR_HIP_CHECK(hipMemsetAsync(ctx->work_counter.ptr, 0, 1024u, ctx->stream));
    auto count = (p->flags & LRHIP_RENDER_COUNTERS) != 0u;
    auto generic = ctx->scene.sampler_kind != LR_SAMPLER_INDEPENDENT;// generic-sampler instantiation
    auto features = ctx->features;
    features |= ctx->diag_force_features & lrd::kFeatSceneMask;// lrhip_set_diagnostics: A/B of a variant on a scene that does not need it
    // nested Combined environments are walked by out-of-line code (dev_shade.h: LR_ENV_TREE), which the variants that make real calls
    // anyway hold -- the ones with the Mix interpreter (the auxiliary and volumetric kernels are such variants already)
    if (ctx->env_tree && (features & (lrd::kFeatAux | lrd::kFeatVpt)) == 0u) { features |= lrd::kFeatMix; }
    const auto byte_texels = ctx->packed_texel_words != 0u;// (only a kernel that decodes 8-bit texels will do: pick_variant)
    auto vi = pick_variant(features, count, generic, false, byte_texels);
    // round 4: the path-pool scheduler (megapool_kernel.h) where a pool kernel is compiled for a scene the legacy search would have given
    // a lean kernel (no out-of-line closures, no sibling integrator), and the fixed-point film can hold the frame
    auto pool = false;
    if (wants_pool(ctx) && fixed_bits >= 0 && ctx->scene.max_depth < 65536u && vi >= 0 && (kVariants[vi].mask & (lrd::kFeatMix | lrd::kFeatLayered | lrd::kFeatAux | lrd::kFeatVpt)) == 0u) {
        const auto vp = pick_variant(features, count, generic, true, byte_texels);
        if (vp >= 0 && kVariants[vp].launch != nullptr && kVariants[vp].occupancy != nullptr && (kVariants[vp].mask & lrd::kFeatWf) == 0u) { vi = vp, pool = true; }
    }
    if (vi < 0 || kVariants[vi].launch == nullptr || kVariants[vi].occupancy == nullptr) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_render: no megakernel variant for feature mask " + std::to_string(ctx->features) +
                                                 " was compiled into this library");
    }
    // round 6: a pool kernel compiled for the PaddedSobol sampler, where the scene's sampler is that and such a kernel exists for the mask
    auto entry = &kVariants[vi];
    auto entry_blocks = &ctx->variant_blocks[vi];
    static_assert(kPaddedVariantCount <= sizeof(ctx->padded_blocks) / sizeof(ctx->padded_blocks[0]), "lrhip_ctx::padded_blocks");
    if (pool && generic && ctx->scene.sampler_kind == LR_SAMPLER_PADDED_SOBOL) {
        const auto pv = find_variant(kPaddedVariants, kPaddedVariantCount, kVariants[vi].mask | lrd::kFeatPadded);
        if (pv >= 0 && kPaddedVariants[pv].launch != nullptr && kPaddedVariants[pv].occupancy != nullptr) { entry = &kPaddedVariants[pv], entry_blocks = &ctx->padded_blocks[pv]; }
    }
    if (*entry_blocks < 0) {
        int blocks_per_cu = 0;
        LR_HIP_CHECK(entry->occupancy(&blocks_per_cu));
        *entry_blocks = std::max(1, std::min(blocks_per_cu, static_cast<int>(kMaxBlocksPerCu)));
    }
    auto resident = ctx->cu_count * static_cast<uint32_t>(*entry_blocks);
    args.total_threads = resident * lrd::kBlockThreads;
    const auto pool_film = pool;
    if (chunk_count > 1u && !pool_film) {// (the pool kernels add every item to the frame's fixed-point sums: no partial planes)
        if (auto r = ensure(ctx->partial, static_cast<size_t>(chunk_count) * pixel_count * sizeof(float4)); r != LRHIP_OK) { return r; }
        args.partial = static_cast<float4 *>(ctx->partial.ptr);
    }
    auto blocks = std::min(resident, (args.item_count + 3u) / 4u);
    if (pool) {
        if (auto r = ensure_pool(ctx, resident); r != LRHIP_OK) { return r; }
        args.pool = static_cast<float4 *>(ctx->pool.ptr);
        // the frame's fixed-point sums (megapool_kernel.h: FILM)
        if (auto r = ensure_accum(ctx, pixel_count); r != LRHIP_OK) { return r; }
        ctx->scene.wf.accum = static_cast<unsigned long long *>(ctx->wf_accum.ptr);
        ctx->scene.wf.accum_scale = static_cast<float>(std::ldexp(1.0, fixed_bits));
    }
    // how this launch's kernel counts its samples, set on EVERY path (the field lives in ctx->scene and used to keep whatever the
    // last pool / wavefront call left in it): a pool kernel counts an item's samples when its wave leaves the item, the others where they finish
    ctx->scene.wf.count_at_flush = pool ? 1u : 0u;
    // the scene record of THIS launch (shutter weight, film clamp, ...) in stream order; ctx->scene is pageable host memory, so
    // the copy has left it when the call returns
    if (auto r = ensure(ctx->scene_record, sizeof(lrd::DScene)); r != LRHIP_OK) { return r; }
    LR_HIP_CHECK(hipMemcpyAsync(ctx->scene_record.ptr, &ctx->scene, sizeof(lrd::DScene), hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->in_split) { LR_HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream)); }
    LR_HIP_CHECK(entry->launch(blocks, ctx->stream, static_cast<const lrd::DScene *>(ctx->scene_record.ptr), &args));
    ctx->last_variant = entry->mask;
    LR_HIP_CHECK(hipGetLastError());
    LR_HIP_CHECK(hipEventRecord(ctx->ev_end, ctx->stream));
    ctx->timed = true;
    if (pool_film) {// the frame's fixed-point sums join the film (and are cleared for the next call)
        hipLaunchKernelGGL(lrd::wf_resolve_kernel, dim3((pixel_count + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->film,
                           static_cast<unsigned long long *>(ctx->wf_accum.ptr), pixel_count, std::ldexp(1.0, -fixed_bits));
        LR_HIP_CHECK(hipGetLastError());
    } else if (chunk_count > 1u) {
        hipLaunchKernelGGL(lrd::resolve_partial_kernel, dim3((pixel_count + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->film,
                           args.partial, pixel_count, chunk_count, ctx->width, tiles_x, p->tile_begin, p->tile_end, p->tile_stride);
        LR_HIP_CHECK(hipGetLastError());
    }
    return LRHIP_OK;
}

uint64_t lrhip_packed_texels(lrhip_ctx *ctx) { return ctx != nullptr ? ctx->packed_texel_words : 0u; }

int lrhip_set_texture_storage(lrhip_ctx *ctx, uint32_t mode) {
    if (ctx == nullptr || mode > 2u) { return fail(LRHIP_ERROR_INVALID, "lrhip_set_texture_storage: invalid argument"); }
    ctx->byte_textures = mode;
    return LRHIP_OK;
}

uint32_t lrhip_pool_auto_triangles(uint32_t max_depth, uint32_t scene_spp) { return pool_auto_triangles(max_depth, scene_spp); }

int lrhip_set_diagnostics(lrhip_ctx *ctx, uint32_t force_features, double item_scale) {
    if (ctx == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_set_diagnostics: ctx is NULL"); }
    ctx->diag_force_features = force_features, ctx->diag_item_scale = item_scale;
    return LRHIP_OK;
}

int lrhip_work_items(uint32_t width, uint32_t height, uint32_t spp, uint32_t balance_shards, uint32_t out[4]) {
    if (out == nullptr || width == 0u || height == 0u || spp == 0u) { return fail(LRHIP_ERROR_INVALID, "lrhip_work_items: invalid argument"); }
    const auto tile_count = ((width + 7u) / 8u) * ((height + 7u) / 8u);
    const auto c = chunking_of(spp, static_cast<double>(tile_count) / std::max(balance_shards, 1u), 1.25, true);
    out[0] = c.count, out[1] = c.big_count, out[2] = c.big, out[3] = c.small;
    return LRHIP_OK;
}

int lrhip_set_scheduler(lrhip_ctx *ctx, uint32_t mode) {
    if (ctx == nullptr || mode > 2u) { return fail(LRHIP_ERROR_INVALID, "lrhip_set_scheduler: invalid argument"); }
    ctx->scheduler = mode;
    return LRHIP_OK;
}

int lrhip_set_wavefront(lrhip_ctx *ctx, uint32_t mode, uint32_t slice_paths) {
    if (ctx == nullptr || (mode & 0xffu) > 2u || (mode >> 24u) != 0u) { return fail(LRHIP_ERROR_INVALID, "lrhip_set_wavefront: invalid argument"); }
    ctx->wf_mode = mode & 0xffu, ctx->wf_slice_paths = slice_paths;
    ctx->diag_wf_carry_rounds = mode >> 8u;// (bits 8-23: rounds before a slice hands its parked paths over; 0 = default, 65535 = never)
    return LRHIP_OK;
}

int lrhip_synchronize(lrhip_ctx *ctx) {
    if (ctx == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_synchronize: ctx is NULL"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LRHIP_OK;
}

int lrhip_film_download(lrhip_ctx *ctx, float *rgba, int converted) {
    if (ctx == nullptr || rgba == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_film_download: invalid argument"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    auto pixel_count = ctx->width * ctx->height;
    auto bytes = static_cast<size_t>(pixel_count) * sizeof(float4);
    const void *src = ctx->film;
    if (converted) {
        hipLaunchKernelGGL(lrd::film_convert_kernel, dim3((pixel_count + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->film,
                           static_cast<float4 *>(ctx->converted.ptr), pixel_count, ctx->film_scale[0], ctx->film_scale[1], ctx->film_scale[2]);
        LR_HIP_CHECK(hipGetLastError());
        src = ctx->converted.ptr;
    }
    LR_HIP_CHECK(hipMemcpyAsync(rgba, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LRHIP_OK;
}

namespace {
// librccl.so is loaded on first use (no link-time dependency): the handful of entry points the multi-GPU path needs
struct Rccl {
    using unique_id = struct { char internal[128]; };
    int (*get_unique_id)(unique_id *){nullptr};
    int (*comm_init_rank)(void **, int, unique_id, int){nullptr};
    int (*comm_init_all)(void **, int, const int *){nullptr};
    int (*comm_destroy)(void *){nullptr};
    int (*reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t){nullptr};
    int (*group_start)(){nullptr};
    int (*group_end)(){nullptr};
    int (*comm_count)(void *, int *){nullptr};// (optional: lrhip_comm_info)
    int (*comm_user_rank)(void *, int *){nullptr};
    int (*comm_device)(void *, int *){nullptr};
    bool ok{false};
    Rccl() {
        auto lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (lib == nullptr) { lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); }
        if (lib == nullptr) { return; }
        get_unique_id = reinterpret_cast<decltype(get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        comm_init_rank = reinterpret_cast<decltype(comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(lib, "ncclCommInitAll"));
        comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(lib, "ncclReduce"));
        group_start = reinterpret_cast<decltype(group_start)>(dlsym(lib, "ncclGroupStart"));
        group_end = reinterpret_cast<decltype(group_end)>(dlsym(lib, "ncclGroupEnd"));
        comm_count = reinterpret_cast<decltype(comm_count)>(dlsym(lib, "ncclCommCount"));
        comm_user_rank = reinterpret_cast<decltype(comm_user_rank)>(dlsym(lib, "ncclCommUserRank"));
        comm_device = reinterpret_cast<decltype(comm_device)>(dlsym(lib, "ncclCommCuDevice"));
        ok = get_unique_id && comm_init_rank && comm_init_all && comm_destroy && reduce && group_start && group_end;
    }
};
extern "C++" const Rccl &rccl() {
    static Rccl r;
    return r;
}
}// namespace

int lrhip_device_count(int *count) {
    if (count == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_device_count: NULL argument"); }
    LR_HIP_CHECK(hipGetDeviceCount(count));
    return LRHIP_OK;
}

int lrhip_comm_unique_id(unsigned char id[LRHIP_COMM_ID_BYTES]) {
    if (id == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_comm_unique_id: NULL argument"); }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_comm_unique_id: librccl.so could not be loaded"); }
    Rccl::unique_id u{};
    static_assert(sizeof(u) == LRHIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (auto rc = rccl().get_unique_id(&u); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclGetUniqueId failed with code " + std::to_string(rc)); }
    std::memcpy(id, &u, sizeof(u));
    return LRHIP_OK;
}

int lrhip_comm_init_rank(lrhip_ctx *ctx, int world, int rank, const unsigned char id[LRHIP_COMM_ID_BYTES], void **comm) {
    if (ctx == nullptr || id == nullptr || comm == nullptr || world < 1 || rank < 0 || rank >= world) { return fail(LRHIP_ERROR_INVALID, "lrhip_comm_init_rank: invalid argument"); }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_comm_init_rank: librccl.so could not be loaded"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    Rccl::unique_id u{};
    std::memcpy(&u, id, sizeof(u));
    if (auto rc = rccl().comm_init_rank(comm, world, u, rank); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommInitRank failed with code " + std::to_string(rc)); }
    return LRHIP_OK;
}

int lrhip_comm_init_all(int count, const int *devices, void **comms) {
    if (count < 1 || devices == nullptr || comms == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_comm_init_all: invalid argument"); }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_comm_init_all: librccl.so could not be loaded"); }
    if (auto rc = rccl().comm_init_all(comms, count, devices); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommInitAll failed with code " + std::to_string(rc)); }
    return LRHIP_OK;
}

int lrhip_comm_destroy(void *comm) {
    if (comm == nullptr) { return LRHIP_OK; }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_comm_destroy: librccl.so could not be loaded"); }
    if (auto rc = rccl().comm_destroy(comm); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommDestroy failed with code " + std::to_string(rc)); }
    return LRHIP_OK;
}

int lrhip_comm_info(void *comm, int out[3]) {
    if (comm == nullptr || out == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_comm_info: NULL argument"); }
    if (!rccl().ok || !rccl().comm_count || !rccl().comm_user_rank || !rccl().comm_device) {
        return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_comm_info: librccl.so (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) could not be loaded");
    }
    if (auto rc = rccl().comm_count(comm, out + 0); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommCount failed with code " + std::to_string(rc)); }
    if (auto rc = rccl().comm_user_rank(comm, out + 1); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommUserRank failed with code " + std::to_string(rc)); }
    if (auto rc = rccl().comm_device(comm, out + 2); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclCommCuDevice failed with code " + std::to_string(rc)); }
    return LRHIP_OK;
}

int lrhip_film_reduce(lrhip_ctx *ctx, void *nccl_comm, int root) {
    if (ctx == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_film_reduce: no scene uploaded"); }
    if (nccl_comm == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_film_reduce: communicator is NULL"); }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_film_reduce: librccl.so (ncclReduce) could not be loaded"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    auto count = static_cast<size_t>(ctx->width) * ctx->height * 4u;
    // ncclReduce(sendbuff, recvbuff, count, ncclFloat32 = 7, ncclSum = 0, root, comm, stream), rccl.h
    if (auto rc = rccl().reduce(ctx->film, ctx->film, count, 7, 0, root, nccl_comm, ctx->stream); rc != 0) {
        return fail(LRHIP_ERROR_DEVICE, "lrhip_film_reduce: ncclReduce failed with code " + std::to_string(rc));
    }
    return LRHIP_OK;
}

// One host thread drives several contexts of one process (the C++ host's multi-GPU path): the reduces of all of them go out as
// ONE group, as RCCL requires of a single thread that owns several communicators.
int lrhip_film_reduce_group(int count, lrhip_ctx *const *ctxs, void *const *comms, int root) {
    if (count < 1 || ctxs == nullptr || comms == nullptr) { return fail(LRHIP_ERROR_INVALID, "lrhip_film_reduce_group: invalid argument"); }
    if (!rccl().ok) { return fail(LRHIP_ERROR_UNSUPPORTED, "lrhip_film_reduce_group: librccl.so could not be loaded"); }
    if (auto rc = rccl().group_start(); rc != 0) { return fail(LRHIP_ERROR_DEVICE, "ncclGroupStart failed with code " + std::to_string(rc)); }
    auto status = LRHIP_OK;
    for (auto i = 0; i < count && status == LRHIP_OK; i++) { status = lrhip_film_reduce(ctxs[i], comms[i], root); }
    if (auto rc = rccl().group_end(); rc != 0 && status == LRHIP_OK) { return fail(LRHIP_ERROR_DEVICE, "ncclGroupEnd failed with code " + std::to_string(rc)); }
    return status;
}

int lrhip_get_counters(lrhip_ctx *ctx, lrhip_counters *out) {
    if (ctx == nullptr || out == nullptr || !ctx->scene_ready) { return fail(LRHIP_ERROR_INVALID, "lrhip_get_counters: invalid argument"); }
    LR_HIP_CHECK(hipSetDevice(ctx->device));
    LR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    static_assert(sizeof(lrhip_counters) == sizeof(lrd::DCounters), "counter layouts must match");
    LR_HIP_CHECK(hipMemcpy(out, ctx->counters.ptr, sizeof(lrhip_counters), hipMemcpyDeviceToHost));
    return LRHIP_OK;
}

uint32_t lrhip_last_variant(lrhip_ctx *ctx) { return ctx != nullptr ? ctx->last_variant : 0u; }

double lrhip_last_render_ms(lrhip_ctx *ctx) {
    if (ctx == nullptr || !ctx->timed) { return 0.0; }
    if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(ctx->ev_end) != hipSuccess) { return -1.0; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) != hipSuccess) { return -1.0; }
    return static_cast<double>(ms);
}

}// extern "C"
