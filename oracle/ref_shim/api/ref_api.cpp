// ref_api.cpp -- the C entry points of oracle/_ref/libref.so.
//
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim; see ../lc_types.h).  libref.so is the REFERENCE's own render code
// (/root/reference/src/{util,sdl,base}/*.cpp and its plugins, compiled in place by oracle/Makefile.ref against the scalar
// LuisaCompute stand-in of this directory) behind a few `extern "C"` hooks, so that tests/test_oracle_vs_ref.py can hold
// oracle/ against what the reference's code computes -- per function, per sample and per image.  Everything below calls
// reference code; the only arithmetic that is NOT the reference's is what LuisaCompute itself would have supplied
// (builtins, the ray-tracing unit, texture filtering: ../lc_types.h, ../lc_runtime.h).
#include <util/rng.h>
#include <util/sampling.h>
#include <util/scattering.h>
#include <util/frame.h>
#include <util/thread_pool.h>
#include <sdl/scene_desc.h>
#include <sdl/scene_parser.h>
#include <base/scene.h>
#include <base/pipeline.h>
#include <base/integrator.h>

#include <map>
#include <csignal>
#include <execinfo.h>

namespace libref {
std::map<std::string, std::vector<float>> &saved_images() noexcept;// ref_imageio.cpp
}

using namespace luisa;
using namespace luisa::compute;
using namespace luisa::render;

namespace {

struct LiAccess : public ProgressiveIntegrator::Instance {
    using ProgressiveIntegrator::Instance::Li;// protected in the reference; reached through a pointer to member
};

struct RefScene {
    Context context;
    Device device;
    Stream stream;
    luisa::unique_ptr<SceneDesc> desc;
    luisa::unique_ptr<Scene> scene;
    luisa::unique_ptr<Pipeline> pipeline;
};

[[nodiscard]] Interaction flat_patch(const RefScene &s, uint inst, const float ns_in[3], float3 wo) noexcept {
    auto handle = s.pipeline->geometry()->instance(inst);
    auto ng = make_float3(0.f, 0.f, 1.f);
    auto ns = normalize(make_float3(ns_in[0], ns_in[1], ns_in[2]));
    return Interaction{handle, inst, 0u, 1.f, make_float3(0.f), ng, make_float2(.25f, .75f), make_float3(0.f),
                       face_forward(ns, ng), Frame::make(ng).s(), dot(wo, ng) < 0.f};
}

}// namespace

// Heap objects start zeroed, as DSL variables do (see oracle/Makefile.ref on -ftrivial-auto-var-init=zero).  libref.so and its
// plugins resolve operator new here; nothing else in the process does (the library is test infrastructure, loaded by ctypes).
void *operator new(std::size_t n) {
    if (auto p = std::calloc(1u, n == 0u ? 1u : n)) { return p; }
    throw std::bad_alloc{};
}
void *operator new[](std::size_t n) { return ::operator new(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

static void libref_abort_handler(int) {// LUISA_ERROR / assert -> abort: say where, the reference's code is not ours to read blind
    void *frames[48];
    auto n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    std::_Exit(134);
}

extern "C" {

// ---- scene ------------------------------------------------------------------------------------------------------------------
void *ref_scene_load(const char *scene_file, const char *plugin_dir) {
    if (std::getenv("LIBREF_BACKTRACE") != nullptr) { std::signal(SIGABRT, libref_abort_handler); }
    auto s = new RefScene{Context{std::filesystem::path{plugin_dir}}};
    SceneParser::MacroMap macros;
    s->desc = SceneParser::parse(scene_file, macros);
    s->scene = Scene::create(s->context, s->desc.get());
    s->pipeline = Pipeline::create(s->device, s->stream, *s->scene);
    if (auto integrator = s->pipeline->integrator(); integrator != nullptr && integrator->sampler() != nullptr) {
        luisa::render::CommandBuffer cb{&s->stream};
        auto camera = s->pipeline->camera(0u);
        auto res = camera->film()->node()->resolution();
        integrator->sampler()->reset(cb, res, res.x * res.y, camera->node()->spp());
    }
    return s;
}
void ref_scene_destroy(void *scene) { delete static_cast<RefScene *>(scene); }

// Pipeline::render (src/base/pipeline.cpp:115-117) of every camera; out = the converted film of camera 0 as the reference's
// save_image received it (float RGBA, row 0 = top)
int ref_render(void *scene, float *rgba_out) {
    auto s = static_cast<RefScene *>(scene);
    libref::saved_images().clear();
    s->pipeline->render(s->stream);
    auto path = s->pipeline->camera(0u)->node()->file().string();
    auto it = libref::saved_images().find(path);
    if (it == libref::saved_images().end()) { return -1; }
    std::memcpy(rgba_out, it->second.data(), it->second.size() * sizeof(float));
    return 0;
}
void ref_resolution(void *scene, uint32_t out[2]) {
    auto r = static_cast<RefScene *>(scene)->pipeline->camera(0u)->film()->node()->resolution();
    out[0] = r.x, out[1] = r.y;
}

// Li of one (pixel, sample index) at shutter time `time` -- the integrator's own Li (e.g. src/integrators/mega_path.cpp:49-156)
void ref_li(void *scene, uint32_t px, uint32_t py, uint32_t sample_index, float time, float rgb_out[3]) {
    auto s = static_cast<RefScene *>(scene);
    auto integrator = dynamic_cast<ProgressiveIntegrator::Instance *>(s->pipeline->integrator());
    auto L = (integrator->*(&LiAccess::Li))(s->pipeline->camera(0u), sample_index, make_uint2(px, py), time);
    rgb_out[0] = L.x, rgb_out[1] = L.y, rgb_out[2] = L.z;
}

// Camera::Instance::generate_ray (src/base/camera.cpp:212-224) with the draws Li makes before it: out = origin, direction, weight
void ref_camera_ray(void *scene, uint32_t px, uint32_t py, uint32_t sample_index, float time, float out[7]) {
    auto s = static_cast<RefScene *>(scene);
    auto sampler = s->pipeline->integrator()->sampler();
    auto camera = s->pipeline->camera(0u);
    sampler->start(make_uint2(px, py), sample_index);
    auto u_filter = sampler->generate_pixel_2d();
    auto u_lens = camera->node()->requires_lens_sampling() ? sampler->generate_2d() : make_float2(.5f);
    auto cs = camera->generate_ray(make_uint2(px, py), time, u_filter, u_lens);
    auto o = cs.ray->origin(), d = cs.ray->direction();
    out[0] = o.x, out[1] = o.y, out[2] = o.z, out[3] = d.x, out[4] = d.y, out[5] = d.z, out[6] = cs.weight;
}

// out[0..1] = generate_pixel_2d(), out[2 .. 2 + n) = the following generate_1d() draws of the scene's sampler
void ref_sampler_stream(void *scene, uint32_t px, uint32_t py, uint32_t sample_index, uint32_t n, float *out) {
    auto sampler = static_cast<RefScene *>(scene)->pipeline->integrator()->sampler();
    sampler->start(make_uint2(px, py), sample_index);
    auto p = sampler->generate_pixel_2d();
    out[0] = p.x, out[1] = p.y;
    for (auto i = 0u; i < n; i++) { out[2u + i] = sampler->generate_1d(); }
}

// Filter::Instance::sample (src/base/filter.cpp:49-64) of camera 0's filter: out = offset.x, offset.y, weight
void ref_filter_sample(void *scene, float ux, float uy, float out[3]) {
    auto f = static_cast<RefScene *>(scene)->pipeline->camera(0u)->filter()->sample(make_float2(ux, uy));
    out[0] = f.offset.x, out[1] = f.offset.y, out[2] = f.weight;
}
// the tables Filter::Instance::Instance builds (src/base/filter.cpp:24-47): lut[64], pdf[63], alias_prob[63], alias_idx[63]
void ref_filter_tables(void *scene, float *lut, float *pdf, float *alias_prob, uint32_t *alias_idx) {
    auto f = static_cast<RefScene *>(scene)->pipeline->camera(0u)->filter();
    std::copy(f->look_up_table().begin(), f->look_up_table().end(), lut);
    std::copy(f->pdf_table().begin(), f->pdf_table().end(), pdf);
    std::copy(f->alias_table_probabilities().begin(), f->alias_table_probabilities().end(), alias_prob);
    std::copy(f->alias_table_indices().begin(), f->alias_table_indices().end(), alias_idx);
}

// Geometry::trace_closest (src/base/geometry.cpp:218-261): out_ids = {inst, prim}, out_bary = {u, v}
void ref_trace_closest(void *scene, const float o[3], const float d[3], float t_min, float t_max, uint32_t out_ids[2], float out_bary[2]) {
    auto s = static_cast<RefScene *>(scene);
    auto hit = s->pipeline->geometry()->trace_closest(make_ray(make_float3(o[0], o[1], o[2]), make_float3(d[0], d[1], d[2]), t_min, t_max));
    out_ids[0] = hit.inst, out_ids[1] = hit.prim, out_bary[0] = hit.bary.x, out_bary[1] = hit.bary.y;
}

uint32_t ref_instance_count(void *scene) { return static_cast<uint32_t>(static_cast<RefScene *>(scene)->pipeline->geometry()->instances().size()); }
// the packed instance handle Shape::Handle::encode produced (src/base/shape.cpp:46-70)
void ref_instance_handle(void *scene, uint32_t inst, uint32_t out[4]) {
    auto h = static_cast<RefScene *>(scene)->pipeline->geometry()->instances()[inst];
    out[0] = h.x, out[1] = h.y, out[2] = h.z, out[3] = h.w;
}
uint32_t ref_surface_count(void *scene) { return static_cast<uint32_t>(static_cast<RefScene *>(scene)->pipeline->surfaces().size()); }
uint32_t ref_light_count(void *scene) { return static_cast<uint32_t>(static_cast<RefScene *>(scene)->pipeline->lights().size()); }

// Geometry::interaction of (inst, prim, bary) seen from wo (src/base/geometry.cpp:281-389):
// out = p[3], ng[3], ns[3], s[3], t[3], uv[2], area, back_facing  (19 floats)
void ref_interaction(void *scene, uint32_t inst, uint32_t prim, float bu, float bv, const float wo[3], float out[19]) {
    auto s = static_cast<RefScene *>(scene);
    auto it = s->pipeline->geometry()->interaction(inst, prim, make_float3(1.f - bu - bv, bu, bv), make_float3(wo[0], wo[1], wo[2]));
    auto put = [&out](int i, float3 v) noexcept { out[i] = v.x, out[i + 1] = v.y, out[i + 2] = v.z; };
    put(0, it->p()), put(3, it->ng()), put(6, it->shading().n()), put(9, it->shading().s()), put(12, it->shading().t());
    out[15] = it->uv().x, out[16] = it->uv().y, out[17] = it->triangle_area(), out[18] = it->back_facing() ? 1.f : 0.f;
}

// Surface::Closure::evaluate / sample (src/base/surface.cpp:45-68) of the surface of instance `inst` on a flat patch with
// geometric normal +z, shading normal ns, uv = (.25, .75) -- the patch of oracle_surface_evaluate / oracle_surface_sample.
// evaluate: out = f.rgb, pdf;  sample: out = f.rgb, pdf, wi.xyz, event
void ref_surface_evaluate(void *scene, uint32_t inst, const float ns[3], const float wo_in[3], const float wi_in[3], float out[4]) {
    auto s = static_cast<RefScene *>(scene);
    auto wo = make_float3(wo_in[0], wo_in[1], wo_in[2]), wi = make_float3(wi_in[0], wi_in[1], wi_in[2]);
    auto it = flat_patch(*s, inst, ns, wo);
    auto spectrum = s->pipeline->spectrum();
    auto swl = spectrum->sample(0.f);
    PolymorphicCall<Surface::Closure> call;
    s->pipeline->surfaces().dispatch(it.shape().surface_tag(), [&](auto surface) noexcept { surface->closure(call, it, swl, wo, 1.f, 0.f); });
    call.execute([&](const Surface::Closure *closure) noexcept {
        auto eval = closure->evaluate(wo, wi);
        auto f = spectrum->srgb(swl, eval.f);
        out[0] = f.x, out[1] = f.y, out[2] = f.z, out[3] = eval.pdf;
    });
}
void ref_surface_sample(void *scene, uint32_t inst, const float ns[3], const float wo_in[3], float u_lobe, float ux, float uy, float out[8]) {
    auto s = static_cast<RefScene *>(scene);
    auto wo = make_float3(wo_in[0], wo_in[1], wo_in[2]);
    auto it = flat_patch(*s, inst, ns, wo);
    auto spectrum = s->pipeline->spectrum();
    auto swl = spectrum->sample(0.f);
    PolymorphicCall<Surface::Closure> call;
    s->pipeline->surfaces().dispatch(it.shape().surface_tag(), [&](auto surface) noexcept { surface->closure(call, it, swl, wo, 1.f, 0.f); });
    call.execute([&](const Surface::Closure *closure) noexcept {
        auto ss = closure->sample(wo, u_lobe, make_float2(ux, uy));
        auto f = spectrum->srgb(swl, ss.eval.f);
        out[0] = f.x, out[1] = f.y, out[2] = f.z, out[3] = ss.eval.pdf;
        out[4] = ss.wi.x, out[5] = ss.wi.y, out[6] = ss.wi.z, out[7] = static_cast<float>(ss.event);
    });
}

// LightSampler::Instance::sample (src/base/light_sampler.cpp:57-62) from a flat patch at p with normal n:
// out = L.rgb, pdf, shadow ray origin[3], direction[3], t_max  (11 floats)
void ref_light_sample(void *scene, const float p[3], const float n_in[3], float u_sel, float ux, float uy, float out[11]) {
    auto s = static_cast<RefScene *>(scene);
    auto ng = normalize(make_float3(n_in[0], n_in[1], n_in[2]));
    auto handle = s->pipeline->geometry()->instance(0u);
    Interaction it{handle, 0u, 0u, 1.f, make_float3(p[0], p[1], p[2]), ng, make_float2(.25f, .75f),
                   make_float3(p[0], p[1], p[2]), ng, Frame::make(ng).s(), false};
    auto spectrum = s->pipeline->spectrum();
    auto swl = spectrum->sample(0.f);
    auto ls = s->pipeline->integrator()->light_sampler()->sample(it, u_sel, make_float2(ux, uy), swl, 0.f);
    auto L = spectrum->srgb(swl, ls.eval.L);
    auto o = ls.shadow_ray->origin(), d = ls.shadow_ray->direction();
    out[0] = L.x, out[1] = L.y, out[2] = L.z, out[3] = ls.eval.pdf;
    out[4] = o.x, out[5] = o.y, out[6] = o.z, out[7] = d.x, out[8] = d.y, out[9] = d.z, out[10] = ls.shadow_ray->t_max();
}
// LightSampler::Instance::evaluate_miss (environment seen along wi): out = L.rgb, pdf
void ref_evaluate_miss(void *scene, const float wi[3], float out[4]) {
    auto s = static_cast<RefScene *>(scene);
    auto spectrum = s->pipeline->spectrum();
    auto swl = spectrum->sample(0.f);
    auto e = s->pipeline->integrator()->light_sampler()->evaluate_miss(make_float3(wi[0], wi[1], wi[2]), swl, 0.f);
    auto L = spectrum->srgb(swl, e.L);
    out[0] = L.x, out[1] = L.y, out[2] = L.z, out[3] = e.pdf;
}

// ---- unit-level hooks (no scene): src/util/rng.cpp, sampling.cpp, scattering.cpp, frame.cpp -------------------------------------
uint32_t ref_xxhash32_1(uint32_t x) { return xxhash32(x); }
uint32_t ref_xxhash32_2(uint32_t x, uint32_t y) { return xxhash32(make_uint2(x, y)); }
uint32_t ref_xxhash32_3(uint32_t x, uint32_t y, uint32_t z) { return xxhash32(make_uint3(x, y, z)); }
uint32_t ref_xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return xxhash32(make_uint4(x, y, z, w)); }
uint32_t ref_pcg(uint32_t v) { return pcg(v); }
void ref_pcg4d(const uint32_t v[4], uint32_t out[4]) {
    auto r = pcg4d(make_uint4(v[0], v[1], v[2], v[3]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
float ref_lcg(uint32_t *state) {
    UInt s = *state;
    auto u = lcg(s);
    *state = s;
    return u;
}
void ref_pcg32_seed(uint64_t seq_index, uint64_t *state, uint64_t *inc) {
    PCG32 rng{U64{seq_index}};
    *state = uint2_to_u64(rng.state().bits()), *inc = uint2_to_u64(rng.inc().bits());
}
uint32_t ref_pcg32_next(uint64_t *state, uint64_t *inc) {
    PCG32 rng{U64{*state}, U64{*inc}};
    auto v = rng.uniform_uint();
    *state = uint2_to_u64(rng.state().bits()), *inc = uint2_to_u64(rng.inc().bits());
    return v;
}
void ref_create_alias_table(const float *values, uint32_t n, float *prob, uint32_t *alias, float *pdf) {
    auto [table, p] = create_alias_table(luisa::span<const float>{values, n});
    for (auto i = 0u; i < n; i++) { prob[i] = table[i].prob, alias[i] = table[i].alias, pdf[i] = p[i]; }
}
void ref_sample_alias_table(const float *prob, const uint32_t *alias, uint32_t n, float u, uint32_t *index, float *u_remapped) {
    std::vector<AliasEntry> table(n);
    for (auto i = 0u; i < n; i++) { table[i] = {prob[i], alias[i]}; }
    Constant<AliasEntry> t{table.data(), table.size()};
    auto [i, uu] = sample_alias_table(t, n, u);
    *index = i, *u_remapped = uu;
}
void ref_sample_uniform_triangle(float ux, float uy, float out[3]) {
    auto b = sample_uniform_triangle(make_float2(ux, uy));
    out[0] = b.x, out[1] = b.y, out[2] = b.z;
}
void ref_sample_cosine_hemisphere(float ux, float uy, float out[3]) {
    auto w = sample_cosine_hemisphere(make_float2(ux, uy));
    out[0] = w.x, out[1] = w.y, out[2] = w.z;
}
void ref_sample_uniform_sphere(float ux, float uy, float out[3]) {
    auto w = sample_uniform_sphere(make_float2(ux, uy));
    out[0] = w.x, out[1] = w.y, out[2] = w.z;
}
void ref_sample_uniform_cone(float ux, float uy, float cos_theta_max, float out[3]) {
    auto w = sample_uniform_cone(make_float2(ux, uy), cos_theta_max);
    out[0] = w.x, out[1] = w.y, out[2] = w.z;
}
float ref_balance_heuristic(float a, float b) { return balance_heuristic(a, b); }
float ref_power_heuristic(float a, float b) { return power_heuristic(a, b); }
float ref_fresnel_dielectric(float cos_i, float eta_i, float eta_t) { return fresnel_dielectric(cos_i, eta_i, eta_t); }
void ref_fresnel_conductor(float cos_i, float eta_i, const float eta_t[3], const float k[3], float out[3]) {
    SampledSpectrum e{3u}, kk{3u};
    for (auto i = 0u; i < 3u; i++) { e[i] = eta_t[i], kk[i] = k[i]; }
    auto f = fresnel_conductor(cos_i, eta_i, e, kk);
    for (auto i = 0u; i < 3u; i++) { out[i] = f[i]; }
}
float ref_fresnel_dielectric_integral(float eta) { return fresnel_dielectric_integral(eta); }
int ref_refract(const float wi[3], const float n[3], float eta, float wt_out[3]) {
    auto wt = make_float3();
    auto ok = refract(make_float3(wi[0], wi[1], wi[2]), make_float3(n[0], n[1], n[2]), eta, &wt);
    wt_out[0] = wt.x, wt_out[1] = wt.y, wt_out[2] = wt.z;
    return ok ? 1 : 0;
}
// TrowbridgeReitzDistribution (src/util/scattering.cpp:145-237): out = D(wh), Lambda(wo), G(wo, wi), pdf(wo, wh)
void ref_ggx(const float alpha[2], const float wo[3], const float wi[3], const float wh[3], float out[4]) {
    TrowbridgeReitzDistribution d{make_float2(alpha[0], alpha[1])};
    auto o = make_float3(wo[0], wo[1], wo[2]), i = make_float3(wi[0], wi[1], wi[2]), h = make_float3(wh[0], wh[1], wh[2]);
    out[0] = d.D(h), out[1] = d.Lambda(o), out[2] = d.G(o, i), out[3] = d.pdf(o, h);
}
void ref_ggx_sample_wh(const float alpha[2], const float wo[3], float ux, float uy, float out[3]) {
    TrowbridgeReitzDistribution d{make_float2(alpha[0], alpha[1])};
    auto h = d.sample_wh(make_float3(wo[0], wo[1], wo[2]), make_float2(ux, uy));
    out[0] = h.x, out[1] = h.y, out[2] = h.z;
}
// Frame::make(n) / Frame::make(n, s) (src/util/frame.cpp:21-34): out = s[3], t[3], n[3]
void ref_frame_make(const float n[3], const float *s_or_null, float out[9]) {
    auto nn = make_float3(n[0], n[1], n[2]);
    auto f = s_or_null ? Frame::make(nn, make_float3(s_or_null[0], s_or_null[1], s_or_null[2])) : Frame::make(nn);
    out[0] = f.s().x, out[1] = f.s().y, out[2] = f.s().z, out[3] = f.t().x, out[4] = f.t().y, out[5] = f.t().z;
    out[6] = f.n().x, out[7] = f.n().y, out[8] = f.n().z;
}
void ref_clamp_shading_normal(const float ns[3], const float ng[3], const float w[3], float out[3]) {
    auto r = clamp_shading_normal(make_float3(ns[0], ns[1], ns[2]), make_float3(ng[0], ng[1], ng[2]), make_float3(w[0], w[1], w[2]));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void ref_encode_handle(uint32_t buffer_base, uint32_t flags, uint32_t surface_tag, uint32_t light_tag, uint32_t medium_tag,
                       uint32_t tri_count, float shadow_term, float isect_offset, uint32_t out[4]) {
    auto h = Shape::Handle::encode(buffer_base, flags, surface_tag, light_tag, medium_tag, tri_count, shadow_term, isect_offset);
    out[0] = h.x, out[1] = h.y, out[2] = h.z, out[3] = h.w;
}

}// extern "C"
