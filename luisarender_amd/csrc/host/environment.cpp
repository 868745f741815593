// environment.cpp — importance tables of image-based Spherical environments.
//
// Replaces the host half of Spherical::build (src/environments/spherical.cpp:144-235): a 2048 x 1024
// "scale map" (Gaussian-filtered luminance * sin(theta)), optional MIS compensation, one conditional
// alias table per row plus the marginal table over rows, and the per-texel pdf.  The reference runs
// the filter as a one-off device kernel; here it is a one-off multithreaded host pass so that the
// tables are part of the flattened scene (lr_environment::alias / ::pdf) and the HIP kernel and the
// CPU oracle read the same bits.
#include <algorithm>
#include <cmath>
#include <thread>

#include "scene.h"

namespace lr {

namespace {

constexpr auto kMapWidth = 2048u, kMapHeight = 1024u;// Spherical::sample_map_size, spherical.cpp:22
constexpr auto kPi = 3.14159265358979323846f;

struct float4v {
    float x, y, z, w;
};

// host evaluation of a texture node: constant.cpp:73-79, image.cpp:132-168 (bilinear, level 0),
// checkerboard.cpp — the same conventions the device uses (dev_shade.h texture_eval)
class TextureEvaluator {
    const SceneData &_scene;

    float4v _fetch(const lr_texture &t, int x, int y) const {
        auto zero = false;
        auto wrap = [&](int v, int n) {
            switch (t.address) {
                case LR_TEX_ADDR_EDGE: return std::clamp(v, 0, n - 1);
                case LR_TEX_ADDR_MIRROR: {
                    auto period = 2 * n;
                    auto m = ((v % period) + period) % period;
                    return m < n ? m : period - 1 - m;
                }
                case LR_TEX_ADDR_ZERO:
                    if (v < 0 || v >= n) { zero = true; return 0; }
                    return v;
                default: return ((v % n) + n) % n;
            }
        };
        auto xx = wrap(x, static_cast<int>(t.width)), yy = wrap(y, static_cast<int>(t.height));
        if (zero) { return {0.f, 0.f, 0.f, 0.f}; }
        auto p = _scene.texels.data() + (t.texel_offset + static_cast<uint64_t>(yy) * t.width + static_cast<uint64_t>(xx)) * 4u;
        return {p[0], p[1], p[2], p[3]};
    }

public:
    explicit TextureEvaluator(const SceneData &scene) : _scene{scene} {}

    float4v operator()(int32_t id, float u, float v) const {
        auto &t = _scene.textures[static_cast<size_t>(id)];
        if (t.kind == LR_TEX_CONSTANT) { return {t.v[0], t.v[1], t.v[2], t.v[3]}; }
        if (t.kind == LR_TEX_CHECKERBOARD) {
            auto parity = (static_cast<int>(std::floor(u * t.checker_scale)) + static_cast<int>(std::floor(v * t.checker_scale))) & 1;
            auto child = t.child[parity ? 1 : 0];
            if (child < 0) { return parity ? float4v{0.f, 0.f, 0.f, 1.f} : float4v{1.f, 1.f, 1.f, 1.f}; }
            return (*this)(child, u, v);
        }
        auto s = u * t.uv_scale[0] + t.uv_offset[0], r = v * t.uv_scale[1] + t.uv_offset[1];
        float4v c;
        if (t.filter == LR_TEX_FILTER_POINT) {
            c = _fetch(t, static_cast<int>(std::floor(s * static_cast<float>(t.width))), static_cast<int>(std::floor(r * static_cast<float>(t.height))));
        } else {
            auto fx = s * static_cast<float>(t.width) - 0.5f, fy = r * static_cast<float>(t.height) - 0.5f;
            auto x0 = std::floor(fx), y0 = std::floor(fy);
            auto tx = fx - x0, ty = fy - y0;
            auto ix = static_cast<int>(x0), iy = static_cast<int>(y0);
            auto c00 = _fetch(t, ix, iy), c10 = _fetch(t, ix + 1, iy), c01 = _fetch(t, ix, iy + 1), c11 = _fetch(t, ix + 1, iy + 1);
            auto mix = [&](float a, float b, float cc, float d) { return (a * (1.f - tx) + b * tx) * (1.f - ty) + (cc * (1.f - tx) + d * tx) * ty; };
            c = {mix(c00.x, c10.x, c01.x, c11.x), mix(c00.y, c10.y, c01.y, c11.y), mix(c00.z, c10.z, c01.z, c11.z), mix(c00.w, c10.w, c01.w, c11.w)};
        }
        auto decode = [&](float x, int ch) {
            if (t.encoding == LR_TEX_ENC_SRGB) {
                x = x <= 0.04045f ? x * (1.0f / 12.92f) : std::pow((x + 0.055f) * (1.0f / 1.055f), 2.4f);
            } else if (t.encoding == LR_TEX_ENC_GAMMA) {
                x = std::pow(x, t.gamma[std::min(ch, 2)]);
            }
            return t.scale[ch] * x;
        };
        return {decode(c.x, 0), decode(c.y, 1), decode(c.z, 2), decode(c.w, 3)};
    }
};

}// namespace

// Spherical::build, spherical.cpp:144-235.  The filter taps sit on a grid of 1/8 map texel
// ((pixel + .5 + k/8) / size, k = -8..8), so each band of map rows evaluates the texture once per grid
// point (17 grid rows resident) and then accumulates the 17 x 17 taps in the reference's dy-outer /
// dx-inner order.
void build_environment_tables(const SceneData &scene, lr_environment &env, std::vector<lr_alias_entry> &env_alias, std::vector<float> &env_pdf) {
    env_alias.clear(), env_pdf.clear();
    env.map_width = env.map_height = 0u;
    if (env.kind != LR_ENV_SPHERICAL || env.emission_tex < 0) { return; }
    if (scene.textures[static_cast<size_t>(env.emission_tex)].kind == LR_TEX_CONSTANT) { return; }// uniform sphere sampling
    constexpr auto W = kMapWidth, H = kMapHeight;
    constexpr auto pixel_count = W * H;
    constexpr auto n = 8;              // ceil(filter_radius / filter_step) = ceil(1 / .125)
    constexpr auto taps = 2 * n + 1;   // 17
    constexpr auto fine_w = W * 8u + taps;// grid columns g = 8 x + 4 + dx, dx in [-8, 8]  ->  index g + 4
    float weight[taps][taps];
    auto sum_weight = 0.f;
    for (auto dy = -n; dy <= n; dy++) {
        for (auto dx = -n; dx <= n; dx++) {
            auto ox = static_cast<float>(dx) * .125f, oy = static_cast<float>(dy) * .125f;
            weight[dy + n][dx + n] = std::exp(-4.f * (ox * ox + oy * oy));// gaussian kernel with an approximate radius of 1
            sum_weight += weight[dy + n][dx + n];
        }
    }
    std::vector<float> scale_map(pixel_count);
    TextureEvaluator texture{scene};
    auto emission = env.emission_tex;
    auto band = [&](uint32_t y_begin, uint32_t y_end) {
        std::vector<float> ring(static_cast<size_t>(taps) * fine_w);
        auto fill_row = [&](int gy) {// fine row gy = 8 y + 4 + dy
            auto row = ring.data() + static_cast<size_t>(((gy % taps) + taps) % taps) * fine_w;
            auto v = (static_cast<float>(gy) * .125f) / static_cast<float>(H);
            auto sin_theta = std::sin(v * kPi);
            for (auto i = 0u; i < fine_w; i++) {
                auto gx = static_cast<int>(i) - 4;
                auto u = (static_cast<float>(gx) * .125f) / static_cast<float>(W);
                auto c = texture(emission, u, v);
                // evaluate_illuminant_spectrum(...).strength with the sRGB spectrum (srgb.cpp:48-54)
                auto r = std::max(c.x, 0.f), g = std::max(c.y, 0.f), b = std::max(c.z, 0.f);
                auto strength = 0.212671f * r + 0.715160f * g + 0.072169f * b;
                row[i] = std::min(strength * sin_theta, 1e8f);
            }
        };
        auto filled_to = static_cast<int>(y_begin) * 8 + 4 - n - 1;
        for (auto y = y_begin; y < y_end; y++) {
            auto g0 = static_cast<int>(y) * 8 + 4 - n, g1 = static_cast<int>(y) * 8 + 4 + n;
            for (auto gy = std::max(filled_to + 1, g0); gy <= g1; gy++) { fill_row(gy); }
            filled_to = g1;
            for (auto x = 0u; x < W; x++) {
                auto sum_scale = 0.f;
                for (auto dy = -n; dy <= n; dy++) {
                    auto gy = static_cast<int>(y) * 8 + 4 + dy;
                    auto row = ring.data() + static_cast<size_t>(((gy % taps) + taps) % taps) * fine_w + x * 8u + 4u + 4u;
                    for (auto dx = -n; dx <= n; dx++) { sum_scale += weight[dy + n][dx + n] * row[dx]; }
                }
                scale_map[static_cast<size_t>(y) * W + x] = sum_scale / sum_weight;
            }
        }
    };
    {
        auto workers = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
        std::vector<std::thread> pool;
        auto rows = (H + workers - 1u) / workers;
        for (auto w = 0u; w < workers; w++) {
            auto b = w * rows, e = std::min(H, b + rows);
            if (b < e) { pool.emplace_back(band, b, e); }
        }
        for (auto &t : pool) { t.join(); }
    }
    if (env.compensate_mis) {// spherical.cpp:187-192
        auto sum_scale = 0.;
        for (auto s : scale_map) { sum_scale += s; }
        auto average_scale = static_cast<float>(sum_scale / pixel_count);
        for (auto &s : scale_map) { s = std::max(s - average_scale, 0.f); }
    }
    std::vector<float> row_averages(H);
    env_pdf.resize(pixel_count);
    env_alias.resize(H + pixel_count);
    {// conditional tables, one per row (independent -> threaded)
        auto workers = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
        std::vector<std::thread> pool;
        auto rows = (H + workers - 1u) / workers;
        for (auto w = 0u; w < workers; w++) {
            auto b = w * rows, e = std::min(H, b + rows);
            if (b >= e) { continue; }
            pool.emplace_back([&, b, e] {
                std::vector<lr_alias_entry> table;
                std::vector<float> pdf;
                for (auto i = b; i < e; i++) {
                    auto values = scale_map.data() + static_cast<size_t>(i) * W;
                    auto sum = 0.;
                    for (auto x = 0u; x < W; x++) { sum += values[x]; }
                    row_averages[i] = static_cast<float>(sum * (1.0 / W));
                    create_alias_table(values, W, table, pdf);
                    std::copy_n(pdf.data(), W, env_pdf.data() + static_cast<size_t>(i) * W);
                    std::copy_n(table.data(), W, env_alias.data() + H + static_cast<size_t>(i) * W);
                }
            });
        }
        for (auto &t : pool) { t.join(); }
    }
    std::vector<lr_alias_entry> table;
    std::vector<float> pdf;
    create_alias_table(row_averages.data(), H, table, pdf);// marginal over rows
    std::copy_n(table.data(), H, env_alias.data());
    for (auto y = 0u; y < H; y++) {
        auto scale = static_cast<float>(pdf[y] * pixel_count);
        for (auto x = 0u; x < W; x++) { env_pdf[static_cast<size_t>(y) * W + x] *= scale; }
    }
    env.map_width = W, env.map_height = H;
}

}// namespace lr
