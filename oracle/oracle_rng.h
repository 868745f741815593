// oracle_rng.h — hashes, generators and samplers of the reference, restated on the CPU.
// TEST INFRASTRUCTURE ONLY (see oracle_math.h).
#pragma once
#include "oracle_math.h"

namespace oracle {

// ---- src/util/rng.cpp:12-69 (xxhash32 avalanche variants over 1..4 words)
constexpr uint32_t PRIME32_2 = 2246822519u, PRIME32_3 = 3266489917u;
constexpr uint32_t PRIME32_4 = 668265263u, PRIME32_5 = 374761393u;
inline uint32_t rotl17(uint32_t h) { return (h << 17u) | (h >> 15u); }
inline uint32_t xxhash32_finish(uint32_t h32) {
    h32 = PRIME32_2 * (h32 ^ (h32 >> 15u));
    h32 = PRIME32_3 * (h32 ^ (h32 >> 13u));
    return h32 ^ (h32 >> 16u);
}
inline uint32_t xxhash32(uint32_t p) {// rng.cpp:12-23
    auto h32 = p + PRIME32_5;
    h32 = PRIME32_4 * rotl17(h32);
    return xxhash32_finish(h32);
}
inline uint32_t xxhash32(uint32_t x, uint32_t y) {// rng.cpp:25-36
    auto h32 = y + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xxhash32_finish(h32);
}
inline uint32_t xxhash32(uint32_t x, uint32_t y, uint32_t z) {// rng.cpp:38-51
    auto h32 = z + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += y * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xxhash32_finish(h32);
}
inline uint32_t xxhash32(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {// rng.cpp:53-69
    auto h32 = w + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += y * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += z * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xxhash32_finish(h32);
}

inline float uniform_uint_to_float(uint32_t u) {// rng.cpp:128-130
    return std::min(one_minus_epsilon, static_cast<float>(u) * 0x1p-32f);
}
inline float lcg(uint32_t &state) {// rng.cpp:132-140
    state = 1664525u * state + 1013904223u;
    return uniform_uint_to_float(state);
}

// ---- PCG32 (rng.cpp:142-176; the reference emulates u64 with two u32, util/u64.h)
struct PCG32 {
    static constexpr uint64_t default_state = 0x853c49e6748fea9bull;
    static constexpr uint64_t default_stream = 0xda3e39cb94b95bdbull;
    static constexpr uint64_t mult = 0x5851f42d4c957f2dull;
    uint64_t state{default_state}, inc{default_stream};
    PCG32() = default;
    explicit PCG32(uint64_t seq_index) { set_sequence(seq_index); }
    uint32_t uniform_uint() {
        auto oldstate = state;
        state = oldstate * mult + inc;
        auto xorshifted = static_cast<uint32_t>(((oldstate >> 18u) ^ oldstate) >> 27u);
        auto rot = static_cast<uint32_t>(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    void set_sequence(uint64_t init_seq) {
        state = 0u;
        inc = (init_seq << 1u) | 1u;
        (void)uniform_uint();
        state += default_state;
        (void)uniform_uint();
    }
    float uniform_float() { return uniform_uint_to_float(uniform_uint()); }
};

// ---- sampling helpers, src/util/sampling.{h,cpp}
inline float2 sample_uniform_disk_concentric(float2 u_in) {// sampling.cpp:13-22
    float2 u{u_in.x * 2.0f - 1.0f, u_in.y * 2.0f - 1.0f};
    auto p = std::abs(u.x) > std::abs(u.y);
    auto r = p ? u.x : u.y;
    auto theta = p ? pi_over_four * (u.y / u.x) : pi_over_two - pi_over_four * (u.x / u.y);
    return {r * std::cos(theta), r * std::sin(theta)};
}
inline float3 sample_cosine_hemisphere(float2 u) {// sampling.cpp:24-31
    auto d = sample_uniform_disk_concentric(u);
    auto z = std::sqrt(std::max(1.0f - d.x * d.x - d.y * d.y, 0.0f));
    return {d.x, d.y, z};
}
inline float3 sample_uniform_triangle(float2 u) {// sampling.cpp:89-98
    float2 uv = u.x < u.y ? float2{0.5f * u.x, -0.5f * u.x + u.y} : float2{-0.5f * u.y + u.x, 0.5f * u.y};
    return {uv.x, uv.y, 1.0f - uv.x - uv.y};
}
inline float3 sample_uniform_sphere(float2 u) {// sampling.cpp:100-108
    auto z = 1.0f - 2.0f * u.x;
    auto r = std::sqrt(std::max(1.0f - z * z, 0.0f));
    auto phi = 2.0f * pi * u.y;
    return {r * std::cos(phi), r * std::sin(phi), z};
}
constexpr float uniform_sphere_pdf = inv_pi * 0.25f;// sampling.h:24
inline float balance_heuristic(float f_pdf, float g_pdf) {// sampling.cpp:133-140,153-155
    auto sum_f = 1.f * f_pdf;
    auto sum = sum_f + 1.f * g_pdf;
    return sum == 0.0f ? 0.0f : sum_f / sum;
}

struct AliasSample {
    uint32_t index;
    float u;
};
// sample_alias_table, src/util/sampling.h:38-66 (both overloads share the arithmetic)
template<typename ProbAt, typename AliasAt>
inline AliasSample sample_alias_table(ProbAt prob_at, AliasAt alias_at, uint32_t n, float u_in) {
    auto u = u_in * static_cast<float>(n);
    auto i = std::min(static_cast<uint32_t>(std::max(u, 0.f)), n - 1u);// clamp(cast<uint>(u), 0, n - 1)
    auto u_remapped = fract(u);
    auto prob = prob_at(i);
    auto index = u_remapped < prob ? i : alias_at(i);
    auto uu = u_remapped < prob ? u_remapped / prob : (u_remapped - prob) / (1.0f - prob);
    return {index, uu};
}

}// namespace oracle
