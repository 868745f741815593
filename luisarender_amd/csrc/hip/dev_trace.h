// dev_trace.h — BVH4 traversal + triangle intersection for gfx950 (no ray-tracing hardware:
// this is VALU + memory, one ray per lane, 64 rays per wavefront).
//
// Replaces the reference's Accel::intersect / intersect_any (LuisaCompute, absent submodule;
// call sites src/base/geometry.cpp:218-279) with the published semantics: closest hit returns
// {inst, prim, bary} (src/base/geometry.h:16-28), any-hit returns a bool.
//
// Shape of the loop:
//   * one 128-byte node per step, read as eight dwordx4 loads (node = one cache line);
//   * four slab tests, hits ordered near->far with a 5-comparator network on packed
//     (t | slot) integer keys (t >= 0, so float order == unsigned order);
//   * traversal stack in LDS, [entry][lane] interleaved so a wave's push/pop is conflict-free
//     (ds_write_b32 / ds_read_b32 at consecutive banks); entries beyond kStackLds spill to a
//     per-thread HBM area;
//   * `trace_pair` runs the shadow ray and the next closest-hit ray of a path back-to-back in
//     ONE loop, so a wavefront pays max_lane(steps_shadow + steps_closest) instead of
//     max_lane(steps_shadow) + max_lane(steps_closest).
#pragma once
#include "dev_scene.h"

namespace lrd {

constexpr uint32_t kBlockThreads = 256u;
constexpr uint32_t kStackLds = 24u;      // entries per lane kept in LDS (24 KB per block)
constexpr uint32_t kSpillEntries = 72u;  // HBM overflow entries per lane
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kInvalid = 0xffffffffu;

struct Ray {
    f3 o;
    float t_min;
    f3 d;
    float t_max;
};

struct HitRecord {
    uint32_t inst, prim;
    float u, v;
};

struct TraceStats {
    uint32_t nodes, tris;
};

struct TraversalStack {
    uint32_t *lds;      // &stack[0][tid]; stride kBlockThreads
    uint32_t *spill;    // &spill[0][gtid]; stride total_threads
    uint32_t spill_stride;
    LR_D void push(uint32_t sp, uint32_t v) const {
        if (sp < kStackLds) { lds[sp * kBlockThreads] = v; }
        else { spill[static_cast<size_t>(sp - kStackLds) * spill_stride] = v; }
    }
    LR_D uint32_t pop(uint32_t sp) const {
        return sp < kStackLds ? lds[sp * kBlockThreads] : spill[static_cast<size_t>(sp - kStackLds) * spill_stride];
    }
};

LR_D void cswap(uint32_t &a, uint32_t &b) {
    auto lo = min(a, b), hi = max(a, b);
    a = lo, b = hi;
}

// Traces `shadow` (any-hit, if has_shadow) and then `closest` (closest-hit, if has_closest) for this
// lane.  Returns occlusion of the shadow ray in `occluded`, the closest hit in `hit` (inst == kInvalid
// on miss).  COUNT enables the per-ray node/triangle counters.
template<bool COUNT>
LR_D void trace_pair(const DScene &scene, const TraversalStack &stack, bool has_shadow, const Ray &shadow,
                     bool has_closest, const Ray &closest, bool &occluded, HitRecord &hit, TraceStats &stats) {
    occluded = false;
    hit.inst = kInvalid, hit.prim = kInvalid, hit.u = 0.f, hit.v = 0.f;
    auto phase_shadow = has_shadow;
    if (!has_shadow && !has_closest) { return; }
    f3 o = phase_shadow ? shadow.o : closest.o;
    f3 d = phase_shadow ? shadow.d : closest.d;
    auto t_min = phase_shadow ? shadow.t_min : closest.t_min;
    auto t_max = phase_shadow ? shadow.t_max : closest.t_max;
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    auto nodes = reinterpret_cast<const float4 *>(scene.nodes);
    auto tris = reinterpret_cast<const float4 *>(scene.bvh_tris);
    uint32_t sp = 0u;
    uint32_t cur = 0u;// root
    for (;;) {
        if (cur != kInvalid && !(cur & kLeafFlag)) {
            // ---- inner node: 8 x dwordx4
            auto base = nodes + static_cast<size_t>(cur) * 8u;
            auto lox = base[0], loy = base[1], loz = base[2];
            auto hix = base[3], hiy = base[4], hiz = base[5];
            auto ch = reinterpret_cast<const uint4 *>(base)[6];
            if (COUNT) { stats.nodes++; }
            uint32_t key[4];
#define LR_SLAB(i, LX, LY, LZ, HX, HY, HZ, C)                                                        \
    {                                                                                                \
        auto t0x = (LX - o.x) * inv.x, t1x = (HX - o.x) * inv.x;                                     \
        auto t0y = (LY - o.y) * inv.y, t1y = (HY - o.y) * inv.y;                                     \
        auto t0z = (LZ - o.z) * inv.z, t1z = (HZ - o.z) * inv.z;                                     \
        auto tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), t_min));     \
        auto tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), t_max));     \
        auto h = (tn <= tf * 1.0000004f) && (C != kInvalid);                                         \
        key[i] = h ? ((__float_as_uint(tn) & 0xfffffffcu) | i##u) : kInvalid;                        \
    }
            LR_SLAB(0, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x)
            LR_SLAB(1, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y)
            LR_SLAB(2, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z)
            LR_SLAB(3, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w)
#undef LR_SLAB
            // sort ascending: 5-comparator network
            cswap(key[0], key[1]);
            cswap(key[2], key[3]);
            cswap(key[0], key[2]);
            cswap(key[1], key[3]);
            cswap(key[1], key[2]);
            auto ref_of = [&](uint32_t k) {
                auto slot = k & 3u;
                return slot == 0u ? ch.x : (slot == 1u ? ch.y : (slot == 2u ? ch.z : ch.w));
            };
            // push far -> near so that the nearest is popped first; keep the nearest in `cur`
            if (key[3] != kInvalid) { stack.push(sp++, ref_of(key[3])); }
            if (key[2] != kInvalid) { stack.push(sp++, ref_of(key[2])); }
            if (key[1] != kInvalid) { stack.push(sp++, ref_of(key[1])); }
            cur = key[0] != kInvalid ? ref_of(key[0]) : kInvalid;
            if (cur != kInvalid) { continue; }
        } else if (cur != kInvalid) {
            // ---- leaf: Moeller-Trumbore on pre-transformed triangles (3 x dwordx4 each)
            auto first = cur & ((1u << 27u) - 1u);
            auto count = ((cur >> 27u) & 15u) + 1u;
            auto found = false;
            for (auto k = 0u; k < count; k++) {
                auto tb = tris + static_cast<size_t>(first + k) * 3u;
                auto a = tb[0], b = tb[1], c = tb[2];
                if (COUNT) { stats.tris++; }
                auto flags = __float_as_uint(c.w);
                f3 p0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
                auto pvec = cross(d, e2);
                auto det = dot(e1, pvec);
                auto inv_det = 1.0f / det;
                auto tvec = o - p0;
                auto u = dot(tvec, pvec) * inv_det;
                auto qvec = cross(tvec, e1);
                auto v = dot(d, qvec) * inv_det;
                auto t = dot(e2, qvec) * inv_det;
                auto ok = det != 0.f && u >= 0.f && v >= 0.f && u + v <= 1.f && t > t_min && t < t_max && (flags & 1u);
                if (ok) {
                    t_max = t;
                    found = true;
                    if (!phase_shadow) {
                        hit.inst = __float_as_uint(a.w), hit.prim = __float_as_uint(b.w);
                        hit.u = u, hit.v = v;
                    }
                }
            }
            if (phase_shadow && found) {
                occluded = true;
                sp = 0u;// any-hit: drop the rest of the stack
            }
        }
        // ---- pop, or switch from the shadow ray to the closest-hit ray
        if (sp > 0u) {
            cur = stack.pop(--sp);
            continue;
        }
        if (phase_shadow && has_closest) {
            phase_shadow = false;
            o = closest.o, d = closest.d;
            t_min = closest.t_min, t_max = closest.t_max;
            inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            cur = 0u;
            continue;
        }
        break;
    }
}

}// namespace lrd
