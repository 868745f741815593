// dev_heavy.h — closure loading shared by the megakernel and its out-of-line "heavy" closure path.
//
// The reference JIT inlines every closure a scene uses into one kernel.  On gfx950 that costs the whole kernel its
// registers: with the Disney, Mix and Layered interpreters inlined into the shading block the <everything> variant
// spilled ~1000 VGPRs and ran 5x slower than the lean one on a scene where 85 % of the hits are Matte / Plastic /
// Metal / Glass.  So the megakernel keeps the five basic closures inline (closure_evaluate<false> / closure_sample<false>)
// and reaches Disney / Mix / Layered surfaces through two REAL calls, heavy_evaluate and heavy_sample, which take
// everything they need in a HeavyCtx record (scratch memory) and have their own register allocation.
//
//   Surface::Closure::evaluate / sample              src/base/surface.cpp:35-68
//   MixSurfaceClosure                                src/surfaces/mix.cpp:82-212
//   LayeredSurfaceInstance::populate_closure         src/surfaces/layered.cpp:478-500
#pragma once
#include "dev_layered.h"

namespace lrd {

// closure record + shading frame of surface `t` on top of frame `base`: NormalMapWrapper (surface.h:236-254) and
// per-hit texture resolution for "dynamic" closures (the constant ones were folded at upload by the same resolve_closure)
#ifndef LR_LOBE_IMAGE_PASSES
#define LR_LOBE_IMAGE_PASSES 4// looked-up slots of a surface kept in registers (more than these: looked up where the closure's code asks)
#endif
// LR_LOBE_FORM 2: the first two looked-up slots of every lane in two WAVE-LEVEL passes ahead of the kinds' own code -- a batch whose lanes
// stand on Disney, Plastic and Matte surfaces with two image maps each makes two lookups, not five one kind after the other.  Camera class
// (<4116>) 1075 -> 1125 Msamples/s at 64 spp, films bit-identical; but the six registers the results wait in cost the kernels whose scenes
// hold no image at all their allocation: C2 <4096> 1075 -> 1032, the kitchen class' wavefront passes 585 -> 577 (profiles/r06i_lobe_forms.txt).
// So: in the lean kernels WITH the Disney closure (the variants textured scenes of the camera class' kind take), form 1 elsewhere.
// Measured with it and not kept: the closure resolution as one out-of-line function with the passes inside (its 41 words of result through
// memory, 340 scratch instructions of callee saves per call: every configuration 5 - 12 % slower, r06g), and touching the texels ahead of
// the lookups instead of keeping results (-7 %, r06i).
// LR_LOBE_FORM 3 (round 6): ONE lookup of every lane at one place ahead of everything else -- its normal map if it has one, else its first
// looked-up slot (DSurface::first_lookup, asked for beside the closure record) -- three registers instead of form 2's six.  The lean passes of
// wavefront mode: the kitchen class' floor / Oren-Nayar / plastic albedo and the bumpy plastic's normal map were four lookups one kind after
// the other; 592 -> 607 Msamples/s at 512 spp, films bit-identical (profiles/r06q_lobe_form3.txt).
#ifndef LR_LOBE_FORM
#if defined(LR_VARIANT) && ((LR_VARIANT) & 16) && !((LR_VARIANT) & (96 | 256))
#define LR_LOBE_FORM 2
#elif defined(LR_VARIANT) && ((LR_VARIANT) & 1024) && !((LR_VARIANT) & (96 | 256))
#define LR_LOBE_FORM 3
#else
#define LR_LOBE_FORM 1
#endif
#endif
// closure record + shading frame of surface `t` on top of frame `base`: NormalMapWrapper (surface.h:236-254) and
// per-hit texture resolution for "dynamic" closures (the constant ones were folded at upload by the same resolve_closure)
LR_D void load_lobe(const LobeTables &tb, f2 uv, f3 ng, f3 wo, uint32_t t, const Frame &base, DClosure &c, Frame &fr, float eta_i = 1.f) {
#if LR_LOBE_FORM == 3
    const auto first_id = tb.surfaces[t].first_lookup;// (asked for beside the closure record, not behind it)
#endif
    c = tb.closures[t];
    fr = base;
    if (c.dynamic || eta_i != 1.f) {// (eta_i != 1: the bottom of a Layered surface under a refractive top)
        auto &rec = tb.surfaces[t];
        auto &raw = rec.raw;
#if LR_LOBE_FORM == 3
        // the lookup of a slot whose texture is NOT constant: the out-of-line lambda of round 3 (dev_math.h: LR_TEX_LAMBDA), now asked for
        // those slots only -- a constant slot is a plain load from the surface's own record (dev_scene.h: DSurface), independent of the
        // other slots' and issued with them, where it used to be a call and a dependent round trip through the texture table each
#if LR_TEX_BY_VALUE
        const auto lookup = [&](int32_t id) { return texture_eval_slot(tb.textures, tb.texels, id, uv.x, uv.y); };
#else
        const auto lookup = [&](int32_t id) LR_TEX_LAMBDA_ATTR { return texture_eval_tables(tb.textures, tb.texels, id, uv); };
#endif
        // ONE lookup of every lane at ONE place ahead of everything else: the lane's normal map if it has one, else its first looked-up slot.
        // (The kitchen class' lean hits: floor and Oren-Nayar albedo, the plastic's albedo, the bumpy plastic's normal map -- four lookups one
        // kind after the other become one, and the plastic's roughness.)
        const auto first_is_normal = raw.normal_tex >= 0;
        float4 looked_first = make_float4(0.f, 0.f, 0.f, 0.f);
        if (first_id >= 0) { looked_first = lookup(first_id); }
        if (first_is_normal) {
            auto v = looked_first;
#else
        if (raw.normal_tex >= 0) {
            auto v = texture_eval_tables(tb.textures, tb.texels, raw.normal_tex, uv);
#endif
            auto n_local = mk3(2.f * v.x - 1.f, 2.f * v.y - 1.f, 2.f * v.z - 1.f);
            if (raw.normal_strength != 1.f) { n_local = n_local * mk3(raw.normal_strength, raw.normal_strength, 1.f); }
            auto normal = to_world(base, n_local);
            fr = frame_from_normal_tangent(clamp_shading_normal(normal, ng, wo), base.s);
        }
        auto dyn = c.dynamic;
#if LR_LOBE_FORM != 3
        // the lookup of a slot whose texture is NOT constant: the out-of-line lambda of round 3 (dev_math.h: LR_TEX_LAMBDA), now asked for
        // those slots only -- a constant slot is a plain load from the surface's own record (dev_scene.h: DSurface), independent of the
        // other slots' and issued with them, where it used to be a call and a dependent round trip through the texture table each
#if LR_TEX_BY_VALUE
        const auto lookup = [&](int32_t id) { return texture_eval_slot(tb.textures, tb.texels, id, uv.x, uv.y); };
#else
        const auto lookup = [&](int32_t id) LR_TEX_LAMBDA_ATTR { return texture_eval_tables(tb.textures, tb.texels, id, uv); };
#endif
#endif
        const auto mask = rec.dynamic_mask;
#if LR_LOBE_FORM == 2
        f3 looked0 = mk3(0.f), looked1 = mk3(0.f);
        {
            auto left = mask;
            if (lr_any(left != 0u)) {
                if (left != 0u) {
                    const auto slot = static_cast<uint32_t>(__builtin_ctz(left));
                    left &= left - 1u;
                    const auto v = lookup(raw.tex[slot]);
                    looked0 = mk3(v.x, v.y, v.z);
                }
                if (lr_any(left != 0u)) {
                    if (left != 0u) {
                        const auto slot = static_cast<uint32_t>(__builtin_ctz(left));
                        const auto v = lookup(raw.tex[slot]);
                        looked1 = mk3(v.x, v.y, v.z);
                    }
                }
            }
        }
#endif
        c = resolve_closure(
            raw,
            [&](int slot) {
#if LR_LOBE_FORM == 2
                if ((mask >> slot) & 1u) {
                    const auto rank = __builtin_popcount(mask & ((1u << slot) - 1u));
                    if (rank >= 2) { return lookup(raw.tex[slot]); }
                    const auto v = rank == 0 ? looked0 : looked1;
                    return make_float4(v.x, v.y, v.z, 0.f);
                }
#elif LR_LOBE_FORM == 3
                if ((mask >> slot) & 1u) {
                    if (!first_is_normal && (mask & ((1u << slot) - 1u)) == 0u) { return looked_first; }
                    return lookup(raw.tex[slot]);
                }
#else
                if ((mask >> slot) & 1u) { return lookup(raw.tex[slot]); }
#endif
                return *reinterpret_cast<const float4 *>(rec.value[slot]);
            },
            [&](int slot) { return (rec.channels[slot >> 3] >> ((slot & 7) * 4)) & 15u; }, eta_i);
        c.dynamic = dyn;
    }
}

struct HeavyCtx {// inputs of one heavy closure at one hit (lives in scratch: passed by pointer)
    LobeTables tb;
    f2 uv;
    f3 ng, p, wo;
    Frame shading;   // the surface's own (possibly normal-mapped) frame
    DClosure closure;// the surface's record (kind Disney / Mix / Layered)
};
struct HeavySample {
    BsdfSample bs;
    float eta;
    uint32_t has_eta;
};

LR_D BsdfEval mix_blend(const BsdfEval &a, const BsdfEval &b, float r) {// MixSurfaceClosure::_mix, mix.cpp:97-104
    auto t = 1.f - r;
    return BsdfEval{a.f + t * (b.f - a.f), lerp(a.pdf, b.pdf, t)};
}

// what the closures below a Mix / Layered node are populated from (populate_closure's arguments, mix.cpp:198-212)
struct MixCtx {
    LobeTables tb;
    f2 uv;
    f3 ng, p, wo_pop;// wo_pop: the hit's wo (normal-map clamping of the children); the evaluated direction is a parameter
    float eta_i;
};
LR_D MixCtx mix_ctx_of(const HeavyCtx &cx) { return MixCtx{cx.tb, cx.uv, cx.ng, cx.p, cx.wo, 1.f}; }

// ---- Mix trees.  The reference's Mix closure holds two arbitrary child closures (mix.cpp:82-212), Mix and Layered surfaces
// included, and a Layered surface holds two arbitrary interfaces (layered.cpp:195-253).  Device code has no recursion, so a Mix
// tree is walked with an EXPLICIT stack of its Mix nodes (post-order for evaluate and eta, down the chain of first children and back
// up for sample), in the reference's operation order; kMixMaxDepth levels under the root (lr_scene.h: LR_MIX_MAX_DEPTH, the loader
// rejects deeper trees; rounds 1-2 unrolled the recursion in templates and stopped at 3).  The leaves go through two out-of-line
// functions.  LV: the Layered levels still allowed below (dev_layered.h): a leaf may be a Layered surface while LV > 0, and its
// interfaces are interpreted with LV - 1 -- the loader bounds the Layered levels on a path (LR_LAYERED_MAX_LEVELS), which bounds the
// call graph: mix<2> -> layered<1> -> mix<1> / layered<0> -> mix<0> -> basic / Disney.
constexpr int kMixMaxDepth = LR_MIX_MAX_DEPTH;

struct MixLevel {// a Mix node on the path from the root of the tree to the node being worked on
    Frame frame;      // its own (possibly normal-mapped) frame: its children are loaded on top of it, its sides are validated with it
    float ratio;
    uint32_t child[2];
    uint32_t next;    // the child to visit next (0 / 1)
};

// (in the free-composition variants the eta walk is out of line: one copy per LV)
#if LR_NEST
#define LR_ETA_FN __device__ __noinline__
#else
#define LR_ETA_FN LR_D
#endif
// eta of the closure with tag `tag` under a node with frame `frame`: Surface::Closure::eta of a basic closure, MixSurfaceClosure::eta
// (mix.cpp:148-157: a's, b's, or their lerp by the ratio), LayeredSurfaceClosure::eta = its bottom's (layered.cpp:252; the Mix levels
// of a Layered surface's own interfaces count from zero, as in the loader: a fresh stack, one call per Layered level)
template<int LV>
LR_ETA_FN bool node_eta(const MixCtx &cx, uint32_t tag, const Frame &frame, float &eta) {
    MixLevel level[kMixMaxDepth + 1];
    float first_eta[kMixMaxDepth + 1];
    bool first_has[kMixMaxDepth + 1];
    auto sp = 0;
    auto cur_frame = frame;
    for (;;) {
        auto rec = &cx.tb.closures[tag];// (eta never comes from an image texture: the static record has it)
        bool has;
        float e = 1.f;
        if (rec->kind == LR_SURFACE_MIX && sp <= kMixMaxDepth) {// down
            DClosure node;
            Frame fr;
            load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, tag, cur_frame, node, fr, cx.eta_i);// (its ratio may be textured)
            level[sp] = MixLevel{fr, node.s0, {node.x[0], node.x[1]}, 0u};
            sp++;
            tag = node.x[0], cur_frame = fr;
            continue;
        }
        if (rec->kind == LR_SURFACE_LAYERED) {
            if constexpr (LV > 0) { has = node_eta<LV - 1>(cx, rec->x[1], cur_frame, e); }
            else { has = false; }// (the loader bounds the Layered levels: not reached)
        } else {
            has = closure_eta(*rec, e);
        }
        for (;;) {// up: a finished first child starts the second, a finished second child finishes the node
            if (sp == 0) { eta = e; return has; }
            auto &node = level[sp - 1];
            if (node.next == 0u) {
                first_has[sp - 1] = has, first_eta[sp - 1] = e;
                node.next = 1u, tag = node.child[1], cur_frame = node.frame;
                break;
            }
            const auto ha = first_has[sp - 1];
            const auto ea = first_eta[sp - 1];
            e = !ha ? e : (!has ? ea : lerp(e, ea, node.ratio));
            has = ha || has;
            sp--;
        }
    }
}
// MixSurfaceClosure::eta of a Mix node that is already loaded (the surface a ray hit)
template<int LV>
LR_D bool mix_eta(const MixCtx &cx, const DClosure &node, const Frame &frame, float &eta) {
    float e[2] = {1.f, 1.f};
    const auto ha = node_eta<LV>(cx, node.x[0], frame, e[0]), hb = node_eta<LV>(cx, node.x[1], frame, e[1]);
    eta = !ha ? e[1] : (!hb ? e[0] : lerp(e[1], e[0], node.s0));
    return ha || hb;
}

// LayeredSurfaceInstance::populate_closure, layered.cpp:478-500, of the Layered record `c` with frame `own`
// (LV: the Layered levels allowed inside this stack's interfaces)
template<int LV>
LR_D void layer_stack(const MixCtx &cx, const DClosure &c, const Frame &own, LayerStack &layers) {
    load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, c.x[0], own, layers.top, layers.f_top, cx.eta_i);
    float eta_top = 1.f;
#if LR_NEST
    if (!node_eta<LV>(cx, c.x[0], own, eta_top)) { eta_top = 1.f; }
#else
    closure_eta(layers.top, eta_top);
#endif
    load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, c.x[1], own, layers.bottom, layers.f_bottom, eta_top);
    layers.own = own, layers.ng = cx.ng, layers.p = cx.p;
    layers.thickness = c.s0, layers.g = c.s1;
    layers.albedo = mk3(c.c0[0], c.c0[1], c.c0[2]);
    layers.max_depth = c.x[2], layers.samples = c.x[3];
#if LR_NEST
    layers.tb = cx.tb, layers.uv = cx.uv, layers.wo_pop = cx.wo_pop, layers.eta_i = cx.eta_i, layers.eta_bottom = eta_top;
#endif
}

template<int LV>
LR_HEAVY BsdfEval mix_leaf_evaluate(const MixCtx *cx, const DClosure *c, const Frame *fr, f3 wo, f3 wi, bool importance) {
    if constexpr (LV > 0) {
        if (c->kind == LR_SURFACE_LAYERED) {
            LayerStack layers;
            layer_stack<LV - 1>(*cx, *c, *fr, layers);
            return layered_evaluate<LV - 1>(layers, wo, wi, importance);
        }
    }
    return closure_evaluate<true>(*c, *fr, cx->ng, wo, wi, importance);
}
template<int LV>
LR_HEAVY BsdfSample mix_leaf_sample(const MixCtx *cx, const DClosure *c, const Frame *fr, f3 wo, float u_lobe, f2 u, bool importance) {
    if constexpr (LV > 0) {
        if (c->kind == LR_SURFACE_LAYERED) {
            LayerStack layers;
            layer_stack<LV - 1>(*cx, *c, *fr, layers);
            return layered_sample<LV - 1>(layers, wo, u_lobe, u, importance);
        }
    }
    return closure_sample<true>(*c, *fr, cx->ng, wo, u_lobe, u, importance);
}

// MixSurfaceClosure::_evaluate (mix.cpp:169-177) + the public wrapper's side validation (surface.cpp:45-56), of the tree below the
// loaded Mix node `root`: post-order -- both children, then _mix(a, b, ratio), then the sides against the node's own frame
template<int LV>
LR_D BsdfEval mix_evaluate(const MixCtx &cx, const DClosure &root, const Frame &root_frame, f3 wo, f3 wi, bool importance) {
    MixLevel level[kMixMaxDepth + 1];
    BsdfEval first[kMixMaxDepth + 1];
    level[0] = MixLevel{root_frame, root.s0, {root.x[0], root.x[1]}, 0u};
    auto sp = 1;
    for (;;) {
        DClosure child;
        Frame fr;
        {
            const auto &top = level[sp - 1];
            load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, top.child[top.next], top.frame, child, fr, cx.eta_i);
        }
        if (child.kind == LR_SURFACE_MIX && sp <= kMixMaxDepth) {// down
            level[sp] = MixLevel{fr, child.s0, {child.x[0], child.x[1]}, 0u};
            sp++;
            continue;
        }
        auto e = mix_leaf_evaluate<LV>(&cx, &child, &fr, wo, wi, importance);
        for (;;) {// up
            auto &node = level[sp - 1];
            if (node.next == 0u) {
                first[sp - 1] = e, node.next = 1u;
                break;
            }
            e = mix_blend(first[sp - 1], e, node.ratio);
            if (!valid_sides(cx.ng, node.frame.n, wo, wi)) { e.f = mk3(0.f), e.pdf = 0.f; }
            if (--sp == 0) { return e; }
        }
    }
}

// MixSurfaceClosure::_sample (mix.cpp:178-196); the "sample b" branch samples A and evaluates B (reference quirk, kept), so
// sampling always walks down the chain of first children, and on the way back up every node evaluates its second child's subtree
// in the sampled direction
template<int LV>
LR_D BsdfSample mix_sample(const MixCtx &cx, const DClosure &root, const Frame &root_frame, f3 wo, float u_lobe, f2 u_bsdf, bool importance) {
    MixLevel level[kMixMaxDepth + 1];// (.next: 1 if the node's lobe number fell below its ratio)
    auto sp = 0;
    DClosure child;
    Frame fr;
    {
        auto ratio = root.s0;
        auto first_child = root.x[0], second_child = root.x[1];
        auto frame = root_frame;
        for (;;) {
            const auto first = u_lobe < ratio;
            u_lobe = first ? u_lobe / ratio : (u_lobe - ratio) / (1.f - ratio);
            level[sp] = MixLevel{frame, ratio, {first_child, second_child}, first ? 1u : 0u};
            sp++;
            load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, first_child, frame, child, fr, cx.eta_i);
            if (child.kind != LR_SURFACE_MIX || sp > kMixMaxDepth) { break; }
            ratio = child.s0, first_child = child.x[0], second_child = child.x[1], frame = fr;
        }
    }
    auto bs = mix_leaf_sample<LV>(&cx, &child, &fr, wo, u_lobe, u_bsdf, importance);
    while (sp > 0) {
        sp--;
        const auto &node = level[sp];
        load_lobe(cx.tb, cx.uv, cx.ng, cx.wo_pop, node.child[1], node.frame, child, fr, cx.eta_i);
        const auto eb = child.kind == LR_SURFACE_MIX ? mix_evaluate<LV>(cx, child, fr, wo, bs.wi, importance) :
                                                       mix_leaf_evaluate<LV>(&cx, &child, &fr, wo, bs.wi, importance);
        const auto m = node.next != 0u ? mix_blend(BsdfEval{bs.f, bs.pdf}, eb, node.ratio) : mix_blend(eb, BsdfEval{bs.f, bs.pdf}, node.ratio);
        bs.f = m.f, bs.pdf = m.pdf;
        if (!valid_sides(cx.ng, node.frame.n, wo, bs.wi)) { bs.f = mk3(0.f), bs.pdf = 0.f; }
    }
    return bs;
}

#if LR_NEST
// a Mix tree as an interface of a Layered surface (dev_layered.h: layer_eval / layer_sample); its children are populated with
// the interface's eta_i (mix.cpp:210-211)
template<int LV>
__device__ __noinline__ BsdfEval layer_mix_evaluate(const LayerStack &L, bool is_top, f3 wo, f3 wi, bool importance) {
    const MixCtx cx{L.tb, L.uv, L.ng, L.p, L.wo_pop, is_top ? L.eta_i : L.eta_bottom};
    return mix_evaluate<LV>(cx, is_top ? L.top : L.bottom, is_top ? L.f_top : L.f_bottom, wo, wi, importance);
}
template<int LV>
__device__ __noinline__ BsdfSample layer_mix_sample(const LayerStack &L, bool is_top, f3 wo, float uc, f2 u, bool importance) {
    const MixCtx cx{L.tb, L.uv, L.ng, L.p, L.wo_pop, is_top ? L.eta_i : L.eta_bottom};
    return mix_sample<LV>(cx, is_top ? L.top : L.bottom, is_top ? L.f_top : L.f_bottom, wo, uc, u, importance);
}
// a Layered surface as an interface of a Layered surface: LayeredSurfaceInstance::populate_closure (layered.cpp:478-500) of the
// interface's record, with the eta_i the interface was populated with.  (Only the outermost level calls this: the inner stack's own
// interfaces hold no Layered surface any more.)
__device__ __noinline__ void layer_inner_stack(const LayerStack &L, bool is_top, LayerStack &inner) {
    const MixCtx cx{L.tb, L.uv, L.ng, L.p, L.wo_pop, is_top ? L.eta_i : L.eta_bottom};
    layer_stack<0>(cx, is_top ? L.top : L.bottom, is_top ? L.f_top : L.f_bottom, inner);
}
#endif

// evaluate of a Disney / Mix / Layered surface (MIX / LAYERED: which interpreters this kernel variant holds)
template<bool MIX, bool LAYERED>
LR_HEAVY BsdfEval heavy_evaluate(const HeavyCtx *cxp, f3 wi) {
    auto &cx = *cxp;
    auto &c = cx.closure;
    if (MIX && c.kind == LR_SURFACE_MIX) {// mix.cpp:169-177
        if (c.x[2] != 0u) { return mix_evaluate<(LAYERED && LR_NEST) ? kLayerLevels : 0>(mix_ctx_of(cx), c, cx.shading, cx.wo, wi, false); }// a tree: the general interpreter
        // the common case, two basic / Disney children, keeps the closure interpreter inline (the out-of-line leaves of the
        // general path cost C5 3 %)
        BsdfEval e[2];
#pragma nounroll
        for (auto k = 0u; k < 2u; k++) {
            DClosure child;
            Frame fr;
            load_lobe(cx.tb, cx.uv, cx.ng, cx.wo, c.x[k], cx.shading, child, fr);
            e[k] = closure_evaluate<true>(child, fr, cx.ng, cx.wo, wi);
        }
        auto eval = mix_blend(e[0], e[1], c.s0);
        if (!valid_sides(cx.ng, cx.shading.n, cx.wo, wi)) { eval.f = mk3(0.f), eval.pdf = 0.f; }
        return eval;
    }
    if (LAYERED && c.kind == LR_SURFACE_LAYERED) {
        LayerStack layers;
        layer_stack<kLayerLevels - 1>(mix_ctx_of(cx), c, cx.shading, layers);
        return layered_evaluate<kLayerLevels - 1>(layers, cx.wo, wi, false);
    }
    return closure_evaluate<true>(c, cx.shading, cx.ng, cx.wo, wi);// Disney
}

template<bool MIX, bool LAYERED>
LR_HEAVY HeavySample heavy_sample(const HeavyCtx *cxp, float u_lobe, f2 u_bsdf) {
    auto &cx = *cxp;
    auto &c = cx.closure;
    HeavySample r;
    r.eta = 1.f, r.has_eta = 0u;
    if (MIX && c.kind == LR_SURFACE_MIX && c.x[2] != 0u) {// a Mix tree
        const auto mx = mix_ctx_of(cx);
        r.bs = mix_sample<(LAYERED && LR_NEST) ? kLayerLevels : 0>(mx, c, cx.shading, cx.wo, u_lobe, u_bsdf, false);
        r.has_eta = mix_eta<(LAYERED && LR_NEST) ? kLayerLevels : 0>(mx, c, cx.shading, r.eta) ? 1u : 0u;
        return r;
    }
    if (MIX && c.kind == LR_SURFACE_MIX) {// mix.cpp:178-196; the "sample b" branch samples A and evaluates B (reference quirk, kept)
        const auto ratio = c.s0;
        DClosure ca, cb;
        Frame fa, fb;
        load_lobe(cx.tb, cx.uv, cx.ng, cx.wo, c.x[0], cx.shading, ca, fa);
        auto first = u_lobe < ratio;
        r.bs = closure_sample<true>(ca, fa, cx.ng, cx.wo, first ? u_lobe / ratio : (u_lobe - ratio) / (1.f - ratio), u_bsdf);
        float eta_a = 1.f, eta_b = 1.f;
        auto has_a = closure_eta(ca, eta_a);
        load_lobe(cx.tb, cx.uv, cx.ng, cx.wo, c.x[1], cx.shading, cb, fb);
        auto eb = closure_evaluate<true>(cb, fb, cx.ng, cx.wo, r.bs.wi);
        auto m = first ? mix_blend(BsdfEval{r.bs.f, r.bs.pdf}, eb, ratio) : mix_blend(eb, BsdfEval{r.bs.f, r.bs.pdf}, ratio);
        r.bs.f = m.f, r.bs.pdf = m.pdf;
        if (!valid_sides(cx.ng, cx.shading.n, cx.wo, r.bs.wi)) { r.bs.f = mk3(0.f), r.bs.pdf = 0.f; }
        auto has_b = closure_eta(cx.tb.closures[c.x[1]], eta_b);// (eta never comes from an image texture here)
        r.has_eta = (has_a || has_b) ? 1u : 0u;// MixSurfaceClosure::eta, mix.cpp:148-157
        r.eta = !has_a ? eta_b : (!has_b ? eta_a : lerp(eta_b, eta_a, ratio));
        return r;
    }
    if (LAYERED && c.kind == LR_SURFACE_LAYERED) {
        const auto mx = mix_ctx_of(cx);
        LayerStack layers;
        layer_stack<kLayerLevels - 1>(mx, c, cx.shading, layers);
        r.bs = layered_sample<kLayerLevels - 1>(layers, cx.wo, u_lobe, u_bsdf, false);
#if LR_NEST
        r.has_eta = node_eta<kLayerLevels - 1>(mx, c.x[1], cx.shading, r.eta) ? 1u : 0u;// LayeredSurfaceClosure::eta, layered.cpp:252
#else
        r.has_eta = closure_eta(layers.bottom, r.eta) ? 1u : 0u;// LayeredSurfaceClosure::eta, layered.cpp:252
#endif
        return r;
    }
    r.bs = closure_sample<true>(c, cx.shading, cx.ng, cx.wo, u_lobe, u_bsdf);// Disney
    r.has_eta = closure_eta(c, r.eta) ? 1u : 0u;
    return r;
}

}// namespace lrd
