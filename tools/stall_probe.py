#!/usr/bin/env python3
"""Section-level stall attribution of the pool megakernels (GPU box; the round-6 instrument: the box offers neither PC sampling nor a
thread-trace decoder, profiles/r06a_pc_sampling_unavailable.txt).

    make hip-variant NAME=probe DEFS=-DLR_STALL_PROBE VARIANT_MASKS="1 4097 1033 3081 5129 7177 8213 12309 21 4117 3 4099 20483" HEAVY_MASKS="1 5 9"      (here: the counting twins of both schedulers for C2 - C5 and the PaddedSobol kernels)
    python tools/stall_probe.py <workload> <spp> [out.json]            (on the box; SAMPLER=PaddedSobol, LIB=probe by default)

The probe build of the COUNTING kernels reads s_memtime at every section boundary of the traversal loop's iteration (dev_trace.h: THE
STALL PROBE) and of the shading block (megapool_kernel.h); lane 0 of every wave adds the wave-uniform sums to lrhip_counters::probe.
This script renders one frame with that library and prints, per section: wave cycles per iteration, the share of the loop's cycles,
and -- for the arithmetic-only section -- what it takes beyond its priced issue cycles (= the wave waiting for an issue slot).
The probes cost ~8 SMEM round trips per iteration: the table gives SHARES, the absolute figures are those of the probed kernel
(its own kernel time is printed beside the shipped kernel's).
"""
import json
import os
import sys
import tempfile

sys.path.insert(0, ".")
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import cornell_box, generate_room_scene
from luisarender_amd.scenes.configs import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = {"c1": (512, 512), "c2": (1024, 1024), "c3": (1280, 720), "c4": (3840, 2160), "c5": (1280, 720)}
TRAV = ["issue", "vm_wait", "leaf", "packet", "slab", "chain", "tail", "leaf_wait"]
SHADE = ["hit+light", "lobe (textures)", "evaluate", "draws+sample+rr"]
# priced issue cycles of the arithmetic-only section (slab tests + sort of one packet: 24 cvt + 24 fma + 3 mul + 3 sub/mul + 6 cndmask pairs +
# 8 max/min3 + 4 cmp/cndmask/and_or + 10 min/max of the sort network), tools/isa_census.py classes: full 2.4, half 4.2
SLAB_PRICED = 24 * 4.2 + 27 * 2.4 + 6 * 2.4 + 6 * 6.1 + 16 * 4.2 + 4 * (4.2 + 4.2 + 4.2 + 2.4) + 10 * 4.2


def main():
    wl, spp = sys.argv[1], int(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    sampler = os.environ.get("SAMPLER", "Independent")
    lib = os.path.join(ROOT, "luisarender_amd", "lib", "variants", "liblrhip_%s.so" % os.environ.get("LIB", "probe"))
    with tempfile.TemporaryDirectory() as tmp:
        res = RES[wl]
        if wl == "c1":
            sc = Scene.from_string(cornell_box(resolution=res[0], spp=spp, depth=8, sampler=sampler))
        else:
            gen = {"c2": generate_room_scene, "c3": generate_bedroom_scene, "c4": generate_camera_scene, "c5": generate_kitchen_scene}[wl]
            sc = Scene.load(gen(tmp, resolution=res, spp=spp, sampler=sampler))
        rows = {}
        for name, path, counters in (("shipped", None, False), ("shipped counting twin", None, True), ("probe build", lib, True)):
            r = MegaPathRenderer(0, lib_path=path)
            if os.environ.get("SCHED"):
                r.set_scheduler(os.environ["SCHED"] == "pool")
            r.upload(sc)
            r.render(0, min(spp, 4), counters=counters, sync=True)
            c0 = r.counters()
            r.clear()
            r.render(0, spp, counters=counters, sync=True)
            c1 = r.counters()
            rows[name] = {"kernel_ms": r.last_render_ms(), "variant": r.last_variant(),
                          "counters": {k: ([b - a for a, b in zip(c0[k], v)] if isinstance(v, list) else v - c0[k]) for k, v in c1.items()}}
            r.close()
    c = rows["probe build"]["counters"]
    probe = c["probe"]
    iters = float(probe[12]) or 1.0  # iterations of the SAMPLED waves (one in 36 takes the timestamps)
    all_iters = c["trace_steps"] / 64.0
    trav_total = float(sum(probe[:8])) or 1.0
    out = {"workload": wl, "spp": spp, "sampler": sampler, "kernel_ms": {k: v["kernel_ms"] for k, v in rows.items()}, "variant": {k: v["variant"] for k, v in rows.items()},
           "wave_iterations": iters, "wave_cycles": c["wave_cycles"], "trace_cycles": c["trace_cycles"], "shade_cycles": c["shade_cycles"],
           "traversal": {}, "shading": {}, "counters": c}
    print(f"{wl} {spp} spp {sampler}: " + ", ".join(f"{k} <{v['variant']}> {v['kernel_ms']:.1f} ms" for k, v in rows.items()))
    print(f"  wave cycles: traversal loop {c['trace_cycles'] / c['wave_cycles']:.3f}, shading block {c['shade_cycles'] / c['wave_cycles']:.3f} of a wave's life; "
          f"{all_iters / max(c['paths'], 1) * 64:.1f} lane-iterations per sample; {iters / all_iters:.4f} of the wave iterations sampled; "
          f"sections cover {trav_total / max(probe[13], 1):.3f} of the sampled waves' cycles in the loop ({probe[13] / iters:.0f} per iteration)")
    print("  traversal loop, per wave iteration:      cycles    share")
    for i, n in enumerate(TRAV):
        cyc = probe[i] / max(iters, 1.0)
        out["traversal"][n] = {"cycles_per_iteration": cyc, "share": probe[i] / trav_total}
        print(f"    {n:10s} {cyc:10.1f}  {probe[i] / trav_total:7.3f}")
    node_iters = c["nodes_visited"] / max(c["trace_steps"], 1)  # lanes at an inner node per lane-step
    slab = probe[4] / max(iters, 1.0)
    out["slab_priced_cycles"] = SLAB_PRICED
    out["slab_issue_wait_share"] = max(0.0, 1.0 - SLAB_PRICED / slab) if slab > 0 else None
    if slab > 0:
        print(f"  slab + sort: {slab:.0f} cycles measured against {SLAB_PRICED:.0f} priced issue cycles -> {out['slab_issue_wait_share']:.3f} of the section is the wave waiting for an issue slot "
              f"(lanes at inner nodes {node_iters:.3f}, at leaves {c['tris_tested'] / max(c['trace_steps'], 1):.3f} of the lane-steps)")
    calls = float(probe[14]) or 1.0  # batches of the reporting waves, and their cycles inside the block
    block = float(probe[15]) or 1.0
    all_calls = c["shade_calls"] / 64.0
    print(f"  shading block, per batch of the reporting waves (slowest lane of each section; {block / calls:.0f} cycles per batch, all waves {c['shade_cycles'] / max(all_calls, 1):.0f}):   cycles   share of the block")
    for i, n in enumerate(SHADE):
        cyc = probe[8 + i] / calls
        out["shading"][n] = {"cycles_per_batch": cyc, "share_of_block": probe[8 + i] / block}
        print(f"    {n:18s} {cyc:10.0f}  {probe[8 + i] / block:7.3f}")
    rest = block - sum(probe[8:12])
    out["shading"]["park / state / regeneration / launch"] = {"cycles_per_batch": rest / calls, "share_of_block": rest / block}
    print(f"    {'park/state/regen':18s} {rest / calls:10.0f}  {rest / block:7.3f}   (batches per wave-iteration {all_calls / max(all_iters, 1):.4f}, lanes shading {c['shade_busy'] / max(c['shade_calls'], 1):.3f})")
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
