#!/usr/bin/env python3
"""One GPU stands in for rank 0 of an N-GPU node: time the 1/N screen-tile shard of C2 with and without the
balance_shards item-size hint (lrhip.h).  efficiency = t(N=1) / (N * t(shard))."""
import sys, tempfile
sys.path.insert(0, ".")
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_room_scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with tempfile.TemporaryDirectory() as tmp:
    sc = Scene.load(generate_room_scene(tmp, resolution=(1024, 1024), spp=spp))
    r = MegaPathRenderer(0)
    r.upload(sc)
    modes = (("tapered work items (round 3)", 0.0), ("uniform work items (rounds 1-2)", -1.0))
    if len(sys.argv) > 2:
        modes = modes[:1]
    for label, scale in modes:
        r.set_diagnostics(item_scale=scale)
        base = None
        for world in ((1, 8) if len(sys.argv) > 2 else (1, 2, 4, 8)):
            ms = []
            for _ in range(2):
                r.clear(); r.render(0, spp, rank=0, world=world, sync=True, balance_shards=world)
                ms.append(r.last_render_ms())
            t = min(ms)
            base = base or t
            print(f"{label}: world {world}: shard {t:.1f} ms, kernel-level strong-scaling efficiency {base / (world * t):.3f}", flush=True)
