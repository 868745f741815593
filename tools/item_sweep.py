#!/usr/bin/env python3
"""Sweep of the work-item size (lrhip_set_diagnostics: item_scale x the loss model's constant, lrhip_render) on C2: full frame and the 1/8 shard."""
import os, sys, tempfile
sys.path.insert(0, ".")
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_room_scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with tempfile.TemporaryDirectory() as tmp:
    sc = Scene.load(generate_room_scene(tmp, resolution=(1024, 1024), spp=spp))
    r = MegaPathRenderer(0)
    r.upload(sc)
    for world in (1, 8):
        for scale in ("0.25", "0.5", "1", "2", "4", "8"):
            r.set_diagnostics(item_scale=float(scale))
            ms = []
            for _ in range(2):
                r.clear(); r.render(0, spp, rank=0, world=world, sync=True, balance_shards=world)
                ms.append(r.last_render_ms())
            print(f"world {world} item scale {scale}: {min(ms):.1f} ms", flush=True)
