#!/usr/bin/env python3
"""Same-box A/B of several builds of liblrhip.so (GPU box): kernel time of a frame per build and scheduler, and whether the films are the SAME
films -- bit for bit -- as the first build's.  Round 5: every candidate of the traversal loop's bookkeeping must leave the walk, and with it
the film, untouched; `r04` (the round-4 sources built beside the current tree) is the reference.

    python tools/ab_libs.py <spp> <workloads, comma separated> <build>...      build = base (the shipped library) or a name under lib/variants/
    SCHED=pool|legacy|both (default both)   SAMPLER=Independent|PaddedSobol|...   REPEAT=2
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
import numpy as np
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import cornell_box, generate_room_scene
from luisarender_amd.scenes.configs import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene

spp = int(sys.argv[1])
workloads = sys.argv[2].split(",")
builds = sys.argv[3:] or ["base"]
scheds = {"both": ("legacy", "pool"), "pool": ("pool",), "legacy": ("legacy",), "auto": ("auto",)}[os.environ.get("SCHED", "both")]
sampler = os.environ.get("SAMPLER", "Independent")
repeat = int(os.environ.get("REPEAT", "2"))
RES = {"c1": (512, 512), "c2": (1024, 1024), "c3": (1280, 720), "c4": (3840, 2160), "c5": (1280, 720)}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_of(name):
    return None if name == "base" else os.path.join(ROOT, "luisarender_amd", "lib", "variants", f"liblrhip_{name}.so")


for wl in workloads:
    with tempfile.TemporaryDirectory() as tmp:
        res = RES[wl]
        if wl == "c1":
            sc = Scene.from_string(cornell_box(resolution=res[0], spp=spp, depth=8, sampler=sampler))
        else:
            gen = {"c2": generate_room_scene, "c3": generate_bedroom_scene, "c4": generate_camera_scene, "c5": generate_kitchen_scene}[wl]
            sc = Scene.load(gen(tmp, resolution=res, spp=spp, sampler=sampler))
        first = {}
        for b in builds:
            for sched in scheds:
                try:
                    r = MegaPathRenderer(0, lib_path=lib_of(b))
                    if sched != "auto":
                        r.set_scheduler(sched == "pool")
                    r.upload(sc)
                    r.render(0, min(spp, 4), sync=True)  # warm-up (allocations, code load)
                    ms = []
                    for _ in range(repeat):
                        r.clear()
                        t0 = time.perf_counter()
                        r.render(0, spp, sync=True)
                        ms.append((r.last_render_ms(), (time.perf_counter() - t0) * 1e3))
                    film = r.download(converted=False)
                    variant = r.last_variant()
                    r.close()
                except Exception as e:  # noqa: BLE001  (a build without this variant: say so and go on)
                    print(f"{wl} {b:8s} {sched:6s} FAILED: {e}", flush=True)
                    continue
                k, w = min(ms)
                n = res[0] * res[1] * spp
                ref = first.setdefault(sched, film)
                same = "reference" if ref is film else ("bit-identical" if np.array_equal(ref, film) else
                                                         "DIFFERENT rel-L1 %.3e, %d of %d words" % (np.abs(ref - film).sum() / max(np.abs(ref).sum(), 1e-30), int((ref != film).sum()), film.size))
                print(f"{wl} {b:8s} {sched:6s} variant {variant:5d}  kernel {k:9.2f} ms  {n / k / 1e3:8.1f} Msamples/s  (all runs: {' '.join('%.2f' % m[0] for m in ms)})  film {same}", flush=True)
