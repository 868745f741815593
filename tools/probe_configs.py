#!/usr/bin/env python3
"""GPU-vs-oracle numbers for the C3..C5 stand-ins at reduced size (sets the tolerances of tests/test_gpu_parity.py)
and their full-size throughput.  Usage (GPU box): python tools/probe_configs.py [parity] [bench]"""
import sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from luisarender_amd import Scene
from oracle.check import Oracle, algorithmic_bytes
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene, generate_room_scene

modes = sys.argv[1:] or ["parity", "bench"]
r = MegaPathRenderer(0)
GEN = {"c3": generate_bedroom_scene, "c4": generate_camera_scene, "c5": generate_kitchen_scene}
if "parity" in modes:
    for nm, kw, spp in [("c3", dict(resolution=(256, 144)), 16), ("c4", dict(resolution=(256, 144), texture_size=1024), 8), ("c5", dict(resolution=(256, 144)), 16)]:
        d = tempfile.mkdtemp()
        sc = Scene.load(GEN[nm](d, spp=spp, **kw))
        r.upload(sc); r.render(0, spp, counters=True, sync=True)
        gpu, gc = r.download(False), r.counters()
        cpu, cc = Oracle(sc).render(0, spp)
        g, c = gpu[..., :3], cpu[..., :3]
        rel = np.abs(g - c).sum() / np.abs(c).sum()
        H, W = g.shape[:2]
        blk = lambda f: f[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
        brel = np.abs(blk(g) - blk(c)).sum() / np.abs(blk(c)).sum()
        print(nm, "n equal", np.array_equal(gpu[..., 3], cpu[..., 3]), "rel-L1", rel, "block rel-L1", brel, "mean diff", abs(g.mean() - c.mean()) / c.mean(),
              "closest", gc["closest_rays"], cc["closest_rays"], "frac differing px", float((np.abs(g - c).sum(-1) > 1e-3 * (np.abs(c).sum(-1) + 1e-3)).mean()),
              "bytes/sample", algorithmic_bytes(cc) / cc["paths"], flush=True)
if "bench" in modes:
    for nm, res, spp in [("c3", (1280, 720), 1024), ("c4", (3840, 2160), 128), ("c5", (1280, 720), 1024)]:
        d = tempfile.mkdtemp()
        t0 = time.time()
        sc = Scene.load(GEN[nm](d, spp=spp, resolution=res))
        t1 = time.time()
        r.upload(sc)
        for i in range(2):
            r.clear(); r.render(0, spp, sync=True)
            print(nm, res, spp, f"load {t1 - t0:.1f}s", f"{r.last_render_ms():.1f} ms", f"{res[0] * res[1] * spp / r.last_render_ms() / 1e3:.1f} Msamples/s", flush=True)
