#!/usr/bin/env python3
"""A/B of the two schedulers of the lean kernels on the bench's stand-ins (GPU box): the path-pool kernels of round 4 against the
one-path-per-lane kernels of rounds 1-3 -- kernel time (HIP events), Msamples/s, and how far the two films are apart.
    python tools/ab_sched.py <spp> [c2 c3 c4 c5 ...]      (LRHIP_LIB selects an experimental build of the library)"""
import os, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_room_scene, cornell_box
from luisarender_amd.scenes.configs import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
workloads = sys.argv[2:] or ["c2"]
RES = {"c1": (512, 512), "c2": (1024, 1024), "c3": (1280, 720), "c4": (3840, 2160), "c5": (1280, 720)}
for wl in workloads:
    with tempfile.TemporaryDirectory() as tmp:
        res = RES[wl]
        if wl == "c1":
            sc = Scene.from_string(cornell_box(resolution=res[0], spp=spp, depth=8))
        else:
            gen = {"c2": generate_room_scene, "c3": generate_bedroom_scene, "c4": generate_camera_scene, "c5": generate_kitchen_scene}[wl]
            sc = Scene.load(gen(tmp, resolution=res, spp=spp))
        films = {}
        for sched in ("legacy", "pool"):
            r = MegaPathRenderer(0)
            r.set_scheduler(sched == "pool")
            if os.environ.get("LR_ITEM_SCALE"):
                r.set_diagnostics(item_scale=float(os.environ["LR_ITEM_SCALE"]))
            r.upload(sc)
            r.render(0, min(spp, 4), sync=True)  # warm-up (allocations, code load)
            r.clear()
            ms = []
            for _ in range(2):
                r.clear()
                t0 = time.perf_counter()
                r.render(0, spp, sync=True)
                wall = (time.perf_counter() - t0) * 1e3
                ms.append((r.last_render_ms(), wall))
            films[sched] = r.download(converted=False)
            k, w = min(ms)
            n = res[0] * res[1] * spp
            print(f"{wl} {sched:6s} variant {r.last_variant():5d}  kernel {k:9.2f} ms  wall {w:9.2f} ms  {n / k / 1e3:8.1f} Msamples/s (kernel)  {n / w / 1e3:8.1f} (wall)", flush=True)
            r.close()
        a, b = films["legacy"], films["pool"]
        print(f"{wl} films: counts equal {bool((a[..., 3] == b[..., 3]).all())}  mean legacy {a[..., :3].mean():.6f} pool {b[..., :3].mean():.6f}  "
              f"rel L1 {np.abs(a[..., :3] - b[..., :3]).sum() / max(np.abs(a[..., :3]).sum(), 1e-30):.3e}  finite {bool(np.isfinite(b).all())}", flush=True)
