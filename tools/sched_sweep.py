#!/usr/bin/env python3
"""Where do the two schedulers of the lean kernels break even (GPU box)?  The room stand-in swept over its triangle count, the path depth
and the samples per pixel -- the quantities that drive what the path pool buys (long walks, the drain of a launch) and what it costs (a
costlier shading block) -- kernel time of the one-path-per-lane kernel, the pool kernel, and which of them lrhip_render's automatic rule
(lrhip.hip: wants_pool) picks.  VERDICT r04 item 8: no configuration where the automatic choice is the slower one by more than noise.

    python tools/sched_sweep.py [quick]
"""
import sys
import tempfile

sys.path.insert(0, ".")
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import cornell_box, generate_room_scene

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
TRIS = (5_000, 30_000, 60_000, 100_000, 400_000) if not quick else (30_000, 100_000)
DEPTHS = (4, 16)
SPPS = (16, 256)
RES = (768, 768)
worst = 1.0
rows = []


def time_both(scene, spp, label):
    global worst
    out = {}
    for name, pool in (("lane", False), ("pool", True), ("auto", None)):
        r = MegaPathRenderer(0)
        r.set_scheduler(pool)
        r.upload(scene)
        r.render(0, min(spp, 4), sync=True)
        ms = []
        for _ in range(2):
            r.clear()
            r.render(0, spp, sync=True)
            ms.append(r.last_render_ms())
        out[name] = (min(ms), r.last_variant())
        r.close()
    lane, pool, auto = out["lane"][0], out["pool"][0], out["auto"][0]
    picked = "pool" if out["auto"][1] & 4096 else "lane"
    best = min(lane, pool)
    loss = (lane if picked == "lane" else pool) / best  # what the automatic choice costs against the better of the two
    worst = max(worst, loss)
    print(f"{label}: one path per lane {lane:8.2f} ms  pool {pool:8.2f} ms  lane/pool {lane / pool:.3f}  automatic -> {picked} ({auto:.2f} ms), {loss:.3f} x the better one", flush=True)


with tempfile.TemporaryDirectory() as tmp:
    for depth in (4, 8):
        for spp in (16, 64, 1024):
            time_both(Scene.from_string(cornell_box(resolution=512, spp=spp, depth=depth)), spp, f"cornell 512 x 512 depth {depth:2d} spp {spp:4d}")
    for tris in TRIS:
        for depth in DEPTHS:
            path = generate_room_scene(tmp, target_triangles=tris, resolution=RES, spp=max(SPPS), depth=depth, name=f"room_{tris}_{depth}")
            scene = Scene.load(path)
            for spp in SPPS:
                time_both(scene, spp, f"room {tris:7d} target triangles depth {depth:2d} spp {spp:4d}")
print(f"worst case of the automatic rule: {worst:.3f} x the better scheduler")
