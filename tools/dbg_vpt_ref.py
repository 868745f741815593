"""GPU debugging aid: the volumetric megakernel against the oracle (== the reference, tests/test_oracle_vs_ref.py) on fog scenes
of growing complexity; prints per-pixel / block errors and ray counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes.cornell import cornell_box
from oracle.check import Oracle
from ref_scenes import FOG

def scene(case, res=32, spp=8):
    extra = "" if case == "vacuum" else FOG
    boxed = "box" in case
    t = cornell_box(resolution=res, spp=spp, depth=8, extra_surfaces=extra, short_box_surface="skin" if boxed else "white")
    t = t.replace("integrator : MegaPath {", "integrator : MegaVPTNaive {")
    if case != "vacuum":
        t = t.replace("render {", "render {\n  environment_medium { @fog }")
    if boxed:
        t = t.replace("surface { @skin }", "surface { @skin } medium { @inner }")
    if case == "box_only":
        t = t.replace("\n  environment_medium { @fog }", "")
    return t

r = MegaPathRenderer(0)
for case in ["vacuum", "fog_lamp", "box_only", "fog_lamp_box"]:
    for spp in (8, 64):
        sc = Scene.from_string(scene(case, 32, spp))
        r.upload(sc)
        r.render(0, spp, counters=True, sync=True)
        gpu, gc = r.download(converted=False), r.counters()
        cpu, cc = Oracle(sc).render(0, spp)
        g, c = gpu[..., :3], cpu[..., :3]
        blocks = lambda f: f.reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
        print(f"{case:14s} spp {spp:3d}: rel-L1 {np.abs(g - c).sum() / np.abs(c).sum():.4f} block {np.abs(blocks(g) - blocks(c)).sum() / np.abs(blocks(c)).sum():.4f} "
              f"mean {g.mean() / c.mean() - 1:+.4f} rays gpu {gc['closest_rays']} cpu {cc['closest_rays']} nee {gc['nee_samples']} {cc['nee_samples']} "
              f"n-diff px {(gpu[..., 3] != cpu[..., 3]).sum()} identical px {(np.abs(g - c).max(axis=-1) <= 1e-4 * np.abs(c).max(axis=-1) + 1e-7).mean():.3f}")

# which samples differ: one sample at a time
sc = Scene.from_string(scene("fog_lamp", 32, 8))
o = Oracle(sc)
r.upload(sc)
r.render(0, 8, counters=True, sync=True)
gpu = r.download(converted=False)
cpu, _ = o.render(0, 8)
bad = np.argwhere(np.abs(gpu[..., :3] - cpu[..., :3]).max(axis=-1) > 1e-3 * np.abs(cpu[..., :3]).max(axis=-1) + 1e-6)
print("differing pixels:", len(bad))
for py, px in bad[:6]:
    for s in range(8):
        r.upload(sc)
        r.render(s, s + 1, counters=True, sync=True)
        g = r.download(converted=False)[py, px]
        c = o.li(int(px), int(py), s)
        if not np.allclose(g[:3], c, rtol=1e-3, atol=1e-6) or g[3] != 1:
            print(f"px ({px},{py}) sample {s}: gpu {g} oracle {c}")
