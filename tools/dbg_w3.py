"""Diagnose a kernel build variant on the Layered parity scenes (GPU box): LRHIP_LIB=<variant .so> python tools/dbg_w3.py
prints how many samples the film rejected (NaN / Inf) against the oracle and where — this is how the 3-waves-per-SIMD build of the
Layered variants was found to be miscompiled."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import MATERIALS
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box
r = MegaPathRenderer(0)
for material in ("layered", "layered_medium"):
    extra = MATERIALS[material].replace("Surface m ", f"Surface {material} ") + "\n"
    spp = 256
    sc = Scene.from_string(cornell_box(resolution=64, spp=spp, short_box_surface=material, tall_box_surface=material, extra_surfaces=extra))
    r.upload(sc); r.render(0, spp, counters=True, sync=True)
    gpu = r.download(converted=False); gc = r.counters()
    cpu, cc = Oracle(sc).render(0, spp)
    d = cpu[..., 3] - gpu[..., 3]
    print(material, 'variant', r.last_variant(), 'samples missing on gpu:', int(d.sum()), 'pixels affected', int((d != 0).sum()), 'max per pixel', int(d.max()), 'min', int(d.min()))
    ys, xs = np.nonzero(d)
    print('where', list(zip(ys[:12].tolist(), xs[:12].tolist())))
    b = lambda f: f[..., :3].reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
    print('block rel L1', float(np.abs(b(gpu) - b(cpu)).sum() / np.abs(b(cpu)).sum()), 'rays', gc['closest_rays'], cc['closest_rays'])
