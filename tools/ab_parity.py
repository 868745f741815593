"""Parity summary of a kernel variant (LRHIP_LIB) vs the oracle on Cornell + materials + room scene."""
import sys, tempfile, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box, generate_room_scene
from golden.make_golden import cornell_materials_text
r = MegaPathRenderer(0)
def rel(a, b): return float(np.abs(a[..., :3] - b[..., :3]).sum() / np.abs(b[..., :3]).sum())
with tempfile.TemporaryDirectory() as tmp:
    for name, sc, spp in (("cornell", Scene.from_string(cornell_box(128, 16)), 16), ("materials", Scene.from_string(cornell_materials_text()), 8),
                          ("room", Scene.load(generate_room_scene(tmp, resolution=(192, 192), spp=4)), 4)):
        r.upload(sc); r.render(0, spp, counters=True, sync=True)
        g, gc = r.download(False), r.counters()
        c, cc = Oracle(sc).render(0, spp)
        print(name, 'relL1 %.3e' % rel(g, c), 'closest rays gpu/cpu', gc['closest_rays'], cc['closest_rays'], 'mean diff %.2e' % (abs(g[..., :3].mean() - c[..., :3].mean()) / c[..., :3].mean()))
