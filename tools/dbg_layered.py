import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import MATERIALS
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box
r = MegaPathRenderer(0)
for material in sys.argv[1:]:
    extra = MATERIALS[material].replace("Surface m ", f"Surface {material} ") + "\n"
    for depth in (2, 8):
        sc = Scene.from_string(cornell_box(resolution=64, spp=128, depth=depth, short_box_surface=material, tall_box_surface=material, extra_surfaces=extra))
        r.upload(sc); r.render(0, 128, counters=True, sync=True)
        gpu = r.download(converted=False); gc = r.counters()
        cpu, cc = Oracle(sc).render(0, 128)
        b = lambda f: f[..., :3].reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
        g, c = b(gpu), b(cpu)
        print(material, 'depth', depth, 'blockL1', float(np.abs(g - c).sum() / np.abs(c).sum()), 'mean ratio', float(g.mean() / c.mean()),
              {k: (gc[k], cc[k]) for k in ('closest_rays', 'shadow_rays')})
        d = (g - c).mean(axis=2) / c.mean()
        print(np.array2string(d, precision=2, suppress_small=True))
