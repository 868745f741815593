import sys, re
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from test_gpu_parity import FOG
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box
text = cornell_box(resolution=64, spp=1, depth=8, extra_surfaces=FOG).replace("integrator : MegaPath {", "integrator : MegaVPTNaive {").replace("render {", "render {\n  environment_medium { @fog }")
shapes = re.search(r"shapes \{ (.*?) \}\n  integrator", text, re.S).group(1)
text = text.replace(f"shapes {{ {shapes} }}", "shapes { @tilted }")
text = text.replace("Camera cam", f"Shape tilted : Group {{ shapes {{ {shapes} }} transform : SRT {{ rotate {{ 0.2, 1, 0.1, 5 }} translate {{ -23, 0, 25 }} }} }}\nCamera cam")
sc = Scene.from_string(text)
r = MegaPathRenderer(0); r.upload(sc)
o = Oracle(sc)
tot_g = tot_c = 0.0; nd = 0; n = 0
for s in range(8):
    r.clear(); r.render(s, s + 1, sync=True)
    g = r.download(False)
    c, _ = o.render(s, s + 1)
    d = np.abs(g[..., :3] - c[..., :3]).sum(-1)
    bad = d > 1e-3 * (np.abs(c[..., :3]).sum(-1) + 1e-3)
    nd += bad.sum(); n += bad.size
    tot_g += g[..., :3].sum(); tot_c += c[..., :3].sum()
    if s == 0:
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:12]:
            print('px', x, y, 'gpu', g[y, x], 'cpu', c[y, x])
print('differing paths', nd, 'of', n, 'sum gpu/cpu', tot_g / tot_c)
