"""GPU-vs-oracle diagnostics of one cornell material scene under the current LRHIP_LIB (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from helpers import MATERIALS
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box
MATERIALS = dict(MATERIALS)
def mixof(a, b, ratio):
    return (MATERIALS[a].replace("Surface m ", "Surface mix_a ") + " " + MATERIALS[b].replace("Surface m ", "Surface mix_b ") +
            " Surface m : Mix { a { @mix_a } b { @mix_b } ratio : Constant { v { %g } } }" % ratio)
MATERIALS.update({"mm": mixof("matte", "matte", 0.3), "rr": mixof("mirror", "mirror", 0.3), "mr1": mixof("matte", "mirror", 1.0),
                  "mr0": mixof("matte", "mirror", 0.0), "rm": mixof("mirror", "matte", 0.3), "mr5": mixof("matte", "mirror", 0.5)})
r = MegaPathRenderer(0)
for material, force_full in [(m, False) for m in sys.argv[1:]]:
    extra = MATERIALS[material].replace("Surface m ", f"Surface {material} ") + "\n"
    if force_full:
        extra += 'Surface dummy : Disney { color : Constant { v { 0.5 } } }\nShape far : InlineMesh { positions { 5000,5000,5000, 5001,5000,5000, 5000,5001,5000 } indices { 0,1,2 } surface { @dummy } }\n'
    text = cornell_box(resolution=64, spp=16, short_box_surface=material, tall_box_surface=material, extra_surfaces=extra)
    if force_full:
        text = text.replace("shapes {", "shapes { @far,", 1)
    sc = Scene.from_string(text)
    r.upload(sc)
    r.render(0, 16, counters=True, sync=True)
    gpu = r.download(converted=False)
    gc = r.counters()
    o = Oracle(sc)
    cpu, cc = o.render(0, 16)
    print(material, 'full' if force_full else 'lean', {k: (gc[k], cc[k]) for k in ('closest_rays', 'shadow_rays', 'surface_hits')},
          'relL1', float(np.abs(gpu[..., :3] - cpu[..., :3]).sum() / np.abs(cpu[..., :3]).sum()))
