#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
LRHIP_HEAVY_QUEUE=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "twins and layered" 2>&1 | grep -B5 -A25 "def test_shipped" | grep "^E\|assert" | cut -c1-300 | head -20
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import cornell_box
from helpers import MATERIALS
mat = lambda *names: "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in names)
text = cornell_box(resolution=64, spp=8, short_box_surface="layered", tall_box_surface="layered_medium", extra_surfaces=mat("layered", "layered_medium"))
sc = Scene.from_string(text)
r = MegaPathRenderer(0)
for q in ("0", "2048"):
    os.environ["LRHIP_HEAVY_QUEUE"] = q
    films = []
    for count in (True, False, False):
        r.upload(sc); r.render(0, 8, counters=count, sync=True)
        f = r.download(converted=False); films.append(f)
        print("queue", q, "count", count, "variant", r.last_variant(), "n==8:", float((f[..., 3] == 8).mean()), "min n", f[..., 3].min(), "finite", bool(np.isfinite(f).all()), "sum", float(f[..., :3].sum()))
    print("  twin rel diff", float(np.abs(films[0][..., :3] - films[1][..., :3]).sum() / films[0][..., :3].sum()), "repeat identical", bool(np.array_equal(films[1], films[2])))
PY
} > gpurun_out/r02ac.txt 2>&1
cat gpurun_out/r02ac.txt
