#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle (fixed seed 19980810).

The reference holds no golden images for this path and its own binary cannot be built in this
container, so these fixtures pin the ORACLE, not the reference: they are regression
vectors that the HIP path is then compared against on the GPU box.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from luisarender_amd import Scene  # noqa: E402
from oracle.check import Oracle  # noqa: E402
from luisarender_amd.scenes import cornell_box  # noqa: E402


def cornell_materials_text():
    from helpers import MATERIALS
    extra = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("mirror", "glass", "plastic", "metal"))
    return cornell_box(resolution=48, spp=8, short_box_surface="glass", tall_box_surface="metal", extra_surfaces=extra)


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    sc = Scene.from_string(cornell_box(resolution=32, spp=8))
    film, counters = Oracle(sc).render(0, 8)
    np.savez_compressed(os.path.join(out, "cornell_32_8spp.npz"), film=film, closest_rays=counters["closest_rays"])
    sc = Scene.from_string(cornell_materials_text())
    film, counters = Oracle(sc).render(0, 8)
    np.savez_compressed(os.path.join(out, "cornell_materials_48_8spp.npz"), film=film, closest_rays=counters["closest_rays"])
    print("golden fixtures written to", out)


if __name__ == "__main__":
    main()
