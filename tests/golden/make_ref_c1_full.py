#!/usr/bin/env python3
"""Regenerates tests/golden/ref_c1_full.npz: BASELINE configs[0] AT ITS STATED SIZE -- Cornell box, 512x512, 64 spp, depth 8 --
rendered by the REFERENCE'S OWN CODE (oracle/_ref/libref.so = /root/reference/src compiled in place on the scalar LuisaCompute
stand-in, oracle/Makefile.ref) through its own frame loop (src/base/integrator.cpp:34-113, MegakernelPathTracingInstance::Li
src/integrators/mega_path.cpp:49-156, film convert src/films/color.cpp:87-93).  One thread, about a minute.

`image` is the RGB of what the reference's save_image received (fp32; alpha is 1 everywhere and not stored).
tests/test_ref_golden.py holds the oracle to it on the CPU (windows of the frame, bit for bit) and the SHIPPED HIP kernel <0> to it
on the GPU box (rel-L1, per-pixel RMSE, a FLIP-class perceptual error).
    python tests/golden/make_ref_c1_full.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import ref_helpers as R  # noqa: E402
from luisarender_amd.scenes.cornell import cornell_box  # noqa: E402

C1 = dict(resolution=512, spp=64, depth=8)  # BASELINE.json configs[0]; cornell_box() defaults: Independent sampler, Box filter


def main():
    t = time.time()
    rs = R.RefScene(cornell_box(**C1))
    image = rs.render()
    rs.close()
    assert (image[..., 3] == 1.0).all()
    np.savez_compressed(os.path.join(HERE, "ref_c1_full.npz"), image=np.ascontiguousarray(image[..., :3]), spp=C1["spp"])
    print(f"ref_c1_full.npz: {image.shape[1]}x{image.shape[0]} mean {image[..., :3].mean():.6f} in {time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
