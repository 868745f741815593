#!/usr/bin/env python3
"""Regenerates tests/golden/ref_*.npz: frames rendered by the REFERENCE'S OWN CODE (oracle/_ref/libref.so = /root/reference/src
compiled in place against the scalar LuisaCompute stand-in, oracle/Makefile.ref), through its own frame loop
(src/base/integrator.cpp:34-113): `image` is what its save_image received (float RGBA, converted film).

These are the reference-produced golden vectors of SURVEY §8(c): tests/test_ref_golden.py holds the oracle to them on the CPU
(bit for bit) and the HIP path to them on the GPU box.  Needs /root/reference (make ref); the fixtures travel, libref need not.
    python tests/golden/make_ref_golden.py [name ...]      (no names: every fixture)
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ref_helpers as R  # noqa: E402
from ref_scenes import scenes  # noqa: E402


def main():
    with tempfile.TemporaryDirectory() as d:
        for name, (text, spp) in scenes(d).items():
            if len(sys.argv) > 1 and name not in sys.argv[1:]:
                continue
            rs = R.RefScene(text, d)
            image = rs.render()
            rs.close()
            np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), image=image, spp=spp)
            print(f"ref_{name}.npz: {image.shape[1]}x{image.shape[0]} mean {image[..., :3].mean():.5f}")


if __name__ == "__main__":
    main()
