#!/usr/bin/env python3
"""Does the default SGPR spilling (SGPRs into lanes of a VGPR) still miscompile the kernels that make real calls?  (DESIGN.md 4.1:
NaN samples / lost samples in <124> three times in rounds 1-2, each time depending on unrelated code; every call-making object is
built with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 since.)  Renders the Layered / Mix / nested parity scenes with the SHIPPED library
and with a build of the same sources WITHOUT the flag (make hip-variant NAME=unsafe CALL_SAFE_FLAGS= ...), all-in-one megakernels and
wavefront mode, counting and non-counting binaries, three times each, and compares: rejected (NaN / Inf) samples, films against the
shipped build's, run-to-run identity.
    python tools/sgpr_spill_repro.py luisarender_amd/lib/variants/liblrhip_unsafe.so"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import MATERIALS  # noqa: E402
from luisarender_amd import Scene  # noqa: E402
from luisarender_amd.render import MegaPathRenderer  # noqa: E402
from luisarender_amd.scenes import cornell_box  # noqa: E402

other = os.path.abspath(sys.argv[1])
mat = lambda *names: "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in names)
scenes = {
    "layered": cornell_box(resolution=64, spp=64, short_box_surface="layered", tall_box_surface="layered_medium", extra_surfaces=mat("layered", "layered_medium")),
    "mix": cornell_box(resolution=64, spp=64, short_box_surface="mix_nested", tall_box_surface="mix_glass", extra_surfaces=mat("mix_nested", "mix_glass")),
    "nested": cornell_box(resolution=64, spp=64, short_box_surface="mix_layered", tall_box_surface="layered_mix", extra_surfaces=mat("mix_layered", "layered_mix")),
}
bad = 0
for name, text in scenes.items():
    sc = Scene.from_string(text)
    for wavefront in (False, True):
        ref = None
        for label, lib in (("shipped", None), ("no-flag", other)):
            r = MegaPathRenderer(0, lib_path=lib)
            r.set_wavefront(wavefront)
            for count in (False, True):
                films = []
                for _ in range(3):
                    r.upload(sc)
                    r.render(0, 64, counters=count, sync=True)
                    films.append(r.download(False))
                missing = int((64 - films[0][..., 3]).sum())
                same = all(np.array_equal(films[0], f) for f in films[1:])
                if ref is None:
                    ref = films[0]
                err = float(np.abs(films[0][..., :3] - ref[..., :3]).sum() / np.abs(ref[..., :3]).sum())
                flag = "" if (missing == 0 and same and err < 1e-3) else "   <-- BROKEN"
                bad += flag != ""
                print(f"{name:8s} {'wavefront ' if wavefront else 'all-in-one'} {label:8s} count={int(count)} variant {r.last_variant():5d}: rejected samples {missing:6d}, "
                      f"three runs identical {same}, rel-L1 vs shipped non-counting {err:.2e}{flag}", flush=True)
            r.close()
print("broken configurations:", bad)
