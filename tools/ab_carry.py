#!/usr/bin/env python3
"""Wavefront mode: the rounds a slice runs before it hands its parked paths over to the next slice (lrhip_set_wavefront, mode | rounds << 8), kernel
time of a kitchen-class frame per setting and whether the films are THE SAME films (GPU box).      python tools/ab_carry.py <spp> [rounds ...]"""
import sys
import tempfile

sys.path.insert(0, ".")
import numpy as np
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes.configs import generate_kitchen_scene

spp = int(sys.argv[1])
settings = [int(a) for a in sys.argv[2:]] or [65535, 0, 2, 4]
with tempfile.TemporaryDirectory() as tmp:
    sc = Scene.load(generate_kitchen_scene(tmp, resolution=(1280, 720), spp=spp))
    first = None
    for rounds in settings:
        r = MegaPathRenderer(0)
        r.set_wavefront(True, carry_rounds=rounds)
        r.upload(sc)
        r.render(0, min(spp, 4), sync=True)
        ms = []
        for _ in range(2):
            r.clear()
            r.render(0, spp, sync=True)
            ms.append(r.last_render_ms())
        film = r.download(converted=False)
        r.close()
        if first is None:
            first = film
        same = "reference" if first is film else ("bit-identical" if np.array_equal(first, film) else "DIFFERENT rel-L1 %.3e" % (np.abs(first - film).sum() / np.abs(first).sum()))
        name = {65535: "never (every slice drains)", 0: "default"}.get(rounds, f"{rounds} rounds")
        print(f"c5 {spp} spp  hand-over after {name:28s} kernel {min(ms):9.2f} ms  {1280 * 720 * spp / min(ms) / 1e3:8.1f} Msamples/s  (all runs: {' '.join('%.2f' % m for m in ms)})  film {same}", flush=True)
