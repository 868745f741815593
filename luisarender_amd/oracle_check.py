"""Python handle on the CPU oracle (oracle/liboracle.so).

CHECKER ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing in the product path (render.py, csrc/) imports this module."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi
from .scene import Scene


class Oracle:
    def __init__(self, scene: Scene, camera: int = 0):
        self._lib = _ffi.oracle_lib()
        self._scene = scene
        self._view = scene.view(camera)
        self._ctx = self._lib.oracle_create(C.byref(self._view))
        self.width, self.height = int(self._view.camera.width), int(self._view.camera.height)

    def render(self, spp_begin: int, spp_end: int, rect=None, threads: int | None = None, film: np.ndarray | None = None):
        """-> (raw film float4[H, W] = (sum rgb, n), counters dict)"""
        x0, y0, x1, y1 = rect if rect else (0, 0, self.width, self.height)
        if film is None:
            film = np.zeros((self.height, self.width, 4), np.float32)
        cnt = _ffi.OracleCounters()
        threads = threads or os.cpu_count() or 1
        rc = self._lib.oracle_render(self._ctx, spp_begin, spp_end, x0, y0, x1, y1, threads, film.ctypes.data, C.byref(cnt))
        if rc != 0:
            raise RuntimeError("oracle_render failed")
        return film, cnt.as_dict()

    @staticmethod
    def render_frame(scene: Scene, camera: int = 0, threads: int | None = None):
        """The reference's loop over shutter samples (src/base/integrator.cpp:86-107) on the CPU: one oracle per sample time"""
        film, total = None, None
        begin = 0
        for time, weight, spp in scene.shutter_samples(camera):
            scene.set_time(time)
            o = Oracle(scene, camera)
            o._lib.oracle_set_shutter_weight(o._ctx, weight)
            film, cnt = o.render(begin, begin + spp, threads=threads, film=film)
            total = cnt if total is None else {k: total[k] + v for k, v in cnt.items()}
            begin += spp
            o.close()
        return film, total

    def convert(self, film: np.ndarray) -> np.ndarray:
        out = np.empty_like(film)
        self._lib.oracle_film_convert(C.byref(self._view), film.ctypes.data, out.ctypes.data)
        return out

    def li(self, px: int, py: int, sample: int) -> np.ndarray:
        out = np.zeros(3, np.float32)
        self._lib.oracle_li(self._ctx, px, py, sample, out.ctypes.data)
        return out

    def trace_closest(self, origin, direction, t_min=0.0, t_max=3.0e38):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        ids = np.zeros(2, np.uint32)
        bt = np.zeros(3, np.float32)
        self._lib.oracle_trace_closest(self._ctx, o.ctypes.data, d.ctypes.data, t_min, t_max, ids.ctypes.data, bt.ctypes.data)
        return int(ids[0]), int(ids[1]), float(bt[0]), float(bt[1]), float(bt[2])

    def camera_ray(self, px: int, py: int, sample: int) -> np.ndarray:
        out = np.zeros(7, np.float32)
        self._lib.oracle_camera_ray(self._ctx, px, py, sample, out.ctypes.data)
        return out

    def close(self):
        if self._ctx:
            self._lib.oracle_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def algorithmic_bytes(counters: dict) -> float:
    """B = 64*nodes + 48*tris + 188*hits + 208*nee + 32 per sample (film read+write)."""
    return (64.0 * counters["nodes_visited"] + 48.0 * counters["tris_tested"] + 188.0 * counters["surface_hits"]
            + 208.0 * counters["nee_samples"] + 32.0 * counters["paths"])
