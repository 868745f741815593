"""Python handle on the CPU oracle (oracle/liboracle.so): ctypes prototypes + the Oracle class.

TEST INFRASTRUCTURE ONLY, like everything under oracle/: imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg
and the debugging scripts under tools/ -- never by the product package luisarender_amd (which only lends it its ctypes mirror of
include/lr_scene.h, the structs both sides read)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from luisarender_amd import _ffi
from luisarender_amd.scene import Scene

u32, u64, f32 = C.c_uint32, C.c_uint64, C.c_float


class OracleCounters(C.Structure):
    _fields_ = [(n, u64) for n in ("paths", "closest_rays", "shadow_rays", "nodes_visited", "tris_tested",
                                   "surface_hits", "nee_samples", "path_length_sum")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}



def oracle_lib(name: str = "liboracle.so") -> C.CDLL:
    """The CPU checker.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg call this.
    name: liboracle.so, or liboracle_fma.so = the same source built with -ffp-contract=fast (Makefile: what fused multiply-adds alone do to
    the oracle's frames -- the oracle's OWN sensitivity, tests/test_gpu_parity.py::test_lamp_lit_fog_converges_on_the_oracle)"""
    lib = _ffi._load(os.path.join(os.path.dirname(os.path.abspath(__file__)), name))
    if not getattr(lib, "_lr_ready", False):
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.POINTER(_ffi.Scene)]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_set_shutter_weight.argtypes = [C.c_void_p, f32]
        lib.oracle_set_bake.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_render.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u32, C.c_int, C.c_void_p,
                                      C.POINTER(OracleCounters)]
        lib.oracle_film_convert.argtypes = [C.POINTER(_ffi.Scene), C.c_void_p, C.c_void_p]
        lib.oracle_li.argtypes = [C.c_void_p, u32, u32, u32, C.c_void_p]
        lib.oracle_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32, f32, C.c_void_p, C.c_void_p]
        lib.oracle_camera_ray.argtypes = [C.c_void_p, u32, u32, u32, C.c_void_p]
        for name, n in (("oracle_xxhash32_1", 1), ("oracle_xxhash32_2", 2), ("oracle_xxhash32_3", 3), ("oracle_xxhash32_4", 4)):
            fn = getattr(lib, name)
            fn.restype = u32
            fn.argtypes = [u32] * n
        lib.oracle_sampler_stream.argtypes = [C.POINTER(_ffi.Scene), u32, u32, u32, u32, C.c_void_p]
        lib.oracle_lcg.restype = f32
        lib.oracle_lcg.argtypes = [C.POINTER(u32)]
        lib.oracle_pcg32_next.restype = u32
        lib.oracle_pcg32_next.argtypes = [C.POINTER(u64), C.POINTER(u64)]
        lib.oracle_pcg32_seed.argtypes = [u64, C.POINTER(u64), C.POINTER(u64)]
        lib.oracle_create_alias_table.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p]
        lib.oracle_sample_alias_table.argtypes = [C.c_void_p, u32, f32, C.POINTER(u32), C.POINTER(f32)]
        lib.oracle_filter_sample.argtypes = [C.POINTER(_ffi.Filter), f32, f32, C.c_void_p]
        lib.oracle_encode_handle.argtypes = [u32] * 6 + [f32, f32, C.c_void_p]
        lib.oracle_offset_ray_origin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_surface_evaluate.argtypes = [C.POINTER(_ffi.Scene), u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_surface_sample.argtypes = [C.POINTER(_ffi.Scene), u32, C.c_void_p, C.c_void_p, f32, f32, f32, C.c_void_p]
        lib._lr_ready = True
    return lib



class Oracle:
    def __init__(self, scene: Scene, camera: int = 0, bake_instances: bool = False, lib: str = "liboracle.so"):
        """bake_instances: intersect the fp32 WORLD-space triangles the host bakes for the device (lr_scene.accel.triangles) instead of the
        reference's object-space triangles behind the instance transform -- isolates the kernel from that design choice (oracle_bvh.h)"""
        self._lib = oracle_lib(lib)
        self._scene = scene
        self._view = scene.view(camera)
        self._ctx = self._lib.oracle_create(C.byref(self._view))
        self.width, self.height = int(self._view.camera.width), int(self._view.camera.height)
        if bake_instances and self._lib.oracle_set_bake(self._ctx, 1) != 0:
            raise RuntimeError("oracle: the scene view holds no baked triangles (acceleration structure not built)")

    def render(self, spp_begin: int, spp_end: int, rect=None, threads: int | None = None, film: np.ndarray | None = None):
        """-> (raw film float4[H, W] = (sum rgb, n), counters dict)"""
        x0, y0, x1, y1 = rect if rect else (0, 0, self.width, self.height)
        if film is None:
            film = np.zeros((self.height, self.width, 4), np.float32)
        cnt = OracleCounters()
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


def reference_rate(scene_text: str, budget_s: float = 10.0, first_sample: int = 0):
    """Throughput of the REFERENCE'S OWN MegaPath code on this scene, one host thread: oracle/_ref/libref.so = /root/reference/src
    compiled in place against the scalar LuisaCompute stand-in (oracle/Makefile.ref), driven sample by sample (its Li() through
    ref_li, the entry tests/test_oracle_vs_ref.py pins the oracle with).  None where libref.so is absent.  This is the reference's
    SOURCE on a scalar interpreter of its DSL, not its LLVM / CPU backend: the figure says what a sample costs when every DSL
    statement is one C++ statement, nothing about LuisaCompute's code generation.  bench.py's cpu_baseline leg only."""
    import ctypes as C
    import tempfile
    ref_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
    path = os.path.join(ref_dir, "libref.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.ref_scene_load.restype = C.c_void_p
    lib.ref_scene_load.argtypes = [C.c_char_p, C.c_char_p]
    lib.ref_scene_destroy.argtypes = [C.c_void_p]
    lib.ref_resolution.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_li.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
    with tempfile.TemporaryDirectory() as tmp:
        scene_path = os.path.join(tmp, "scene.luisa")
        with open(scene_path, "w") as f:
            f.write(scene_text)
        handle = lib.ref_scene_load(scene_path.encode(), ref_dir.encode())
        if not handle:
            return None
        res = (C.c_uint32 * 2)()
        lib.ref_resolution(handle, res)
        width, height = int(res[0]), int(res[1])
        out = (C.c_float * 3)()
        # a regular grid of pixels over the whole frame, sample after sample, until the budget is spent
        step = max(1, min(width, height) // 32)
        pixels = [(x, y) for y in range(step // 2, height, step) for x in range(step // 2, width, step)]
        done, sample, t0 = 0, first_sample, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            for x, y in pixels:
                lib.ref_li(handle, x, y, sample, 0.0, out)
            done += len(pixels)
            sample += 1
        dt = time.perf_counter() - t0
        lib.ref_scene_destroy(handle)
    return {"value": done / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's own MegaPath::Li (oracle/_ref: /root/reference/src compiled in place on a scalar LuisaCompute stand-in, NOT its "
                      f"LLVM backend), {len(pixels)} pixels on a regular grid of the {width}x{height} frame x {sample - first_sample} samples = {done} paths in {dt:.1f} s, 1 thread"}
