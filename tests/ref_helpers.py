"""ctypes handle on oracle/_ref/libref.so: the REFERENCE's own render code compiled in place (oracle/Makefile.ref) against the
scalar LuisaCompute stand-in of oracle/ref_shim.  Test infrastructure; only tests/ and tests/golden/make_ref_golden.py use it."""
import ctypes as C
import os
import tempfile

import numpy as np

REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
LIBREF = os.path.join(REF_DIR, "libref.so")


def available() -> bool:
    return os.path.exists(LIBREF)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIBREF, mode=C.RTLD_GLOBAL)
        u32, f32, vp, cp = C.c_uint32, C.c_float, C.c_void_p, C.c_char_p
        L.ref_scene_load.restype = vp
        L.ref_scene_load.argtypes = [cp, cp]
        L.ref_scene_destroy.argtypes = [vp]
        L.ref_render.argtypes = [vp, vp]
        L.ref_resolution.argtypes = [vp, vp]
        L.ref_li.argtypes = [vp, u32, u32, u32, f32, vp]
        L.ref_camera_ray.argtypes = [vp, u32, u32, u32, f32, vp]
        L.ref_sampler_stream.argtypes = [vp, u32, u32, u32, u32, vp]
        L.ref_filter_sample.argtypes = [vp, f32, f32, vp]
        L.ref_filter_tables.argtypes = [vp, vp, vp, vp, vp]
        L.ref_trace_closest.argtypes = [vp, vp, vp, f32, f32, vp, vp]
        for name in ("ref_instance_count", "ref_surface_count", "ref_light_count"):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = [vp]
        L.ref_instance_handle.argtypes = [vp, u32, vp]
        L.ref_interaction.argtypes = [vp, u32, u32, f32, f32, vp, vp]
        L.ref_surface_evaluate.argtypes = [vp, u32, vp, vp, vp, vp]
        L.ref_surface_sample.argtypes = [vp, u32, vp, vp, f32, f32, f32, vp]
        L.ref_light_sample.argtypes = [vp, vp, vp, f32, f32, f32, vp]
        L.ref_evaluate_miss.argtypes = [vp, vp, vp]
        for name, n in (("ref_xxhash32_1", 1), ("ref_xxhash32_2", 2), ("ref_xxhash32_3", 3), ("ref_xxhash32_4", 4), ("ref_pcg", 1)):
            getattr(L, name).restype = u32
            getattr(L, name).argtypes = [u32] * n
        L.ref_pcg4d.argtypes = [vp, vp]
        L.ref_lcg.restype = f32
        L.ref_lcg.argtypes = [vp]
        L.ref_pcg32_seed.argtypes = [C.c_uint64, vp, vp]
        L.ref_pcg32_next.restype = u32
        L.ref_pcg32_next.argtypes = [vp, vp]
        L.ref_create_alias_table.argtypes = [vp, u32, vp, vp, vp]
        L.ref_sample_alias_table.argtypes = [vp, vp, u32, f32, vp, vp]
        for name in ("ref_sample_uniform_triangle", "ref_sample_cosine_hemisphere", "ref_sample_uniform_sphere"):
            getattr(L, name).argtypes = [f32, f32, vp]
        L.ref_sample_uniform_cone.argtypes = [f32, f32, f32, vp]
        for name in ("ref_balance_heuristic", "ref_power_heuristic"):
            getattr(L, name).restype = f32
            getattr(L, name).argtypes = [f32, f32]
        L.ref_fresnel_dielectric.restype = f32
        L.ref_fresnel_dielectric.argtypes = [f32, f32, f32]
        L.ref_fresnel_conductor.argtypes = [f32, f32, vp, vp, vp]
        L.ref_fresnel_dielectric_integral.restype = f32
        L.ref_fresnel_dielectric_integral.argtypes = [f32]
        L.ref_refract.restype = C.c_int
        L.ref_refract.argtypes = [vp, vp, f32, vp]
        L.ref_ggx.argtypes = [vp, vp, vp, vp, vp]
        L.ref_ggx_sample_wh.argtypes = [vp, vp, f32, f32, vp]
        L.ref_frame_make.argtypes = [vp, vp, vp]
        L.ref_clamp_shading_normal.argtypes = [vp, vp, vp, vp]
        L.ref_encode_handle.argtypes = [u32, u32, u32, u32, u32, u32, f32, f32, vp]
        _lib = L
    return _lib


def f32a(*v):
    return np.ascontiguousarray(np.array(v, np.float32).ravel())


class RefScene:
    """A scene parsed by the reference's own parser (src/sdl/scene_parser.cpp), instantiated by its own plugins and turned
    into its own Pipeline (src/base/pipeline.cpp:44-99)."""

    def __init__(self, source: str, directory: str | None = None):
        self._tmp = None
        if directory is None:
            self._tmp = tempfile.TemporaryDirectory()
            directory = self._tmp.name
        self.path = os.path.join(directory, "scene.luisa")
        with open(self.path, "w") as f:
            f.write(source)
        self.L = lib()
        self.h = self.L.ref_scene_load(self.path.encode(), REF_DIR.encode())
        res = np.zeros(2, np.uint32)
        self.L.ref_resolution(self.h, res.ctypes.data)
        self.width, self.height = int(res[0]), int(res[1])

    def close(self):
        if self.h:
            self.L.ref_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def li(self, px, py, sample, time=0.0):
        out = np.zeros(3, np.float32)
        self.L.ref_li(self.h, px, py, sample, time, out.ctypes.data)
        return out

    def render(self):
        """the reference's whole frame loop (src/base/integrator.cpp:34-113) -> converted film float[H, W, 4]"""
        out = np.zeros((self.height, self.width, 4), np.float32)
        assert self.L.ref_render(self.h, out.ctypes.data) == 0
        return out

    def camera_ray(self, px, py, sample, time=0.0):
        out = np.zeros(7, np.float32)
        self.L.ref_camera_ray(self.h, px, py, sample, time, out.ctypes.data)
        return out

    def sampler_stream(self, px, py, sample, n):
        out = np.zeros(n + 2, np.float32)
        self.L.ref_sampler_stream(self.h, px, py, sample, n, out.ctypes.data)
        return out

    def filter_sample(self, ux, uy):
        out = np.zeros(3, np.float32)
        self.L.ref_filter_sample(self.h, ux, uy, out.ctypes.data)
        return out

    def filter_tables(self):
        lut, pdf, prob, idx = np.zeros(64, np.float32), np.zeros(63, np.float32), np.zeros(63, np.float32), np.zeros(63, np.uint32)
        self.L.ref_filter_tables(self.h, lut.ctypes.data, pdf.ctypes.data, prob.ctypes.data, idx.ctypes.data)
        return lut, pdf, prob, idx

    def trace_closest(self, o, d, t_min=0.0, t_max=3.0e38):
        o, d = f32a(*o), f32a(*d)
        ids, bary = np.zeros(2, np.uint32), np.zeros(2, np.float32)
        self.L.ref_trace_closest(self.h, o.ctypes.data, d.ctypes.data, t_min, t_max, ids.ctypes.data, bary.ctypes.data)
        return int(ids[0]), int(ids[1]), bary

    def instance_handles(self):
        n = self.L.ref_instance_count(self.h)
        out = np.zeros((n, 4), np.uint32)
        for i in range(n):
            self.L.ref_instance_handle(self.h, i, out[i].ctypes.data)
        return out

    def surface_evaluate(self, inst, ns, wo, wi):
        ns, wo, wi = f32a(*ns), f32a(*wo), f32a(*wi)
        out = np.zeros(4, np.float32)
        self.L.ref_surface_evaluate(self.h, inst, ns.ctypes.data, wo.ctypes.data, wi.ctypes.data, out.ctypes.data)
        return out

    def surface_sample(self, inst, ns, wo, u_lobe, ux, uy):
        ns, wo = f32a(*ns), f32a(*wo)
        out = np.zeros(8, np.float32)
        self.L.ref_surface_sample(self.h, inst, ns.ctypes.data, wo.ctypes.data, u_lobe, ux, uy, out.ctypes.data)
        return out

    def light_sample(self, p, n, u_sel, ux, uy):
        p, n = f32a(*p), f32a(*n)
        out = np.zeros(11, np.float32)
        self.L.ref_light_sample(self.h, p.ctypes.data, n.ctypes.data, u_sel, ux, uy, out.ctypes.data)
        return out

    def evaluate_miss(self, wi):
        wi = f32a(*wi)
        out = np.zeros(4, np.float32)
        self.L.ref_evaluate_miss(self.h, wi.ctypes.data, out.ctypes.data)
        return out
