"""ctypes mirrors of include/lr_scene.h and loaders for the in-tree shared libraries.

Python here is plumbing (tests, bench, graft entry): the product is liblrhost.so (C++ host
side) + liblrhip.so (HIP kernels) behind the C ABIs of include/lrhost.h and include/lrhip.h.
"""
from __future__ import annotations

import ctypes as C
import os

_ROOT = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_ROOT)
LIB_DIR = os.path.join(_ROOT, "lib")

u32, u64, i32, f32 = C.c_uint32, C.c_uint64, C.c_int32, C.c_float


class Vertex(C.Structure):
    _fields_ = [(n, f32) for n in ("px", "py", "pz", "nx", "ny", "nz", "u", "v")]


class Triangle(C.Structure):
    _fields_ = [("i0", u32), ("i1", u32), ("i2", u32)]


class AliasEntry(C.Structure):
    _fields_ = [("prob", f32), ("alias", u32)]


class UInt4(C.Structure):
    _fields_ = [("x", u32), ("y", u32), ("z", u32), ("w", u32)]


class LightHandle(C.Structure):
    _fields_ = [("instance_id", u32), ("light_tag", u32)]


class Mesh(C.Structure):
    _fields_ = [("vertex_offset", u32), ("vertex_count", u32), ("triangle_offset", u32), ("triangle_count", u32)]


class Instance(C.Structure):
    _fields_ = [("handle", UInt4), ("object_to_world", f32 * 16), ("visible", u32), ("pad", u32 * 3)]


class Texture(C.Structure):
    _fields_ = [("kind", u32), ("channels", u32), ("v", f32 * 4), ("width", u32), ("height", u32),
                ("texel_offset", u64), ("address", u32), ("filter", u32), ("encoding", u32),
                ("gamma", f32 * 3), ("uv_scale", f32 * 2), ("uv_offset", f32 * 2), ("scale", f32 * 4),
                ("child", i32 * 2), ("checker_scale", f32), ("pad", u32)]


class Surface(C.Structure):
    _fields_ = [("kind", u32), ("flags", u32), ("tex", i32 * 16), ("f", f32 * 8), ("u", u32 * 4),
                ("alpha_tex", i32), ("normal_tex", i32), ("normal_strength", f32), ("pad", u32)]


class Light(C.Structure):
    _fields_ = [("kind", u32), ("emission_tex", i32), ("scale", f32), ("two_sided", u32)]


class Environment(C.Structure):
    _fields_ = [("kind", u32), ("emission_tex", i32), ("scale", f32), ("compensate_mis", u32),
                ("world_to_env", f32 * 9), ("env_to_world", f32 * 9), ("map_width", u32), ("map_height", u32),
                ("alias", C.c_void_p), ("pdf", C.c_void_p), ("direction", f32 * 3), ("cos_half_angle", f32),
                ("visible", u32), ("child_scale", f32 * 2), ("child", u32 * 2), ("pad", u32)]


class Camera(C.Structure):
    _fields_ = [("kind", u32), ("width", u32), ("height", u32), ("spp", u32), ("camera_to_world", f32 * 16),
                ("tan_half_fov", f32), ("focus_distance", f32), ("lens_radius", f32), ("projected_pixel_size", f32),
                ("ortho_scale", f32), ("clip_near", f32), ("clip_far", f32), ("pad", u32)]


class Filter(C.Structure):
    _fields_ = [("radius", f32), ("shift", f32 * 2), ("pad", f32), ("lut", f32 * 64), ("pdf", f32 * 63),
                ("alias_prob", f32 * 63), ("alias_index", u32 * 63), ("pad2", u32 * 3)]


class Film(C.Structure):
    _fields_ = [("scale", f32 * 3), ("clamp", f32)]


class Sampler(C.Structure):
    _fields_ = [("kind", u32), ("seed", u32), ("spp", u32), ("scale", u32), ("sobol_matrices", C.c_void_p),
                ("vdc_sobol", C.c_void_p), ("vdc_sobol_inv", C.c_void_p), ("tile_size", u32 * 2), ("tile_jitter", u32), ("tile_pad", u32)]


class Integrator(C.Structure):
    _fields_ = [("max_depth", u32), ("rr_depth", u32), ("rr_threshold", f32), ("env_prob", f32),
                ("light_count", u32), ("kind", u32), ("flags", u32), ("environment_medium_tag", u32)]


class Medium(C.Structure):
    _fields_ = [("kind", u32), ("priority", u32), ("eta", f32), ("g", f32), ("sigma_a", f32 * 3), ("sigma_s", f32 * 3),
                ("le", f32 * 3), ("pad", u32 * 3)]


class Bvh4Node(C.Structure):
    _fields_ = [("lo_x", f32 * 4), ("lo_y", f32 * 4), ("lo_z", f32 * 4), ("hi_x", f32 * 4), ("hi_y", f32 * 4),
                ("hi_z", f32 * 4), ("child", u32 * 4), ("pad", u32 * 4)]


class BvhTriangle(C.Structure):
    _fields_ = [("v0", f32 * 3), ("inst", u32), ("e1", f32 * 3), ("prim", u32), ("e2", f32 * 3), ("flags", u32)]


class Accel(C.Structure):
    _fields_ = [("nodes", C.POINTER(Bvh4Node)), ("node_count", u32), ("triangles", C.POINTER(BvhTriangle)),
                ("triangle_count", u32), ("world_min", f32 * 3), ("world_max", f32 * 3)]


class Scene(C.Structure):
    _fields_ = [("vertices", C.POINTER(Vertex)), ("vertex_count", u64),
                ("triangles", C.POINTER(Triangle)), ("triangle_count", u64),
                ("tri_alias", C.POINTER(AliasEntry)), ("tri_pdf", C.POINTER(f32)),
                ("meshes", C.POINTER(Mesh)), ("mesh_count", u32),
                ("instances", C.POINTER(Instance)), ("instance_count", u32),
                ("light_instances", C.POINTER(LightHandle)), ("light_instance_count", u32),
                ("surfaces", C.POINTER(Surface)), ("surface_count", u32),
                ("lights", C.POINTER(Light)), ("light_count", u32),
                ("textures", C.POINTER(Texture)), ("texture_count", u32),
                ("texels", C.POINTER(f32)), ("texel_count", u64),
                ("environment", Environment), ("camera", Camera), ("filter", Filter), ("film", Film),
                ("sampler", Sampler), ("integrator", Integrator), ("accel", Accel),
                ("any_non_opaque", u32), ("environment_child_count", u32), ("environment_children", C.c_void_p),
                ("media", C.POINTER(Medium)), ("medium_count", u32), ("pad_media", u32)]


STRUCTS = {"lr_scene": Scene, "lr_vertex": Vertex, "lr_triangle": Triangle, "lr_alias_entry": AliasEntry,
           "lr_mesh": Mesh, "lr_instance": Instance, "lr_texture": Texture, "lr_surface": Surface,
           "lr_light": Light, "lr_environment": Environment, "lr_camera": Camera, "lr_filter": Filter,
           "lr_film": Film, "lr_sampler": Sampler, "lr_integrator": Integrator, "lr_bvh4_node": Bvh4Node,
           "lr_bvh_triangle": BvhTriangle, "lr_accel": Accel, "lr_light_handle": LightHandle, "lr_medium": Medium}


class HipCounters(C.Structure):
    _fields_ = [(n, u64) for n in ("paths", "closest_rays", "shadow_rays", "nodes_visited", "tris_tested",
                                   "surface_hits", "nee_samples", "path_length_sum", "trace_steps", "trace_steps_busy",
                                   "shade_calls", "shade_busy", "trace_steps_starved", "shade_cycles", "trace_cycles", "wave_cycles", "nodes_empty",
                                   "shade_light_cycles", "shade_closure_cycles", "shade_regen_cycles")] + [("probe", u64 * 16)]

    def as_dict(self):
        return {n: (int(getattr(self, n)) if n != "probe" else [int(x) for x in self.probe]) for n, _ in self._fields_}


class RenderParams(C.Structure):
    """lrhip_render_params (include/lrhip.h)"""
    _fields_ = [("spp_begin", u32), ("spp_end", u32), ("tile_begin", u32), ("tile_end", u32),
                ("tile_stride", u32), ("flags", u32), ("balance_shards", u32), ("shutter_weight", f32)]


_libs: dict[str, C.CDLL] = {}


def _load(path: str) -> C.CDLL:
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        _libs[path] = C.CDLL(path)
    return _libs[path]


def host_lib() -> C.CDLL:
    lib = _load(os.path.join(LIB_DIR, "liblrhost.so"))
    if not getattr(lib, "_lr_ready", False):
        lib.lrhost_last_error.restype = C.c_char_p
        lib.lrhost_scene_camera_file.restype = C.c_char_p
        lib.lrhost_sizeof.restype = u64
        lib.lrhost_sizeof.argtypes = [C.c_char_p]
        lib.lrhost_scene_destroy.argtypes = [C.c_void_p]
        lib.lrhost_scene_view.argtypes = [C.c_void_p, C.c_int, C.POINTER(Scene)]
        lib.lrhost_scene_build_accel.argtypes = [C.c_void_p]
        lib.lrhost_scene_set_time.argtypes = [C.c_void_p, f32, C.POINTER(C.c_int)]
        lib.lrhost_scene_shutter_sample_count.argtypes = [C.c_void_p, C.c_int]
        lib.lrhost_scene_shutter_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(f32), C.POINTER(f32), C.POINTER(u32)]
        lib.lrhost_scene_camera_count.argtypes = [C.c_void_p]
        lib.lrhost_scene_camera_file.argtypes = [C.c_void_p, C.c_int]
        lib.lrhost_scene_has_lighting.argtypes = [C.c_void_p]
        lib.lrhost_save_image.argtypes = [C.c_char_p, C.c_void_p, u32, u32]
        lib._lr_ready = True
    return lib


def hip_lib(path: str | None = None) -> C.CDLL:
    """The product's device library.  Fails loudly when the HIP extension is missing.  `path`: another build of the same library
    (tests: the IEEE-arithmetic build of the lean kernel; tools: A/B variants) beside the shipped one in the same process."""
    # One HIP runtime and one RCCL per process: PyTorch-ROCm bundles its own libamdhip64 / librccl (same sonames as /opt/rocm's,
    # older versions), and whichever is loaded first serves both.  A Python host shares device pointers and streams with torch
    # (bench.py's film tensor, torch.distributed's process group), so torch's copies must be the ones: import it BEFORE the
    # library pulls /opt/rocm's through its RUNPATH.  (Loaded the other way round, torch ran on a HIP runtime it was not built
    # for and the process died in a double free at exit.)  C / C++ hosts have no torch and use /opt/rocm's alone.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # LRHIP_LIB selects an experimental build variant (tools/ only); the default is the shipped library
    lib = _load(path or os.environ.get("LRHIP_LIB") or os.path.join(LIB_DIR, "liblrhip.so"))
    if not getattr(lib, "_lr_ready", False):
        lib.lrhip_last_error.restype = C.c_char_p
        lib.lrhip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.lrhip_destroy.argtypes = [C.c_void_p]
        lib.lrhip_upload_scene.argtypes = [C.c_void_p, C.POINTER(Scene)]
        lib.lrhip_update_scene.argtypes = [C.c_void_p, C.POINTER(Scene)]
        lib.lrhip_film_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.lrhip_comm_unique_id.argtypes = [C.c_void_p]
        lib.lrhip_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.lrhip_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        lib.lrhip_comm_destroy.argtypes = [C.c_void_p]
        lib.lrhip_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.lrhip_device_count.argtypes = [C.POINTER(C.c_int)]
        lib.lrhip_bind_film.argtypes = [C.c_void_p, C.c_void_p]
        lib.lrhip_film_clear.argtypes = [C.c_void_p]
        lib.lrhip_render.argtypes = [C.c_void_p, C.POINTER(RenderParams)]
        lib.lrhip_synchronize.argtypes = [C.c_void_p]
        lib.lrhip_film_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.lrhip_get_counters.argtypes = [C.c_void_p, C.POINTER(HipCounters)]
        lib.lrhip_last_render_ms.restype = C.c_double
        lib.lrhip_last_render_ms.argtypes = [C.c_void_p]
        lib.lrhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.lrhip_last_variant.restype = C.c_uint32
        lib.lrhip_last_variant.argtypes = [C.c_void_p]
        lib.lrhip_set_diagnostics.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
        lib.lrhip_set_wavefront.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        lib.lrhip_set_scheduler.argtypes = [C.c_void_p, C.c_uint32]
        lib.lrhip_set_texture_storage.argtypes = [C.c_void_p, C.c_uint32]
        lib.lrhip_packed_texels.restype = C.c_uint64
        lib.lrhip_packed_texels.argtypes = [C.c_void_p]
        lib._lr_ready = True
    return lib
