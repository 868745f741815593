"""Scene loading through liblrhost.so (host C ABI, include/lrhost.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


class HostError(RuntimeError):
    pass


class Scene:
    """A parsed + flattened scene (lrhost_scene) and its POD view (lr_scene)."""

    def __init__(self, handle: C.c_void_p):
        self._lib = _ffi.host_lib()
        self._handle = handle
        self._views: dict[int, _ffi.Scene] = {}

    @staticmethod
    def _macros(macros):
        macros = macros or {}
        keys = (C.c_char_p * len(macros))(*[k.encode() for k in macros])
        vals = (C.c_char_p * len(macros))(*[str(v).encode() for v in macros.values()])
        return keys, vals, len(macros)

    @classmethod
    def load(cls, path: str, macros: dict | None = None, build_accel: bool = True) -> "Scene":
        lib = _ffi.host_lib()
        keys, vals, n = cls._macros(macros)
        handle = C.c_void_p()
        if lib.lrhost_scene_load_file(path.encode(), keys, vals, n, C.byref(handle)) != 0:
            raise HostError(lib.lrhost_last_error().decode())
        scene = cls(handle)
        if build_accel:
            scene.build_accel()
        return scene

    @classmethod
    def from_string(cls, source: str, virtual_path: str = "", macros: dict | None = None, json: bool = False,
                    build_accel: bool = True) -> "Scene":
        lib = _ffi.host_lib()
        keys, vals, n = cls._macros(macros)
        handle = C.c_void_p()
        rc = lib.lrhost_scene_load_string(source.encode(), virtual_path.encode(), 1 if json else 0, keys, vals, n,
                                          C.byref(handle))
        if rc != 0:
            raise HostError(lib.lrhost_last_error().decode())
        scene = cls(handle)
        if build_accel:
            scene.build_accel()
        return scene

    def build_accel(self) -> None:
        if self._lib.lrhost_scene_build_accel(self._handle) != 0:
            raise HostError(self._lib.lrhost_last_error().decode())
        self._views.clear()

    def set_time(self, time: float) -> bool:
        """Pipeline::update (src/base/pipeline.cpp:101-113): move every animated transform to `time`; the tables behind
        view() change in place (upload / create the oracle again).  -> whether anything moved"""
        updated = C.c_int(0)
        if self._lib.lrhost_scene_set_time(self._handle, time, C.byref(updated)) != 0:
            raise HostError(self._lib.lrhost_last_error().decode())
        self._views.clear()
        return bool(updated.value)

    def shutter_samples(self, camera: int = 0) -> list[tuple[float, float, int]]:
        """Camera::shutter_samples (src/base/camera.cpp:163-203) -> [(time, weight, spp)]"""
        out = []
        for i in range(self._lib.lrhost_scene_shutter_sample_count(self._handle, camera)):
            t, w, n = C.c_float(), C.c_float(), C.c_uint32()
            if self._lib.lrhost_scene_shutter_sample(self._handle, camera, i, C.byref(t), C.byref(w), C.byref(n)) != 0:
                raise HostError(self._lib.lrhost_last_error().decode())
            out.append((t.value, w.value, n.value))
        return out

    @property
    def camera_count(self) -> int:
        return self._lib.lrhost_scene_camera_count(self._handle)

    def view(self, camera: int = 0) -> _ffi.Scene:
        if camera not in self._views:
            v = _ffi.Scene()
            if self._lib.lrhost_scene_view(self._handle, camera, C.byref(v)) != 0:
                raise HostError(self._lib.lrhost_last_error().decode())
            v._owner = self  # keep the host tables alive as long as the view is referenced
            self._views[camera] = v
        return self._views[camera]

    def camera_file(self, camera: int = 0) -> str:
        return self._lib.lrhost_scene_camera_file(self._handle, camera).decode()

    @property
    def has_lighting(self) -> bool:
        return bool(self._lib.lrhost_scene_has_lighting(self._handle))

    def resolution(self, camera: int = 0) -> tuple[int, int]:
        v = self.view(camera)
        return int(v.camera.width), int(v.camera.height)

    def close(self) -> None:
        if self._handle:
            self._lib.lrhost_scene_destroy(self._handle)
            self._handle = None
            self._views.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_image(path: str, rgba: np.ndarray) -> None:
    """save_image of the reference (src/util/imageio.cpp:694-726): float RGBA -> .exr / .hdr"""
    lib = _ffi.host_lib()
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w = rgba.shape[:2]
    if lib.lrhost_save_image(path.encode(), rgba.ctypes.data, w, h) != 0:
        raise HostError(lib.lrhost_last_error().decode())


def load_image(path: str):
    """LoadedImage::load of the reference (src/util/imageio.cpp:419-538) -> (float RGBA array [H, W, 4], channels)"""
    import ctypes as C
    lib = _ffi.host_lib()
    lib.lrhost_load_image.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.lrhost_free.argtypes = [C.c_void_p]
    ptr = C.POINTER(C.c_float)()
    w, h, ch = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if lib.lrhost_load_image(path.encode(), C.byref(ptr), C.byref(w), C.byref(h), C.byref(ch)) != 0:
        raise HostError(lib.lrhost_last_error().decode())
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 4)).copy(), ch.value
    finally:
        lib.lrhost_free(ptr)
