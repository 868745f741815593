"""Thin Python driver over the device C ABI (include/lrhip.h).  No torch types cross the ABI;
torch is only used by callers that want the film in a tensor (bench.py, multi-GPU reduce)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi
from .scene import Scene


class DeviceError(RuntimeError):
    pass


def tile_count(width: int, height: int) -> int:
    return ((width + 7) // 8) * ((height + 7) // 8)


class MegaPathRenderer:
    """One lrhip_ctx on one GPU.  Mirrors the reference's ProgressiveIntegrator::Instance::render
    (src/base/integrator.cpp:34-49): prepare film -> render spp -> download (convert) -> save."""

    def __init__(self, device: int = 0, lib_path: str | None = None):
        self._lib = _ffi.hip_lib(lib_path)  # raises if liblrhip.so is missing: there is no CPU fallback
        self._ctx = C.c_void_p()
        self._check(self._lib.lrhip_create(device, C.byref(self._ctx)))
        self._scene = None
        self.width = self.height = 0

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise DeviceError(f"lrhip error {rc}: {self._lib.lrhip_last_error().decode()}")

    def set_stream(self, hip_stream: int | None) -> None:
        self._check(self._lib.lrhip_set_stream(self._ctx, C.c_void_p(hip_stream or 0)))

    def upload(self, scene: Scene, camera: int = 0, keep_film: bool = False) -> None:
        """keep_film: lrhip_update_scene (the next shutter sample of a frame: film and counters carry on)"""
        view = scene.view(camera)
        self._check((self._lib.lrhip_update_scene if keep_film else self._lib.lrhip_upload_scene)(self._ctx, C.byref(view)))
        self._scene = scene
        self.width, self.height = int(view.camera.width), int(view.camera.height)

    def bind_film(self, device_ptr: int | None) -> None:
        self._check(self._lib.lrhip_bind_film(self._ctx, C.c_void_p(device_ptr or 0)))

    def clear(self) -> None:
        self._check(self._lib.lrhip_film_clear(self._ctx))

    def render(self, spp_begin: int, spp_end: int, rank: int = 0, world: int = 1, counters: bool = False,
               sync: bool = False, balance_shards: int = 1, shutter_weight: float | None = None, tile_end: int | None = None) -> None:
        """Render samples [spp_begin, spp_end) of the round-robin tile shard `rank` of `world`.
        `balance_shards` sizes the work items for a frame split into that many shards (lrhip.h): films rendered with
        the same value are bit-identical under any sharding; the multi-GPU bench passes its world size."""
        p = _ffi.RenderParams()
        p.balance_shards = balance_shards
        p.spp_begin, p.spp_end = spp_begin, spp_end
        tiles = tile_count(self.width, self.height)
        tiles = min(tiles, tile_end) if tile_end is not None else tiles  # (tile_end: only the first tiles of the frame, for checks)
        p.tile_begin, p.tile_end, p.tile_stride = min(rank, tiles), tiles, world  # rank >= tiles: an empty shard
        p.flags = 1 if counters else 0
        if shutter_weight is not None:  # Camera::ShutterSample weight of these samples (integrator.cpp:74)
            p.flags |= 2
            p.shutter_weight = shutter_weight
        self._check(self._lib.lrhip_render(self._ctx, C.byref(p)))
        if sync:
            self.synchronize()

    def render_frame(self, scene: Scene, camera: int = 0, rank: int = 0, world: int = 1, balance_shards: int = 1) -> None:
        """ProgressiveIntegrator::Instance::_render_one_camera's loop over shutter samples (src/base/integrator.cpp:86-107):
        move the scene to each sample's time, upload it, render the sample's spp range with its weight."""
        samples = scene.shutter_samples(camera)
        begin = 0
        for i, (time, weight, spp) in enumerate(samples):
            moved = scene.set_time(time)
            if i == 0 or moved:
                self.upload(scene, camera, keep_film=i > 0)
            self.render(begin, begin + spp, rank=rank, world=world, balance_shards=balance_shards,
                        shutter_weight=weight if len(samples) > 1 else None)
            begin += spp
            if moved:
                self.synchronize()  # the next set_time rewrites the host tables the upload reads from

    def synchronize(self) -> None:
        self._check(self._lib.lrhip_synchronize(self._ctx))

    def download(self, converted: bool = True) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.lrhip_film_download(self._ctx, out.ctypes.data, 1 if converted else 0))
        return out

    # ---- the one collective of the multi-GPU path, through the C ABI
    def comm_unique_id(self) -> bytes:
        buf = (C.c_ubyte * 128)()
        self._check(self._lib.lrhip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init_rank(self, world: int, rank: int, unique_id: bytes) -> C.c_void_p:
        comm = C.c_void_p()
        self._check(self._lib.lrhip_comm_init_rank(self._ctx, world, rank, (C.c_ubyte * 128).from_buffer_copy(unique_id), C.byref(comm)))
        return comm

    def film_reduce(self, comm, root: int = 0) -> None:
        """sum-reduce of the bound film to `root` over RCCL, in stream order behind the renders (lrhip_film_reduce)"""
        self._check(self._lib.lrhip_film_reduce(self._ctx, comm, root))

    def comm_info(self, comm) -> dict:
        """lrhip_comm_info: what the communicator spans -- ranks, this rank, its device"""
        out = (C.c_int * 3)()
        self._check(self._lib.lrhip_comm_info(comm, out))
        return {"ranks": int(out[0]), "rank": int(out[1]), "device": int(out[2])}

    def comm_destroy(self, comm) -> None:
        self._check(self._lib.lrhip_comm_destroy(comm))

    def counters(self) -> dict:
        c = _ffi.HipCounters()
        self._check(self._lib.lrhip_get_counters(self._ctx, C.byref(c)))
        return c.as_dict()

    def last_render_ms(self) -> float:
        return float(self._lib.lrhip_last_render_ms(self._ctx))

    def last_variant(self) -> int:
        """feature mask of the precompiled megakernel variant the last render() launched (lrhip.h LRHIP_FEAT_*)"""
        return int(self._lib.lrhip_last_variant(self._ctx))

    def set_diagnostics(self, force_features: int = 0, item_scale: float = 0.0) -> None:
        """tests / tools only (lrhip_set_diagnostics): render with a larger precompiled variant than the scene needs, or sweep
        the work-item size; the library itself reads no environment variable"""
        self._check(self._lib.lrhip_set_diagnostics(self._ctx, force_features, item_scale))

    def set_wavefront(self, enabled: bool = True, slice_paths: int = 0, tiny_tile_groups: bool = False, carry_rounds: int = 0) -> None:
        """lrhip_set_wavefront: scenes with Mix / Layered surfaces render in wavefront mode by default (lean megakernel + heavy-closure
        kernel + continuation pass); enabled=False keeps them on the all-in-one megakernel variants (A/B, tests); tiny_tile_groups
        sends eight tiles through the queues at a time (tests: what a GPU short of memory does)"""
        # carry_rounds: rounds before a slice hands its parked paths over to the next one (0 = the library's default, 65535 = never)
        self._check(self._lib.lrhip_set_wavefront(self._ctx, ((2 if tiny_tile_groups else 0) if enabled else 1) | (carry_rounds << 8), slice_paths))

    def set_texture_storage(self, mode: int = 1) -> None:
        """lrhip_set_texture_storage: 8-bit images as 8-bit texels on the device from the next upload on (0 never, 1 automatic: scenes
        whose images exceed 192 MB as floats, 2 every image that qualifies)"""
        self._check(self._lib.lrhip_set_texture_storage(self._ctx, mode))

    def packed_texels(self) -> int:
        """lrhip_packed_texels: how many texels of the uploaded scene the device holds as 8-bit codes"""
        return int(self._lib.lrhip_packed_texels(self._ctx))

    def set_scheduler(self, pool: bool | None = None) -> None:
        """lrhip_set_scheduler: None = automatic (the path-pool kernels of round 4 -- two path contexts per lane, fixed-point film sums,
        overlapping work items -- where they are the faster family (from ~100 thousand BVH triangles; lrhip.h), the one-path-per-lane kernels below), False = one path per
        lane everywhere, True = the pool kernels wherever one exists for the scene"""
        self._check(self._lib.lrhip_set_scheduler(self._ctx, 0 if pool is None else (2 if pool else 1)))

    def close(self) -> None:
        if self._ctx:
            self._lib.lrhip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
