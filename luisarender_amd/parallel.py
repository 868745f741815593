"""Screen-tile sharding of one frame over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for the CPU tests).
Tiles are 8x8 pixels, numbered row-major; rank r of W renders tiles {r, r + W, r + 2W, ...}
(round-robin keeps the load balanced without any exchange).  Every pixel is owned by exactly one
rank and non-owned pixels of a rank's film stay exactly zero, so the single collective of the path —
a sum-reduce of the float4 film to rank 0 over xGMI — reproduces the 1-GPU film bit for bit.
"""
from __future__ import annotations

import numpy as np


def tile_grid(width: int, height: int) -> tuple[int, int]:
    return (width + 7) // 8, (height + 7) // 8


def owned_tiles(width: int, height: int, rank: int, world: int) -> range:
    tx, ty = tile_grid(width, height)
    return range(rank, tx * ty, world)


def tile_rect(width: int, height: int, tile: int) -> tuple[int, int, int, int]:
    tx, _ = tile_grid(width, height)
    x0, y0 = (tile % tx) * 8, (tile // tx) * 8
    return x0, y0, min(x0 + 8, width), min(y0 + 8, height)


def owner_mask(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """bool[H, W]: pixels owned by `rank`"""
    tx, _ = tile_grid(width, height)
    ys, xs = np.mgrid[0:height, 0:width]
    tile = (ys // 8) * tx + xs // 8
    return (tile % world) == rank


def reduce_film(film, dst: int = 0):
    """The path's only collective: sum-reduce of the per-rank films to `dst` (in place)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
