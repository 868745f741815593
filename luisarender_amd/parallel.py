"""Screen-tile sharding of one frame over the GPUs of a node.

One process per GPU (bench.py under torch.distributed.run) or one host thread per GPU (the C++ host, plugin_megapath.cpp).
Tiles are 8x8 pixels; tile number t = ty * tiles_x + j is the tile of row ty in column (j + ty) mod tiles_x (include/lrhip.h:
every row is rotated by its index), and rank r of W renders the numbers {r, r + W, r + 2W, ...}: diagonals of the frame, never
column stripes, balanced without any exchange.  Every pixel is owned by exactly one rank and the non-owned pixels of a rank's
film stay exactly zero, so the single collective of the path -- a sum-reduce of the float4 film to rank 0 over xGMI,
lrhip_film_reduce = ncclReduce on the context's stream -- reproduces the 1-GPU film bit for bit when both are rendered with the
same balance_shards value (the work-item chunking, and with it the fp32 summation order, is a function of it: lrhip.h).
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
    row = tile // tx
    x0, y0 = ((tile % tx + row) % tx) * 8, row * 8  # row `row` is rotated by `row`
    return x0, y0, min(x0 + 8, width), min(y0 + 8, height)


def owner_mask(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """bool[H, W]: pixels owned by `rank`"""
    tx, _ = tile_grid(width, height)
    ys, xs = np.mgrid[0:height, 0:width]
    row = ys // 8
    tile = row * tx + (xs // 8 - row) % tx
    return (tile % world) == rank


class FilmReducer:
    """The product's collective for one-process-per-GPU hosts: an RCCL communicator created through the C ABI
    (lrhip_comm_unique_id on rank 0, the 128 bytes broadcast over the process group the launcher already made,
    lrhip_comm_init_rank on every rank) and lrhip_film_reduce on the renderer's own stream.  bench.py times THIS."""

    def __init__(self, renderer, rank: int, world: int, force: bool = False):
        import torch
        import torch.distributed as dist
        self.renderer, self.comm = renderer, None
        if world <= 1 and not force:  # (force: a one-rank communicator, to run this very code on a 1-GPU box)
            return
        uid = torch.tensor(list(renderer.comm_unique_id()) if rank == 0 else [0] * 128, dtype=torch.uint8, device=f"cuda:{torch.cuda.current_device()}")
        dist.broadcast(uid, src=0)
        self.comm = renderer.comm_init_rank(world, rank, bytes(uid.cpu().tolist()))

    def reduce(self, dst: int = 0) -> None:
        if self.comm is not None:
            self.renderer.film_reduce(self.comm, dst)

    def info(self) -> dict | None:
        """ranks / rank / device as RCCL itself reports them for this communicator (lrhip_comm_info); None without one"""
        return self.renderer.comm_info(self.comm) if self.comm is not None else None

    def close(self) -> None:
        if self.comm is not None:
            self.renderer.comm_destroy(self.comm)
            self.comm = None


def reduce_film(film, dst: int = 0):
    """torch.distributed stand-in of the collective for the CPU tests (gloo, the oracle in the kernel's place)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
