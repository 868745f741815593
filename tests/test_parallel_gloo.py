"""N > 1 path on CPU: world_size-2 gloo processes shard the frame by screen tile exactly like
bench.py does on GPUs (luisarender_amd/parallel.py) and sum-reduce the float4 film to rank 0.
The oracle stands in for the kernel here (no GPU in the CPU suite); the GPU test
test_gpu_parity.py::test_tile_shards_reproduce_full_frame covers the same property on the device."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from luisarender_amd import Scene
    from oracle.check import Oracle
    from luisarender_amd.parallel import owned_tiles, reduce_film, tile_rect
    from luisarender_amd.scenes import cornell_box
    sc = Scene.from_string(cornell_box(resolution=(28, 20), spp=3))
    o = Oracle(sc)
    w, h = sc.resolution()
    film = np.zeros((h, w, 4), np.float32)
    for t in owned_tiles(w, h, rank, world):
        o.render(0, 3, rect=tile_rect(w, h, t), threads=1, film=film)
    tensor = torch.from_numpy(film)
    reduce_film(tensor, dst=0)
    if rank == 0:
        np.save(out_path, tensor.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_tile_shard_and_film_reduce_world2(tmp_path):
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from luisarender_amd import Scene
    from oracle.check import Oracle
    from luisarender_amd.scenes import cornell_box
    sc = Scene.from_string(cornell_box(resolution=(28, 20), spp=3))
    full, _ = Oracle(sc).render(0, 3)
    assert np.array_equal(np.load(out), full)  # bit-identical to the single-process frame


def test_owner_masks_partition_the_frame():
    from luisarender_amd.parallel import owned_tiles, owner_mask, tile_grid
    for w, h, world in [(28, 20, 2), (64, 64, 8), (1280, 720, 4), (17, 9, 3)]:
        total = np.zeros((h, w), np.int32)
        for r in range(world):
            total += owner_mask(w, h, r, world)
        assert (total == 1).all()
        tx, ty = tile_grid(w, h)
        assert sorted(t for r in range(world) for t in owned_tiles(w, h, r, world)) == list(range(tx * ty))
