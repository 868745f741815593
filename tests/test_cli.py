"""Process-level contract of luisa-render-cli (reference src/apps/cli.cpp:59-185): option parsing, help / exit
codes, plugin lookup by `luisa-render-integrator-<impl>` and, on a GPU box, the whole drop-in path scene file -> plugin ->
C ABI -> gfx950 megakernel -> EXR identical to what the Python driver produces through the same ABI."""
import os
import subprocess

import numpy as np
import pytest

from luisarender_amd import Scene
from luisarender_amd.scenes import cornell_box

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "luisarender_amd", "bin", "luisa-render-cli")


def _run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, cwd=cwd, timeout=600)


def test_cli_without_a_scene_prints_help_and_fails():
    r = _run("-b", "hip")
    assert r.returncode == 255 and "Usage:" in r.stdout and "Scene file not specified" in r.stderr  # exit(-1), cli.cpp:96-99
    assert _run("--help").returncode == 0


def test_cli_plugins_exist_for_every_integrator_the_loader_accepts():
    for impl in ("megapath", "direct", "normal", "megavptnaive"):  # `luisa-render-<tag>-<impl>` next to the executable, scene.cpp:54-75
        assert os.path.exists(os.path.join(ROOT, "luisarender_amd", "bin", f"libluisa-render-integrator-{impl}.so"))


def test_cli_reports_scene_errors_and_unknown_options(tmp_path):
    bad = tmp_path / "bad.luisa"
    bad.write_text("render { cameras { } shapes { } integrator : WavePath { } }")
    r = _run("-b", "hip", "--frobnicate", str(bad))
    assert r.returncode != 0 and "Unrecognized options: --frobnicate" in r.stderr and "[error]" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["MegaPath", "Direct", "Normal", "MegaVPTNaive"])
def test_cli_renders_the_same_image_as_the_c_abi(tmp_path, integrator):
    from luisarender_amd.render import MegaPathRenderer
    from luisarender_amd.scene import load_image
    text = cornell_box(resolution=64, spp="#spp", file="out.exr").replace("integrator : MegaPath {", f"integrator : {integrator} {{")
    scene_file = tmp_path / "cornell.luisa"
    scene_file.write_text(text)
    r = _run("-b", "hip", "-d", "0", "-D", "spp=8", str(scene_file))
    assert r.returncode == 0, r.stderr
    assert "Rendering finished in" in r.stderr  # integrator.cpp:112
    img, _ = load_image(str(tmp_path / "out.exr"))
    sc = Scene.load(str(scene_file), macros={"spp": 8})
    renderer = MegaPathRenderer(0)
    renderer.upload(sc)
    renderer.render(0, 8, sync=True)
    ref = renderer.download(converted=True)
    renderer.close()
    assert img.shape == ref.shape and np.array_equal(img, ref)  # same kernel, same chunking: bit-identical
    assert img[..., :3].mean() > 0.01 and (img[..., 3] == 1).all()


@pytest.mark.gpu
def test_cli_renders_motion_blur_like_the_c_abi(tmp_path):
    """the plugin's loop over shutter samples (integrator.cpp:86-107) against MegaPathRenderer.render_frame"""
    from luisarender_amd.render import MegaPathRenderer
    from luisarender_amd.scene import load_image
    from test_motion_blur import MOVING_STRIP
    scene_file = tmp_path / "strip.luisa"
    scene_file.write_text(MOVING_STRIP.replace("SPP", "16").replace("SHUTTER", "4").replace("spp { 16 }", 'spp { 16 } file { "blur.exr" }'))
    r = _run("-b", "hip", str(scene_file))
    assert r.returncode == 0, r.stderr
    img, _ = load_image(str(tmp_path / "blur.exr"))
    sc = Scene.load(str(scene_file))
    renderer = MegaPathRenderer(0)
    renderer.render_frame(sc)
    ref = renderer.download(converted=True)
    renderer.close()
    assert np.array_equal(img, ref)
    assert (img[..., 0] > 0.01).mean() > 0.3  # the streak of the moving strip


@pytest.mark.gpu
def test_cli_shards_a_frame_over_several_gpus(tmp_path):
    """The C++ host's multi-GPU path (plugin_megapath.cpp: one host thread + one lrhip_ctx per GPU, ncclCommInitAll, the tiles
    {r, r + W, ...}, lrhip_film_reduce to the first device) against the same frame rendered on one GPU with the same work-item
    sizing (the CLI sizes its work items for 8 shards whatever the device count, so `-d 0` and `-d 0,1` write the same bits): bit
    for bit.  With a single visible GPU the list `-d 0` / LR_DEVICES=0 still goes through the threaded path (world 1) and
    LR_FORCE_COLLECTIVE=1 makes it create its RCCL communicator and run the reduce with one rank -- the standalone binary has no
    torch beside it and must find librccl on its own; world 2 runs when the box has two GPUs (a 1-GPU box skips that half)."""
    import ctypes as C
    from luisarender_amd import _ffi
    from luisarender_amd.render import MegaPathRenderer
    from luisarender_amd.scene import load_image
    text = cornell_box(resolution=(96, 64), spp=8, file="out.exr")
    scene_file = tmp_path / "cornell.luisa"
    scene_file.write_text(text)
    n = C.c_int()
    assert _ffi.hip_lib().lrhip_device_count(C.byref(n)) == 0 and n.value >= 1
    sc = Scene.load(str(scene_file))
    for world in (1, 2):
        if world > n.value:
            pytest.skip(f"{n.value} GPU(s) visible: the world-{world} half needs {world}")
        env = dict(os.environ, LR_DEVICES=",".join(str(d) for d in range(world)), LR_FORCE_COLLECTIVE="1")
        r = subprocess.run([CLI, "-b", "hip", str(scene_file)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and f"on {world} HIP device(s)" in r.stderr and "films reduced over RCCL" in r.stderr, r.stderr
        img, _ = load_image(str(tmp_path / "out.exr"))
        one = MegaPathRenderer(0)
        one.upload(sc)
        one.render(0, 8, balance_shards=8, sync=True)
        ref = one.download(converted=True)
        one.close()
        assert np.array_equal(img.reshape(ref.shape), ref), world
