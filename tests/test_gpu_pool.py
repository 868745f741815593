"""GPU tests of the path-pool scheduler of round 4 (csrc/hip/megapool_kernel.h, lrhip_set_scheduler mode 2): two path contexts per lane,
work items overlapping inside a wave, the film summed in 64-bit fixed point.

The pool kernels run the SAME shading block on the same random numbers as the one-path-per-lane kernels, so a path's value is the same to
the last bit; what differs is the ORDER in which a pixel's samples are added -- fp32 adds in the wave's own order there, integer adds
here -- i.e. the last bits of a sum.  The one-path-per-lane kernels are held against the oracle (and through it the reference's own code) by
tests/test_gpu_parity.py and tests/test_ref_golden.py; these tests hold the pool kernels against them, against the oracle directly on the
Cornell box, and check what the fixed-point film promises: bit-identical frames run to run, under any sharding AND any work-item partition."""
import numpy as np
import pytest

from luisarender_amd import Scene
from luisarender_amd.scenes import cornell_box, generate_kitchen_scene, generate_room_scene
from oracle.check import Oracle

pytestmark = pytest.mark.gpu
POOL, WF = 4096, 1024  # LRHIP_FEAT_POOL, LRHIP_FEAT_WAVEFRONT


@pytest.fixture(scope="module")
def renderer():
    from luisarender_amd.render import MegaPathRenderer
    r = MegaPathRenderer(0)
    yield r
    r.close()


def _rel_l1(a, b):
    return float(np.abs(a[..., :3] - b[..., :3]).sum() / max(np.abs(b[..., :3]).sum(), 1e-20))


def _both(renderer, scene, spp, counters=False):
    out = {}
    try:
        for name, pool in (("lane", False), ("pool", True)):
            renderer.set_scheduler(pool)
            renderer.upload(scene)
            renderer.render(0, spp, counters=counters, sync=True)
            out[name] = (renderer.download(False), renderer.last_variant(), renderer.counters() if counters else None)
    finally:
        renderer.set_scheduler(None)
    return out


DISNEY = ('Surface paint : Disney { color : Constant { v { 0.8, 0.3, 0.2 } } metallic { 0.3 } roughness { 0.35 } clearcoat { 0.5 } '
          'sheen { 0.3 } specular_trans { 0.2 } }\n')
GLASS = 'Surface crystal : Glass { Kr : Constant { v { 0.95, 0.95, 0.95 } } Kt : Constant { v { 0.9, 0.95, 0.9 } } eta { "bk7" } roughness { 0.1 } }\n'


ALPHA = ('Texture holes : Checkerboard { on : Constant { v { 1 } } off : Constant { v { 0.2 } } scale { 3 } }\n'
         'Surface cutout : Matte { Kd : Constant { v { 0.7, 0.6, 0.2 } } alpha { @holes } }\n')


PADDED = 16384  # LRHIP_FEAT_PADDED_SOBOL: a pool kernel compiled for the PaddedSobol sampler (round 6)


@pytest.mark.parametrize("case", ["lean", "glass", "disney", "sobol", "padded_sobol", "padded_sobol_lens_rr", "padded_sobol_alpha", "padded_sobol_environment", "pcg", "mitchell", "rr",
                                  "alpha", "environment"])
def test_pool_kernels_render_the_frames_of_the_one_path_per_lane_kernels(renderer, case):
    """(padded_sobol*: the pool side is a kernel compiled for that sampler, which keeps only (sample index, pixel) of the stream and derives the
    dimension from the depth -- two for the pixel, two for a thin lens, six per vertex, one more from the roulette depth on -- against the
    run-time generic sampler of the one-path-per-lane kernel, which counts its draws: same numbers, same paths, same film)"""
    kw = dict(resolution=96, spp=24)
    if case == "glass":
        kw.update(extra_surfaces=GLASS, short_box_surface="crystal")
    elif case == "disney":
        kw.update(extra_surfaces=DISNEY, tall_box_surface="paint")
    elif case == "sobol":
        kw.update(sampler="Sobol")
    elif case == "padded_sobol":
        kw.update(sampler="PaddedSobol", extra_surfaces=GLASS, short_box_surface="crystal")
    elif case == "padded_sobol_lens_rr":
        kw.update(sampler="PaddedSobol", rr_depth=3, depth=12, extra_surfaces=GLASS, short_box_surface="crystal")
    elif case == "pcg":
        kw.update(sampler="PCG32")
    elif case == "mitchell":
        kw.update(filter_impl="Mitchell", filter_radius=2.0)  # negative lobes: negative samples (signed fixed-point adds)
    elif case == "rr":
        kw.update(rr_depth=2, depth=12)
    elif case in ("alpha", "padded_sobol_alpha"):  # the alpha-tested traversal: candidates parked for the test outside the loop, beside lanes that wait for a turnover
        kw.update(extra_surfaces=ALPHA, short_box_surface="cutout")
    if case in ("padded_sobol_alpha", "padded_sobol_environment"):
        kw.update(sampler="PaddedSobol")
    text = cornell_box(**kw)
    if case in ("environment", "padded_sobol_environment"):  # the <environment> variants: rays that leave through the open front are lit
        text = text.replace("render {", "render {\n  environment : Spherical { emission : Constant { v { 0.3, 0.4, 0.6 } } }")
    if case == "padded_sobol_lens_rr":
        text = text.replace("Camera cam : Pinhole {\n  fov { 39.3 }", "Camera cam : ThinLens {\n  fov { 39.3 } aperture { 2 } focal_length { 50 } focus_distance { 1000 }")
        assert "ThinLens" in text
    scene = Scene.from_string(text)
    out = _both(renderer, scene, 24, counters=True)
    (lane, v_lane, c_lane), (pool, v_pool, c_pool) = out["lane"], out["pool"]
    assert (v_pool & POOL) != 0 and (v_lane & POOL) == 0 and (v_pool & ~(POOL | PADDED)) == v_lane, (v_lane, v_pool)
    assert ((v_pool & PADDED) != 0) == case.startswith("padded_sobol"), v_pool
    assert np.isfinite(pool).all() and np.array_equal(pool[..., 3], lane[..., 3])
    # the same paths: every counter of the path topology is EQUAL (not close: same kernel code per vertex, same random numbers)
    for k in ("paths", "closest_rays", "shadow_rays", "surface_hits", "nee_samples", "path_length_sum"):
        assert c_pool[k] == c_lane[k], (k, c_pool[k], c_lane[k])
    # ... and the traversal work agrees (a lane's walk is the same sequence of steps under either scheduler; the 5 % are headroom from
    # round 4's leaf-batching experiment, whose postponed leaves let a few more boxes through -- that code is gone since round 5)
    for k in ("nodes_visited", "tris_tested"):
        assert abs(c_pool[k] - c_lane[k]) <= 0.05 * c_lane[k], (k, c_pool[k], c_lane[k])
    err = _rel_l1(pool, lane)
    print(f"{case}: pool vs one-path-per-lane rel-L1 {err:.2e}; pool lanes trace {c_pool['trace_steps_busy'] / c_pool['trace_steps']:.2f} "
          f"(one path per lane {c_lane['trace_steps_busy'] / c_lane['trace_steps']:.2f})")
    assert err < 1e-6  # fp32 sum in the wave's order vs exact integer sum rounded once: ~spp * 2^-24
    assert np.allclose(pool[..., :3], lane[..., :3], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("case", ["lean", "padded_sobol", "pcg", "environment", "env_disney", "wf_mix_alpha", "wf_layered_sobol"])
def test_shipped_pool_kernels_equal_their_counting_twins(renderer, tmp_path, case):
    """the tests above drive the COUNT twins of the pool kernels (they need the counters); bench.py and the CLI launch
    the binaries WITHOUT counters -- <4096> (C2), <4098> (C2 with a low-discrepancy sampler), <4100> (C3), <4116> (C4), <5128> / <7176>
    (C5's camera and continuation passes).  A shipped binary and its twin are the same template with `if (COUNT)` blocks but different
    BINARIES (round 1 saw one that kept its ray counts and emitted NaNs), so each shipped pool binary is held to its twin here: the same
    paths (equal sample counts), the same film -- the pool kernels sum in fixed point, so whatever differs is a path's VALUE (a
    differently contracted fp32 expression on a specular chain), not the order of the adds -- and the shipped binary deterministic."""
    from helpers import MATERIALS
    from test_environment import sky_image
    from luisarender_amd.scene import save_image

    def mat(*names):
        return "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in names)
    sky = str(tmp_path / "sky.exr")
    save_image(sky, sky_image())
    env = f'render {{\n  environment : Spherical {{ emission : Image {{ file {{ "{sky}" }} }} }}'
    text, variant = {
        "lean": (cornell_box(resolution=64, spp=8, short_box_surface="glass", tall_box_surface="metal", extra_surfaces=mat("glass", "metal")), POOL),
        "padded_sobol": (cornell_box(resolution=64, spp=8, short_box_surface="glass", extra_surfaces=mat("glass"), sampler="PaddedSobol"), POOL | 2 | PADDED),
        "pcg": (cornell_box(resolution=64, spp=8, sampler="PCG32"), POOL | 2),
        "environment": (cornell_box(resolution=64, spp=8).replace("render {", env), POOL | 4),
        "env_disney": (cornell_box(resolution=64, spp=8, short_box_surface="disney", extra_surfaces=mat("disney")).replace("render {", env), POOL | 4 | 16),
        # wavefront mode: the lean camera pass <5128> + the continuation pass <7176> around the heavy-closure kernel
        "wf_mix_alpha": (cornell_box(resolution=64, spp=8, short_box_surface="mix_nested", tall_box_surface="cutout", extra_surfaces=mat("mix_nested") + ALPHA), POOL | WF | 8 | 32),
        "wf_layered_sobol": (cornell_box(resolution=64, spp=8, short_box_surface="layered", tall_box_surface="cutout", extra_surfaces=mat("layered") + ALPHA, sampler="PaddedSobol"),
                             POOL | WF | 8 | 16 | 64 | 2 | PADDED),
    }[case]
    sc = Scene.from_string(text)
    films = []
    try:
        renderer.set_scheduler(True)
        for count in (True, False, False):
            renderer.upload(sc)
            renderer.render(0, 8, counters=count, sync=True)
            assert renderer.last_variant() == (variant | (1 if count else 0)), (case, renderer.last_variant())
            films.append(renderer.download(converted=False))
    finally:
        renderer.set_scheduler(None)
    twin, shipped, again = films
    assert twin[..., :3].sum() > 0 and np.isfinite(shipped).all() and np.array_equal(twin[..., 3], shipped[..., 3]), case
    assert np.array_equal(shipped, again), case  # the shipped binary twice: bit for bit
    if case.startswith("wf_layered"):  # (Layered seeds its walk from position bits: block means, as tests/test_gpu_parity.py does)
        def blocks(f):
            h, w = f.shape[0] // 8, f.shape[1] // 8
            return f[:h * 8, :w * 8, :3].reshape(h, 8, w, 8, 3).mean(axis=(1, 3))
        g, c = blocks(shipped), blocks(twin)
        err, bias = float(np.abs(g - c).sum() / np.abs(c).sum()), float(abs(g.mean() - c.mean()) / c.mean())
        print(f"{case}: shipped vs counting twin block rel-L1 {err:.3e}, mean {bias:.2e}")
        assert err < 0.15 and bias < 3e-2
    else:
        err, bias = _rel_l1(shipped, twin), float(abs(shipped[..., :3].mean() - twin[..., :3].mean()) / twin[..., :3].mean())
        print(f"{case}: shipped vs counting twin rel-L1 {err:.3e}, mean {bias:.2e}, bit-identical {bool(np.array_equal(shipped, twin))}")
        assert err < 3e-3 and bias < 1e-3, (case, err, bias)


def test_pool_kernels_against_the_oracle(renderer):
    scene = Scene.from_string(cornell_box(resolution=128, spp=16))
    try:
        renderer.set_scheduler(True)
        renderer.upload(scene)
        renderer.render(0, 16, counters=True, sync=True)
        gpu, gc = renderer.download(False), renderer.counters()
        assert renderer.last_variant() == (POOL | 1)
    finally:
        renderer.set_scheduler(None)
    cpu, cc = Oracle(scene).render(0, 16)
    assert gc["paths"] == cc["paths"]
    for k in ("closest_rays", "surface_hits", "nee_samples", "path_length_sum"):
        assert abs(gc[k] - cc[k]) <= max(1, 1e-5 * cc[k]), (k, gc[k], cc[k])
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and _rel_l1(gpu, cpu) < 1e-4  # (the bars of test_gpu_parity.py::test_cornell_same_paths_and_image)


def test_pool_films_are_bit_identical_under_any_sharding_and_any_work_item_partition(renderer, tmp_path):
    """Integer sums are associative: the frame does not depend on the order of its adds -- not on the run, not on how tiles are dealt
    to GPUs, and (what rounds 1-3 could not offer) not on how the frame is cut into work items: shards rendered with DIFFERENT
    balance_shards hints and item sizes still add up to the unsharded frame bit for bit."""
    scene = Scene.load(generate_room_scene(str(tmp_path), target_triangles=40_000, resolution=(200, 120), spp=12))  # (ragged: 200 x 120 is 25 x 15 tiles)
    try:
        renderer.set_scheduler(True)
        renderer.upload(scene)
        frames = []
        for _ in range(2):
            renderer.clear()
            renderer.render(0, 12, sync=True)
            frames.append(renderer.download(False))
        assert (renderer.last_variant() & POOL) != 0
        assert np.array_equal(frames[0], frames[1]) and np.isfinite(frames[0]).all() and (frames[0][..., 3] == 12).all()
        for world, hint, scale in ((3, 3, 0.0), (4, 1, 0.0), (2, 8, 0.3), (5, 2, 4.0)):
            renderer.set_diagnostics(item_scale=scale)
            total = np.zeros_like(frames[0])
            for rank in range(world):
                renderer.clear()
                renderer.render(0, 12, rank=rank, world=world, balance_shards=hint, sync=True)
                total += renderer.download(False)
            assert np.array_equal(total, frames[0]), (world, hint, scale)
        renderer.set_diagnostics()
        # progressive calls: each call's sums are rounded to fp32 once when they join the film
        renderer.clear()
        renderer.render(0, 5, sync=True)
        renderer.render(5, 12, sync=True)
        two = renderer.download(False)
        assert np.array_equal(two[..., 3], frames[0][..., 3]) and np.allclose(two, frames[0], rtol=1e-6, atol=1e-7)
    finally:
        renderer.set_diagnostics()
        renderer.set_scheduler(None)


def test_pool_kernels_in_wavefront_mode(renderer, tmp_path):
    """Mix / Layered scenes: the lean camera pass and the continuation pass as pool kernels around the same heavy-closure kernels."""
    scene = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(256, 144), spp=16, target_triangles=60_000))
    out = _both(renderer, scene, 16)
    (lane, v_lane, _), (pool, v_pool, _) = out["lane"], out["pool"]
    assert v_lane == (WF | 16 | 32 | 64) and v_pool == (v_lane | POOL)
    err = _rel_l1(pool, lane)
    print(f"wavefront mode, pool vs one-path-per-lane lean kernels: rel-L1 {err:.2e}")
    assert np.array_equal(pool[..., 3], lane[..., 3]) and (pool[..., 3] == 16).all() and np.isfinite(pool).all()
    assert err < 1e-6
    try:  # bit-identical run to run and under sharding, like everything that adds in fixed point
        renderer.set_scheduler(True)
        renderer.upload(scene)
        total = np.zeros_like(pool)
        for rank in range(3):
            renderer.clear()
            renderer.render(0, 16, rank=rank, world=3, sync=True)
            total += renderer.download(False)
        assert np.array_equal(total, pool)
    finally:
        renderer.set_scheduler(None)


def test_padded_sobol_kernels_in_wavefront_mode(renderer, tmp_path):
    """Round 6: the lean passes of wavefront mode compiled for the PaddedSobol sampler (kFeatPadded).  A parked path takes its stream position along in
    the generic sampler's four words, the heavy kernels (run-time sampler) draw a vertex's numbers, and what comes back to the pool's record is
    (sample index, pixel) only -- the dimension is derived from the depth again.  Against the one-path-per-lane passes, which count their draws:
    the same film (Layered walks included: same hit bits, same seeds), several slices with the hand-over between them."""
    scene = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(256, 144), spp=16, target_triangles=60_000, sampler="PaddedSobol"))
    try:
        renderer.set_wavefront(True, slice_paths=256 * 144 * 5)  # four slices, the last one short
        out = _both(renderer, scene, 16)
    finally:
        renderer.set_wavefront(True)
    (lane, v_lane, _), (pool, v_pool, _) = out["lane"], out["pool"]
    assert v_lane == (WF | 16 | 32 | 64 | 2) and v_pool == (v_lane | POOL | PADDED), (v_lane, v_pool)
    err = _rel_l1(pool, lane)
    print(f"wavefront mode under PaddedSobol, padded pool passes vs one-path-per-lane passes: rel-L1 {err:.2e}")
    assert np.array_equal(pool[..., 3], lane[..., 3]) and (pool[..., 3] == 16).all() and np.isfinite(pool).all()
    assert err < 1e-6


def test_a_slice_hands_its_parked_paths_over_to_the_next_one(renderer, tmp_path):
    """Round 6: in wavefront mode a slice runs ONE round { heavy kernels -> continuation pass } and leaves the paths still parked in their queues
    for the next slice's first round (film_kernels.h: wf_carry_kernel; the call's last slice runs all its rounds).  A path does not depend on
    its slice and the pool kernels' film is summed in fixed point: six slices with the hand-over after one, two or three rounds, with none at all
    (every slice drains: the round 3-5 behaviour), eight tiles at a time, and as three shards must be THE SAME film, every sample counted."""
    scene = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(256, 144), spp=16, target_triangles=60_000))
    films = {}
    try:
        renderer.set_scheduler(True)
        renderer.upload(scene)
        for name, kw in (("never", dict(carry_rounds=65535)), ("default", {}), ("two", dict(carry_rounds=2)), ("three", dict(carry_rounds=3)),
                         ("default, eight tiles at a time", dict(tiny_tile_groups=True)), ("one slice", None)):
            if kw is None:
                renderer.set_wavefront(True)
            else:
                renderer.set_wavefront(True, slice_paths=256 * 144 * 3, **kw)  # 3 spp per slice: six slices, the last one short
            renderer.clear()
            renderer.render(0, 16, sync=True)
            films[name] = renderer.download(False)
            assert renderer.last_variant() == (POOL | WF | 16 | 32 | 64), name
        renderer.set_wavefront(True, slice_paths=256 * 144 * 3)
        total = np.zeros_like(films["never"])
        for rank in range(3):
            renderer.clear()
            renderer.render(0, 16, rank=rank, world=3, sync=True)
            total += renderer.download(False)
        films["default, three shards"] = total
    finally:
        renderer.set_wavefront(True)
        renderer.set_scheduler(None)
    ref = films["never"]
    assert np.isfinite(ref).all() and (ref[..., 3] == 16).all() and ref[..., :3].sum() > 0
    for name, f in films.items():
        assert np.array_equal(f, ref), (name, _rel_l1(f, ref))
    # A scene whose paths mostly STAY parked: every wall of the box is a Mix surface, so after a slice's first round far more paths wait than the
    # queues' margin holds (an eighth of a slice) -- the hand-over must not happen until a later round has thinned them out, and nothing may be lost
    text = cornell_box(resolution=64, spp=16, depth=6)
    white = "Surface white : Matte { Kd : Constant { v { 0.725, 0.71, 0.68 } } }\n"
    assert white in text
    text = text.replace(white, "Surface wa : Matte { Kd : Constant { v { 0.8, 0.75, 0.7 } } }\nSurface wb : Matte { Kd : Constant { v { 0.65, 0.67, 0.66 } } sigma : Constant { v { 0.4 } } }\n"
                               "Surface white : Mix { a { @wa } b { @wb } ratio : Constant { v { 0.5 } } }\n")
    heavy = Scene.from_string(text)
    out = {}
    try:
        renderer.set_scheduler(True)
        renderer.upload(heavy)
        for name, kw in (("never", dict(carry_rounds=65535)), ("default", {})):
            renderer.set_wavefront(True, slice_paths=64 * 64 * 4, **kw)  # four slices
            renderer.clear()
            renderer.render(0, 16, sync=True)
            out[name] = renderer.download(False)
            assert renderer.last_variant() & (POOL | WF) == (POOL | WF), name
    finally:
        renderer.set_wavefront(True)
        renderer.set_scheduler(None)
    assert (out["never"][..., 3] == 16).all() and out["never"][..., :3].sum() > 0 and np.array_equal(out["default"], out["never"])


def test_frames_that_do_not_fit_fixed_point_take_the_float_kernels(renderer, tmp_path):
    """A film clamp used to switch clamping off (1e20) leaves no fractional bits in 64-bit fixed point: the pool kernels -- and wavefront
    mode, whose parked paths add in fixed point -- step aside for the float-accumulating kernels instead of quantising the frame
    (round 3 rendered such a wavefront frame in whole integers)."""
    big = Scene.from_string(cornell_box(resolution=64, spp=8).replace("film : Color {", "film : Color { clamp { 1e20 }"))
    ref = Scene.from_string(cornell_box(resolution=64, spp=8).replace("film : Color {", "film : Color { clamp { 1e6 }"))
    try:
        renderer.set_scheduler(True)
        renderer.upload(big)
        renderer.render(0, 8, sync=True)
        a = renderer.download(False)
        assert (renderer.last_variant() & POOL) == 0
        renderer.upload(ref)
        renderer.render(0, 8, sync=True)
        b = renderer.download(False)
        assert (renderer.last_variant() & POOL) != 0  # 1e6 x 8 spp still leaves 38 fractional bits
        assert np.allclose(a, b, rtol=2e-5, atol=1e-6)  # (nothing in a Cornell box reaches either clamp)
    finally:
        renderer.set_scheduler(None)
    kitchen = generate_kitchen_scene(str(tmp_path), resolution=(128, 72), spp=8, target_triangles=30_000)
    text = open(kitchen).read().replace("film : Color {", "film : Color { clamp { 1e20 }")
    path = tmp_path / "kitchen_noclamp.luisa"
    path.write_text(text)
    renderer.upload(Scene.load(str(path)))
    renderer.render(0, 8, sync=True)
    f = renderer.download(False)
    assert (renderer.last_variant() & WF) == 0 and np.isfinite(f).all() and (f[..., 3] == 8).all()
    assert float(np.abs(f[..., :3] - np.round(f[..., :3])).max()) > 1e-3  # not whole numbers


def test_a_call_beyond_the_fixed_point_range_is_rendered_in_sample_sub_ranges(renderer):
    """film clamp x spp beyond 2^37 (a clamp of 1e9 at 1024 spp here) used to send the whole call to the float-accumulating
    kernels -- a Mix / Layered scene silently out of wavefront mode.  lrhip_render now renders such a call in sample sub-ranges that fit
    (128 spp each here), one resolve per range: same kernel family, same samples, the same film as the caller's own progressive calls."""
    from helpers import MATERIALS
    surf = MATERIALS["mix_nested"].replace("Surface m ", "Surface mix_nested ") + "\n"
    text = cornell_box(resolution=32, spp=1024, short_box_surface="mix_nested", extra_surfaces=surf).replace("film : Color {", "film : Color { clamp { 1e9 }")
    scene = Scene.from_string(text)
    try:
        renderer.set_scheduler(True)
        renderer.upload(scene)
        renderer.render(0, 1024, sync=True)
        whole = renderer.download(False)
        assert (renderer.last_variant() & (POOL | WF)) == (POOL | WF), renderer.last_variant()
        assert np.isfinite(whole).all() and (whole[..., 3] == 1024).all() and renderer.last_render_ms() > 0
        renderer.upload(scene)
        for s in range(0, 1024, 128):  # the sub-ranges by hand
            renderer.render(s, s + 128, sync=True)
        parts = renderer.download(False)
        assert np.array_equal(whole, parts)
    finally:
        renderer.set_scheduler(None)


def test_scheduler_argument_is_checked(renderer):
    import ctypes as C
    assert renderer._lib.lrhip_set_scheduler(renderer._ctx, 3) != 0 and b"lrhip_set_scheduler" in renderer._lib.lrhip_last_error()
    assert renderer._lib.lrhip_set_scheduler(C.c_void_p(), 0) != 0


def test_the_overflow_area_of_the_traversal_stack(tmp_path):
    """`make shallow`: the lean kernels of both schedulers with a four-entry LDS stack, so that EVERY ray goes through the HBM overflow
    area (push / pop beyond the LDS part, the wave-level `deep` paths, the five words a pool kernel parks on top of a lane's stack
    across the shading block -- which the shipped sixteen entries reach on a few rays of a few scenes only).  Same traversal order,
    same sums: the frames must be the shipped library's bit for bit."""
    import os
    from luisarender_amd import _ffi as ffi
    from luisarender_amd.render import MegaPathRenderer
    lib = os.path.join(ffi.LIB_DIR, "variants", "liblrhip_shallow.so")
    assert os.path.exists(lib), "make shallow (python __graft_entry__.py builds it)"
    scenes = [Scene.from_string(cornell_box(resolution=96, spp=8)),
              Scene.load(generate_room_scene(str(tmp_path), target_triangles=60_000, resolution=(160, 96), spp=6))]
    for scene, spp in zip(scenes, (8, 6)):
        frames = {}
        for build, path in (("shipped", None), ("shallow", lib)):
            r = MegaPathRenderer(0, lib_path=path) if path else MegaPathRenderer(0)
            try:
                for pool in (False, True):
                    r.set_scheduler(pool)
                    r.upload(scene)
                    r.render(0, spp, counters=True, sync=True)
                    assert bool(r.last_variant() & POOL) == pool
                    frames[build, pool] = (r.download(False), r.counters())
            finally:
                r.close()
        for pool in (False, True):
            (a, ca), (b, cb) = frames["shipped", pool], frames["shallow", pool]
            assert np.array_equal(a, b), (pool, _rel_l1(b, a))
            for k in ("paths", "closest_rays", "shadow_rays", "surface_hits", "nodes_visited", "tris_tested"):
                assert ca[k] == cb[k], (pool, k)
