"""GPU parity tests proper: the HIP megakernel (through the C ABI of include/lrhip.h) against the
CPU oracle on the same seeded inputs.  Sampler streams, alias tables and hit decisions are shared,
so GPU and oracle trace the SAME paths; the residual is fp32 rounding (fma contraction, libm ULPs,
4-wide vs 2-wide BVH visiting order).  Tolerances are stated per test.
"""
import os

import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle

WF = 1024  # LRHIP_FEAT_WAVEFRONT: scenes with Mix / Layered surfaces render in wavefront mode (lean megakernel | closure bits the heavy kernel serves)
from luisarender_amd.scenes import (cornell_box, generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene,
                                    generate_room_scene)

pytestmark = pytest.mark.gpu


def _variant(renderer):
    """the frame's kernel variant without the scheduler bit (LRHIP_FEAT_POOL): which scheduler a scene gets is lrhip.hip: wants_pool's
    business (tests/test_gpu_pool.py holds the two against each other), which closures / traversal / sampler code it needs is these tests'"""
    return renderer.last_variant() & ~4096


# Bars of the headline-scene comparisons: TWICE what was measured on MI355X in round 4 (profiles/r04_parity_bars.txt), for the device against
# the oracle and against the oracle on the device's own baked geometry (_same_geometry).  Round 3 asserted 2e-2 against measured 3e-3 ...
# 9e-3: a regression that doubled the error would have passed.
C2_SMALL_BAR, C2_SMALL_SAME_BAR = 1.0e-2, 1.0e-2   # measured 5.1e-3
C3_SMALL_BAR, C3_SMALL_SAME_BAR = 7e-3, 7e-3         # measured 3.5e-3
C2_FULL_BAR, C2_FULL_SAME_BAR = 1.7e-2, 1.7e-2       # measured 8.5e-3 (central 256 x 256, 8 spp)
FULL_BAR = {"c3": 2.6e-2, "c4": 4.5e-3}               # measured 1.29e-2 (2 spp), 2.2e-3 (1 spp)
FULL_SAME_BAR = {"c3": 2.9e-2, "c4": 4.5e-3}             # measured 1.43e-2, 1.9e-3

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def renderer():
    from luisarender_amd.render import MegaPathRenderer
    r = MegaPathRenderer(0)  # fails loudly if liblrhip.so or the GPU is missing: no fallback exists
    yield r
    r.close()


def _rel_l1(a, b):
    return float(np.abs(a[..., :3] - b[..., :3]).sum() / max(np.abs(b[..., :3]).sum(), 1e-20))


def _same_geometry(scene, spp, film, rect=None, cpu=None):
    """rel-L1 / relative mean difference of a device film against the oracle in BAKED-GEOMETRY mode (oracle_bvh.h: the oracle intersects the
    very fp32 world-space triangles the host bakes for the device, not the reference's object-space ones): what separates the KERNEL
    from the oracle once the design choice of baking is taken out -- the kernel's own arithmetic (fma contraction, hardware rcp,
    quantised BVH4 visiting order).  The reference-side truth is the default oracle, pinned in tests/test_oracle_vs_ref.py."""
    sub, _ = Oracle(scene, bake_instances=True).render(0, spp, rect=rect)
    if rect is not None:
        x0, y0, x1, y1 = rect
        film, sub = film[y0:y1, x0:x1], sub[y0:y1, x0:x1]
        cpu = cpu[y0:y1, x0:x1] if cpu is not None and cpu.shape != sub.shape else cpu
    # the scene's own sensitivity: the oracle against ITSELF in the other geometry mode -- the same algorithm, the same samples, hit points
    # that differ in their last bit.  On the large stand-ins (near-specular chains: GGX alpha 1e-4 mirrors, smooth glass, fixtures of
    # centimetres) that last bit decides after a few bounces which triangle a path meets next; what the device is apart from either
    # oracle cannot be smaller than what the two oracles are apart from each other.
    floor = _rel_l1(sub, cpu) if cpu is not None else float("nan")
    return _rel_l1(film, sub), abs(float(film[..., :3].mean()) - float(sub[..., :3].mean())) / float(sub[..., :3].mean()), floor


def _render_both(renderer, scene, spp):
    renderer.upload(scene)
    renderer.render(0, spp, counters=True, sync=True)
    gpu = renderer.download(converted=False)
    gc = renderer.counters()
    cpu, cc = Oracle(scene).render(0, spp)
    return gpu, gc, cpu, cc


def test_cornell_same_paths_and_image(renderer):
    sc = Scene.from_string(cornell_box(resolution=128, spp=16))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    # same path topology as the oracle: ray / hit / NEE counts agree to 1e-5 (a borderline triangle-edge hit may
    # flip: the kernel is built with fma contraction and 1-ulp hardware division, the oracle with neither)
    assert gc["paths"] == cc["paths"]
    for k in ("closest_rays", "surface_hits", "nee_samples", "path_length_sum"):
        assert abs(gc[k] - cc[k]) <= max(1, 1e-5 * cc[k]), (k, gc[k], cc[k])
    assert np.array_equal(gpu[..., 3], cpu[..., 3])          # sample counts: exact
    assert _rel_l1(gpu, cpu) < 1e-4                            # image: fp32 rounding + at most a few flipped paths
    px = np.abs(gpu[..., :3] - cpu[..., :3]).max(axis=-1) / (np.abs(cpu[..., :3]).max(axis=-1) + 1e-3)
    assert np.quantile(px, 0.999) < 1e-3                       # per pixel, 99.9 % within 1e-3 relative


def test_converted_film_matches_oracle_convert(renderer):
    sc = Scene.from_string(cornell_box(resolution=64, spp=4).replace("resolution { 64, 64 }", "resolution { 64, 64 } exposure { 1, 0, -1 }"))
    renderer.upload(sc)
    renderer.render(0, 4, sync=True)
    raw, conv = renderer.download(False), renderer.download(True)
    assert np.allclose(conv, Oracle(sc).convert(raw), rtol=1e-6, atol=1e-7)
    assert (conv[..., 3] == 1).all()


def test_golden_fixtures(renderer):
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from golden.make_golden import cornell_materials_text
    for name, text, spp in (("cornell_32_8spp.npz", cornell_box(resolution=32, spp=8), 8),
                            ("cornell_materials_48_8spp.npz", cornell_materials_text(), 8)):
        ref = np.load(os.path.join(GOLDEN, name))
        sc = Scene.from_string(text)
        renderer.upload(sc)
        renderer.render(0, spp, counters=True, sync=True)
        gpu = renderer.download(converted=False)
        assert np.array_equal(gpu[..., 3], ref["film"][..., 3])
        # specular chains amplify rounding differences (a flipped lobe choice changes one path): 2e-3 rel L1
        assert _rel_l1(gpu, ref["film"]) < (1e-4 if "materials" not in name else 2e-3), name


@pytest.mark.parametrize("material", ["oren", "mirror", "glass", "plastic", "metal", "disney", "disney_trans", "disney_thin", "mix", "mix_glass", "mix_nested",
                                      "mix_deep"])
def test_each_closure_in_a_cornell_box(renderer, material):
    from helpers import MATERIALS
    extra = MATERIALS[material].replace("Surface m ", f"Surface {material} ") + "\n"
    sc = Scene.from_string(cornell_box(resolution=64, spp=16, short_box_surface=material, tall_box_surface=material, extra_surfaces=extra))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-4 * cc["closest_rays"]  # rare branch flips (RR / lobe pick)
    assert _rel_l1(gpu, cpu) < 3e-3, material
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 1e-3  # no bias


def test_textured_disney_and_mix(renderer):
    """Row a14 on the per-hit ("dynamic") path: checkerboard-driven Disney parameters, a checkerboard Mix ratio and a
    generic sampler (the <COUNT, GENERIC, FULL> instantiation)."""
    extra = """
Texture chk : Checkerboard { on : Constant { v { 0.8, 0.3, 0.2 } } off : Constant { v { 0.2, 0.5, 0.8 } } scale { 4 } }
Texture chk1 : Checkerboard { on : Constant { v { 0.9 } } off : Constant { v { 0.1 } } scale { 3 } }
Surface dis : Disney { color { @chk } metallic { @chk1 } roughness : Constant { v { 0.3 } } clearcoat : Constant { v { 0.5 } } eta : Constant { v { 1.5 } } }
Surface ma : Matte { Kd { @chk } }
Surface mb : Glass { Kr : Constant { v { 0.9 } } Kt : Constant { v { 0.9 } } roughness : Constant { v { 0.1 } } eta : Constant { v { 1.5 } } }
Surface mx : Mix { a { @ma } b { @mb } ratio { @chk1 } }
"""
    sc = Scene.from_string(cornell_box(resolution=64, spp=16, short_box_surface="dis", tall_box_surface="mx", extra_surfaces=extra, sampler="PCG32"))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-4 * cc["closest_rays"]
    assert _rel_l1(gpu, cpu) < 3e-3
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 1e-3


@pytest.mark.parametrize("material", ["layered", "layered_medium", "mix_layered", "layered_mix", "layered_layered"])
def test_layered_closure(renderer, material):
    """Row a14, Layered (layered.cpp:195-470): its random walk is seeded from the BITS of the hit position and direction
    (:271,416), which fp contraction changes between the device and the oracle, so parity is statistical (8x8 block means),
    as for the alpha test."""
    from helpers import MATERIALS
    extra = MATERIALS[material].replace("Surface m ", f"Surface {material} ") + "\n"
    spp = 1024  # the estimator is noisy (fireflies through the rough dielectric coat): many samples, block means
    sc = Scene.from_string(cornell_box(resolution=64, spp=spp, short_box_surface=material, tall_box_surface=material, extra_surfaces=extra))
    gpu, gc, cpu, cc = _render_both(renderer, sc, spp)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    blocks = lambda f: f[..., :3].reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
    g, c = blocks(gpu), blocks(cpu)
    err, bias = np.abs(g - c).sum() / np.abs(c).sum(), abs(g.mean() - c.mean()) / c.mean()
    print(f"layered parity: block rel-L1 {err:.4f}, mean {bias:.5f}")
    assert err < 3e-2 and bias < 8e-3
    assert abs(gc["closest_rays"] - cc["closest_rays"]) < 5e-3 * cc["closest_rays"]
    # and it is not the bare substrate: rendering the bottom closure alone is far outside that tolerance
    bare = {"layered": "lay_b", "layered_medium": "lm_b", "mix_layered": "ml_p", "layered_mix": "lx_m", "layered_layered": "ll_b"}[material]
    sc2 = Scene.from_string(cornell_box(resolution=64, spp=spp, short_box_surface=bare, tall_box_surface=bare, extra_surfaces=extra))
    renderer.upload(sc2)
    renderer.render(0, spp, sync=True)
    assert np.abs(blocks(renderer.download(converted=False)) - c).sum() / np.abs(c).sum() > 5e-2


def test_alpha_tested_traversal(renderer):
    """Rows a6/a12: stochastic alpha test inside traversal.  The skip decision hashes the candidate's barycentric BITS
    (geometry.cpp:169), which depend on the intersector (world-space baked triangles here, object-space in the oracle;
    the reference's own come from the backend's hardware/ray-query unit), so parity is statistical: 8x8 block means."""
    extra = """
Texture holes : Checkerboard { on : Constant { v { 1 } } off : Constant { v { 0.2 } } scale { 3 } }
Surface cutout : Matte { Kd : Constant { v { 0.7, 0.6, 0.2 } } alpha { @holes } }
Surface veil_a : Mirror { color : Constant { v { 0.9 } } roughness : Constant { v { 0.3 } } alpha : Constant { v { 0.5 } } }
Surface veil_b : Matte { Kd : Constant { v { 0.2, 0.3, 0.8 } } }
Surface veil : Mix { a { @veil_a } b { @veil_b } ratio : Constant { v { 0.5 } } }
"""
    spp = 256
    sc = Scene.from_string(cornell_box(resolution=64, spp=spp, short_box_surface="cutout", tall_box_surface="veil", extra_surfaces=extra))
    assert sc.view().any_non_opaque == 1
    gpu, gc, cpu, cc = _render_both(renderer, sc, spp)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    blocks = lambda f: f[..., :3].reshape(8, 8, 8, 8, 3).mean(axis=(1, 3))
    g, c = blocks(gpu), blocks(cpu)
    assert np.abs(g - c).sum() / np.abs(c).sum() < 1.5e-2
    assert abs(g.mean() - c.mean()) / c.mean() < 3e-3
    # the alpha test changes the image: an opaque render of the same scene is far outside that tolerance
    opaque = Scene.from_string(cornell_box(resolution=64, spp=spp, short_box_surface="cutout", tall_box_surface="veil",
                                           extra_surfaces=extra.replace("alpha { @holes }", "").replace("alpha : Constant { v { 0.5 } }", "")))
    renderer.upload(opaque)
    renderer.render(0, spp, sync=True)
    assert np.abs(blocks(renderer.download(converted=False)) - c).sum() / np.abs(c).sum() > 5e-2
    assert abs(gc["closest_rays"] - cc["closest_rays"]) < 5e-3 * cc["closest_rays"]


ENV_SCENE = """
Surface ground : Matte {{ Kd : Constant {{ v {{ 0.5, 0.5, 0.5 }} }} }}
Surface shiny : Plastic {{ Kd : Constant {{ v {{ 0.7, 0.2, 0.1 }} }} roughness : Constant {{ v {{ 0.15 }} }} }}
Shape quad : InlineMesh {{ positions {{ -20,0,-20, 20,0,-20, 20,0,20, -20,0,20 }} indices {{ 0,2,1, 0,3,2 }} surface {{ @ground }} }}
Shape cube : InlineMesh {{
  positions {{ -1,0,-1, 1,0,-1, 1,2,-1, -1,2,-1, -1,0,1, 1,0,1, 1,2,1, -1,2,1 }}
  indices {{ 0,2,1, 0,3,2,  4,5,6, 4,6,7,  0,1,5, 0,5,4,  3,6,2, 3,7,6,  0,7,3, 0,4,7,  1,2,6, 1,6,5 }}
  surface {{ @shiny }} transform : SRT {{ rotate {{ 0, 1, 0, 30 }} }} }}
Camera cam : Pinhole {{ fov {{ 40 }} spp {{ {spp} }} film : Color {{ resolution {{ 64, 48 }} clamp {{ 64 }} }}
  position {{ 0, 4, 9 }} look_at {{ 0, 1, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @quad, @cube }}
  environment : {env}
  integrator : MegaPath {{ depth {{ 6 }} rr_depth {{ 2 }} }} }}
"""


@pytest.mark.parametrize("kind", ["image", "image_rotated", "directional", "directional_hidden", "combined", "combined_constant", "combined_nested",
                                  "combined_nested_constant"])
def test_image_and_directional_environments(renderer, tmp_path, kind):
    """Rows a11 / f1: importance-sampled lat-long environment (alias + pdf tables built by lrhost, shared by both sides) and
    the Directional cone, on the FULL kernel variant.  acos/atan2/sin differ by ulps between libm and the device, which can
    move a lookup across a texel edge, hence 1e-3 instead of the 1e-4 of the all-arithmetic scenes."""
    from test_environment import sky_image
    from luisarender_amd.scene import save_image
    path = str(tmp_path / "sky.exr")
    save_image(path, sky_image())
    env = {"image": f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} }}',
           "image_rotated": f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} scale {{ 2 }} compensate_mis {{ false }} transform : SRT {{ rotate {{ 0.2, 1, 0.1, 130 }} }} }}',
           "directional": "Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 6 } direction { 0.4, 1, 0.3 } }",
           "directional_hidden": "Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 25 } direction { -0.5, 1, 0.2 } visible { false } normalize { false } scale { 4 } }",
           "combined": f'Combined {{ a : Spherical {{ emission : Image {{ file {{ "{path}" }} }} transform : SRT {{ rotate {{ 1, 0, 0, 20 }} }} }} '
                       'b : Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 8 } direction { 0.4, 1, 0.3 } } scale_a { 0.7 } scale_b { 1.5 } '
                       'transform : SRT { rotate { 0, 1, 0, 60 } } }',
           "combined_constant": 'Combined { a : Spherical { emission : Constant { v { 0.4, 0.5, 0.7 } } } '
                                'b : Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 12 } direction { -0.3, 1, 0.2 } } }'}
    if kind.startswith("combined_nested"):
        # Combined nodes three deep (dev_shade.h: env_evaluate_tree / env_sample_tree, on a call-making variant); the oracle's walk of
        # the same tree is bit-exact with the reference's own code (test_oracle_vs_ref.py::test_li_environments_bit_exact[combined_nested])
        from test_oracle_vs_ref import nested_combined_environment
        text = nested_combined_environment(f'Image {{ file {{ "{path}" }} }}')
        if kind == "combined_nested_constant":  # a constant sky as the innermost leaf
            text = text.replace(f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} scale {{ 0.4 }}', "Spherical { emission : Constant { v { 0.2, 0.3, 0.5 } } scale { 0.4 }")
            assert "Constant { v { 0.2, 0.3, 0.5 } }" in text
        env[kind] = text
    env = env[kind]
    sc = Scene.from_string(ENV_SCENE.format(env=env, spp=16))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    if kind.startswith("combined_nested"):
        assert _variant(renderer) & 96 and not _variant(renderer) & 1024, _variant(renderer)  # out-of-line environment code, no wavefront mode
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-4 * cc["closest_rays"]
    assert cpu[..., :3].mean() > 0.01
    assert _rel_l1(gpu, cpu) < 1e-3, kind
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 3e-4


def test_environment_and_thin_lens(renderer):
    text = cornell_box(resolution=64, spp=8).replace("Camera cam : Pinhole {", "Camera cam : ThinLens {\n  aperture { 1.4 } focal_length { 50 } focus_distance { 900 }")
    text = text.replace("render {", "render {\n  environment : Spherical { emission : Constant { v { 0.3, 0.4, 0.6 } } }")
    sc = Scene.from_string(text)
    assert 0.0 < sc.view().integrator.env_prob < 1.0
    gpu, gc, cpu, cc = _render_both(renderer, sc, 8)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert _rel_l1(gpu, cpu) < 1e-4


def test_pcg32_sampler_stream(renderer):
    sc = Scene.from_string(cornell_box(resolution=64, spp=8, sampler="PCG32"))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 8)
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2 and _rel_l1(gpu, cpu) < 1e-4
    ind = Scene.from_string(cornell_box(resolution=64, spp=8))
    renderer.upload(ind)
    renderer.render(0, 8, sync=True)
    assert not np.array_equal(renderer.download(False), gpu)  # a different stream than xxhash+LCG


@pytest.mark.parametrize("sampler", ["Sobol", "PaddedSobol"])
def test_sobol_samplers(renderer, sampler):
    """Row a3': global Owen-scrambled Sobol and PaddedSobol streams are integer pipelines -> the same paths."""
    sc = Scene.from_string(cornell_box(resolution=(96, 64), spp=16, sampler=sampler))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    for k in ("closest_rays", "surface_hits"):
        assert abs(gc[k] - cc[k]) <= max(1, 1e-5 * cc[k]), (k, gc[k], cc[k])
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and _rel_l1(gpu, cpu) < 1e-4


@pytest.mark.parametrize("base", ["Independent", "Sobol"])
def test_tile_shared_sampler_wrapper(renderer, base):
    """src/samplers/tile_shared.cpp (round 3): all pixels of a tile share their base sampler's sequence; with jitter the tile grid moves
    per sample.  Integer pipeline in front of the base sampler: the same paths as the oracle (which is bit-equal to the reference's
    wrapper, tests/test_oracle_vs_ref.py::test_li_tile_shared_sampler_bit_exact)."""
    wrapped = f"TileShared {{ base : {base} {{ seed {{ 77 }} }} tile_size {{ 8, 4 }} jitter {{ true }} }}"
    sc = Scene.from_string(cornell_box(resolution=(96, 64), spp=16).replace("sampler : Independent { seed { 19980810 } }", "sampler : " + wrapped))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    for k in ("closest_rays", "surface_hits"):
        assert abs(gc[k] - cc[k]) <= max(1, 1e-5 * cc[k]), (k, gc[k], cc[k])
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and _rel_l1(gpu, cpu) < 1e-4
    plain = Scene.from_string(cornell_box(resolution=(96, 64), spp=16))
    other, _, _, _ = _render_both(renderer, plain, 16)
    assert not np.array_equal(other, gpu)  # the wrapper does something


def test_bathroom_class_instanced_scene(renderer, tmp_path):
    """~600 k instanced triangles, all five closures (BASELINE C2 geometry at reduced resolution/spp)."""
    path = generate_room_scene(str(tmp_path), resolution=(192, 192), spp=4)
    sc = Scene.load(path)
    gpu, gc, cpu, cc = _render_both(renderer, sc, 4)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 1e-3 * cc["closest_rays"]
    # Tolerance: GPU and oracle trace the same paths except where fp32 rounding flips a discrete decision
    # (lobe pick, RR, alias slot) on the smooth-shaded GGX fixtures; a flipped path changes its pixel by O(1)
    # at 4 spp.  Measured: ~1e-4 of the rays differ, rel-L1 4.7e-3, mean 3e-5.  Bars (round 4: twice the measured values, not 2e-2).
    err, same = _rel_l1(gpu, cpu), _same_geometry(sc, 4, gpu, cpu=cpu)
    print(f"c2 at 192 x 192: rel-L1 {err:.3e}; same baked geometry: rel-L1 {same[0]:.3e}, mean {same[1]:.2e}; oracle vs oracle (object-space vs baked geometry): {same[2]:.3e}")
    assert err < C2_SMALL_BAR and same[0] < C2_SMALL_SAME_BAR
    assert err < 2.0 * same[2] and same[0] < 2.0 * same[2]  # no further from either oracle than they are from each other (measured: 1.05 x, 0.94 x)
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 2e-3


def _blocks(f):
    h, w = f.shape[0] // 8 * 8, f.shape[1] // 8 * 8
    return f[:h, :w, :3].reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))


def test_bedroom_class_scene(renderer, tmp_path):
    """BASELINE C3 stand-in at reduced size: window openings, 10 % Glass (one dispersive), image-based Spherical
    environment with a sun lobe — the <environment> kernel variant.  Measured: rel-L1 3.3e-3, mean 2e-4, 51 of 1.7 M
    closest rays differ (texel-edge flips of the lat-long lookup, rough-glass lobe picks)."""
    sc = Scene.load(generate_bedroom_scene(str(tmp_path), resolution=(256, 144), spp=16))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert _variant(renderer) == 4 | 1  # LRHIP_FEAT_ENVIRONMENT | COUNT: nothing else is compiled in
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 1e-3 * cc["closest_rays"]
    err, same = _rel_l1(gpu, cpu), _same_geometry(sc, 16, gpu, cpu=cpu)
    print(f"c3 at 256 x 144: rel-L1 {err:.3e}; same baked geometry: rel-L1 {same[0]:.3e}, mean {same[1]:.2e}; oracle vs oracle (object-space vs baked geometry): {same[2]:.3e}")
    assert err < C3_SMALL_BAR and same[0] < C3_SMALL_SAME_BAR
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 2e-3


def test_camera_class_scene(renderer, tmp_path):
    """BASELINE C4 stand-in at reduced size: ~1 M triangles, Disney / Plastic / Matte on 8 image textures (sRGB PNG
    albedo, linear PNG roughness), thin lens, image environment + lamps — the <environment + Disney> variant.
    Measured: rel-L1 4.2e-4, mean 6e-5, 1 closest ray of 419 k differs."""
    sc = Scene.load(generate_camera_scene(str(tmp_path), resolution=(256, 144), spp=8, texture_size=1024))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 8)
    assert _variant(renderer) == 4 | 16 | 1
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 1e-3 * cc["closest_rays"]
    assert _rel_l1(gpu, cpu) < 5e-3
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 1e-3


def test_camera_class_scene_with_the_maps_the_bench_holds(tmp_path):
    """VERDICT r05 "weak" 2: the camera class with 2048 x 2048 maps, the size BENCH times -- 512 MB of float texels, above the 192 MB of the
    automatic rule (lrhip_set_texture_storage mode 1), so the eight PNG maps stay 8-bit texels on the device and the frame renders on a
    kernel of the LRHIP_FEAT_BYTE_TEXELS bit (round 6: <12308> / <8212>, the only lean kernels that hold the decode).  A window of the frame
    against the oracle -- which reads the host's floats -- at the C4 bars; the packed frame equals the float frame bit for bit."""
    from luisarender_amd.render import MegaPathRenderer
    sc = Scene.load(generate_camera_scene(str(tmp_path), resolution=(256, 144), spp=8, texture_size=2048))
    films, variants = {}, {}
    for mode in (1, 0):
        r = MegaPathRenderer(0)
        r.set_texture_storage(mode)
        r.upload(sc)
        packed = r.packed_texels()
        r.render(0, 8, counters=True, sync=True)
        films[mode], variants[mode] = r.download(converted=False), r.last_variant()
        if mode == 1:
            gc = r.counters()
            assert packed == 8 * 2048 * 2048, packed                      # the automatic rule fired: every map is held as codes
        else:
            assert packed == 0
        r.close()
    assert variants[1] & 8192 and (variants[1] & ~(4096 | 8192)) == 4 | 16 | 1, variants      # <environment + Disney> with the decode
    assert not variants[0] & 8192 and (variants[0] & ~4096) == 4 | 16 | 1, variants
    assert np.array_equal(films[0], films[1])                             # the decode gives back the host's floats: the same film
    cpu, cc = Oracle(sc).render(0, 8)
    gpu = films[1]
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 1e-3 * cc["closest_rays"]
    print(f"c4 with 2048^2 maps, packed: rel-L1 {_rel_l1(gpu, cpu):.3e}")
    assert _rel_l1(gpu, cpu) < 5e-3
    assert abs(gpu[..., :3].mean() - cpu[..., :3].mean()) / cpu[..., :3].mean() < 1e-3


def test_kitchen_class_scene(renderer, tmp_path):
    """BASELINE C5 stand-in at reduced size: every closure of SURVEY row a14 + NormalMap / alpha wrappers — the
    all-features variant.  Layered and alpha-tested surfaces are statistical by construction (their internal streams are
    seeded from hit-point bits), so the image is compared on 8x8 block means.  Measured: per-pixel rel-L1 4.8e-2,
    block rel-L1 2.1e-2, mean 2e-5, closest rays 1 810 464 vs 1 810 586.  Bars (round 5): twice the measured distance; the mean at twice what
    16 spp of this estimator leave of it (test_kitchen_class_scene_converges_on_the_oracle: +-1.4e-3 at 16 and 64 spp, 3e-4 at 256)."""
    sc = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(256, 144), spp=16))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert _variant(renderer) == WF | 8 | 16 | 32 | 64 | 1  # wavefront mode: lean alpha kernel + Disney / Mix / Layered in the heavy kernel + COUNT
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-3 * cc["closest_rays"]
    g, c = _blocks(gpu), _blocks(cpu)
    err, bias = np.abs(g - c).sum() / np.abs(c).sum(), abs(g.mean() - c.mean()) / c.mean()
    print(f"c5 at 256 x 144, 16 spp: block rel-L1 {err:.3e}, mean {bias:.2e}")
    assert err < 4.8e-2 and bias < 6.4e-3  # (twice the 2.41e-2 / 3.2e-3 measured in round 5)


def test_kitchen_class_scene_converges_on_the_oracle(renderer, tmp_path):
    """"statistical by construction" as a MEASUREMENT.  C5 is the one configuration where the device sits above the
    scene's own noise floor at a few spp: its Layered walks and alpha tests are seeded from hit-point BITS (layered.cpp:270,
    geometry.cpp:165-192), so a last-bit difference of a hit point -- fp contraction, baked vs object-space triangles -- gives such a
    vertex another random stream: another, equally valid, estimate of the same integral.  If that is all there is, the block-mean
    distance between device and oracle is Monte-Carlo noise: it falls like 1 / sqrt(spp), its mean vanishes, and the device is no
    further from the oracle on its own baked triangles than the oracle is from ITSELF across the two geometry modes.  A bias would
    show as a distance that stops falling.  Same frame, samples [0, 16), [0, 64), [0, 256) accumulated on all three sides."""
    sc = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(192, 112), spp=256))
    plain, baked = Oracle(sc), Oracle(sc, bake_instances=True)
    cpu = cpu_baked = None
    renderer.upload(sc)
    rows, begin = {}, 0
    for spp in (16, 64, 256):
        renderer.render(begin, spp, counters=False, sync=True)  # (progressive: the film carries on)
        gpu = renderer.download(converted=False)
        cpu, _ = plain.render(begin, spp, film=cpu)
        cpu_baked, _ = baked.render(begin, spp, film=cpu_baked)
        begin = spp
        assert _variant(renderer) == WF | 8 | 16 | 32 | 64 and np.array_equal(gpu[..., 3], cpu[..., 3]) and (gpu[..., 3] == spp).mean() > 0.999
        g, c, b = _blocks(gpu), _blocks(cpu), _blocks(cpu_baked)
        d = lambda x, y: float(np.abs(x - y).sum() / np.abs(y).sum())  # noqa: E731
        rows[spp] = {"device_vs_oracle": d(g, c), "device_vs_baked_oracle": d(g, b), "oracle_vs_oracle": d(b, c),
                     "bias": float((g.mean() - c.mean()) / c.mean()), "bias_baked": float((g.mean() - b.mean()) / b.mean())}
        print(f"c5 192 x 112 at {spp} spp: block rel-L1 device vs oracle {rows[spp]['device_vs_oracle']:.3e}, vs oracle on baked triangles {rows[spp]['device_vs_baked_oracle']:.3e}, "
              f"oracle vs oracle {rows[spp]['oracle_vs_oracle']:.3e}; mean bias {rows[spp]['bias']:+.2e} (baked {rows[spp]['bias_baked']:+.2e})")
    # (a) the distance falls with the samples -- as fast as the scene's own floor does.  Pure 1 / sqrt(spp) would be a factor 2 per 4 x spp; the
    # ORACLE AGAINST ITSELF across its two geometry modes falls by 1.53 and 1.57 on this frame (measured, round 5: block means of a clamped,
    # firefly-prone estimator), the device against either oracle by 1.63 ... 1.81: asserted at 1.5, and at no slower than 0.95 x the floor's own rate
    for lo, hi in ((16, 64), (64, 256)):
        floor_rate = rows[lo]["oracle_vs_oracle"] / rows[hi]["oracle_vs_oracle"]
        for k in ("device_vs_oracle", "device_vs_baked_oracle"):
            assert rows[hi][k] < rows[lo][k] / 1.5 and rows[lo][k] / rows[hi][k] > 0.95 * floor_rate, (k, lo, hi, floor_rate, rows)
    assert abs(rows[256]["bias"]) < 1e-3 and abs(rows[256]["bias_baked"]) < 1e-3, rows  # (b) the mean vanishes
    for spp in rows:  # (c) no further from the oracle than the oracle's two geometry modes are from each other
        assert rows[spp]["device_vs_baked_oracle"] < 1.5 * rows[spp]["oracle_vs_oracle"], (spp, rows)


def test_kernel_variant_selection(renderer):
    """lrhip_render launches the smallest precompiled superset of the scene's features (include/lrhip.h LRHIP_FEAT_*)."""
    from helpers import MATERIALS

    def variant(material, **kw):
        extra = MATERIALS[material].replace("Surface m ", "Surface probe ") + "\n" if material else ""
        sc = Scene.from_string(cornell_box(resolution=16, spp=1, short_box_surface="probe" if material else "white", extra_surfaces=extra, **kw))
        renderer.upload(sc)
        renderer.render(0, 1, sync=True)
        return _variant(renderer)

    assert variant(None) == 0
    assert variant(None, sampler="Sobol") == 2
    assert variant("glass") == 0 and variant("metal") == 0
    assert variant("disney") == 16 and variant("disney_thin") == 16
    cut = "Surface probe : Matte { Kd : Constant { v { 0.7 } } alpha : Constant { v { 0.5 } } }\n"
    sc = Scene.from_string(cornell_box(resolution=16, spp=1, short_box_surface="probe", extra_surfaces=cut))
    renderer.upload(sc)
    renderer.render(0, 1, sync=True)
    assert _variant(renderer) == 8  # alpha-tested traversal on the lean kernel
    # Mix / Layered surfaces: wavefront mode -- the lean kernel parks them for the heavy-closure kernel (lrhip.h: lrhip_set_wavefront);
    # its alpha-tested form only where a surface may be non-opaque
    assert variant("mix") == WF | 32
    assert variant("layered") == WF | 16 | 64
    try:  # ... unless it is switched off: the all-in-one megakernel variants, next precompiled superset
        renderer.set_wavefront(False)
        assert variant("mix") == 60 and variant("layered") == 124
    finally:
        renderer.set_wavefront(True)


def test_progressive_calls_and_determinism(renderer):
    """spp ranges compose (reference: one launch per sample index, integrator.cpp:92-94) and reruns are
    bit-identical: the film is accumulated in registers per pixel in sample order, no atomics."""
    sc = Scene.from_string(cornell_box(resolution=96, spp=12))
    renderer.upload(sc)
    renderer.render(0, 12, sync=True)
    once = renderer.download(False)
    renderer.clear()
    renderer.render(0, 12, sync=True)
    assert np.array_equal(renderer.download(False), once)
    renderer.clear()
    renderer.render(0, 5)
    renderer.render(5, 12, sync=True)
    split = renderer.download(False)
    assert np.array_equal(split[..., 3], once[..., 3]) and _rel_l1(split, once) < 1e-6


def test_tile_shards_reproduce_full_frame(renderer):
    """Screen-tile sharding: the union of the round-robin shards of 1, 2, 3 or 8 'ranks'
    is bit-identical to the unsharded frame, and each shard leaves non-owned pixels exactly zero."""
    from luisarender_amd.parallel import owner_mask
    sc = Scene.from_string(cornell_box(resolution=(100, 76), spp=6))
    renderer.upload(sc)
    renderer.render(0, 6, sync=True)
    full = renderer.download(False)
    for world in (2, 3, 8):
        total = np.zeros_like(full)
        for rank in range(world):
            renderer.clear()
            renderer.render(0, 6, rank=rank, world=world, sync=True)
            part = renderer.download(False)
            mask = owner_mask(100, 76, rank, world)
            assert (part[~mask] == 0).all()
            total += part
        assert np.array_equal(total, full), world
    # the item-size hint of the multi-GPU bench (lrhip_render_params.balance_shards): same hint -> same bits under any
    # sharding; a different hint only regroups the fp32 sums (more, smaller sample chunks)
    sc = Scene.from_string(cornell_box(resolution=(100, 76), spp=256))
    renderer.upload(sc)
    renderer.render(0, 256, sync=True, balance_shards=4)
    full4 = renderer.download(False)
    total = np.zeros_like(full4)
    for rank in range(4):
        renderer.clear()
        renderer.render(0, 256, rank=rank, world=4, sync=True, balance_shards=4)
        total += renderer.download(False)
    assert np.array_equal(total, full4)
    renderer.clear()
    renderer.render(0, 256, sync=True)
    full1 = renderer.download(False)
    assert np.array_equal(full1[..., 3], full4[..., 3]) and np.allclose(full1, full4, rtol=1e-4, atol=1e-4)


def test_full_size_c2_properties(renderer, tmp_path):
    """BASELINE C2 at full resolution (1024 x 1024, depth 16) with a few spp: size-independent
    properties — every pixel received exactly spp samples, no NaN/Inf, per-sample clamp honoured,
    energy within the Monte-Carlo error of the oracle's estimate on a pixel subset."""
    path = generate_room_scene(str(tmp_path), resolution=(1024, 1024), spp=8)
    sc = Scene.load(path)
    renderer.upload(sc)
    renderer.render(0, 8, sync=True)
    film = renderer.download(False)
    assert (film[..., 3] == 8).all() and np.isfinite(film).all()
    assert film[..., :3].max() <= 8 * 256.0 + 1e-3 and film[..., :3].min() >= 0
    o = Oracle(sc)
    sub, _ = o.render(0, 8, rect=(384, 384, 640, 640))
    a, b = film[384:640, 384:640, :3], sub[384:640, 384:640, :3]
    # Same seeded paths; specular chains through alpha = 1e-4 GGX lobes amplify fp32 rounding differences (fma
    # contraction, hardware division) into per-pixel differences without bias.  Measured: mean 1e-4, rel-L1 8e-3.
    fast = _rel_l1(a, b)
    same = _same_geometry(sc, 8, film, rect=(384, 384, 640, 640), cpu=sub)
    print(f"c2 full size, central 256 x 256 at 8 spp: rel-L1 {fast:.3e}, mean {abs(a.mean() - b.mean()) / b.mean():.2e}; same baked geometry: rel-L1 {same[0]:.3e}, mean {same[1]:.2e}; "
          f"oracle vs oracle (object-space vs baked geometry): {same[2]:.3e}")
    assert abs(a.mean() - b.mean()) / b.mean() < 2e-3 and fast < C2_FULL_BAR
    assert same[0] < C2_FULL_SAME_BAR and same[1] < 1e-3
    assert fast < 2.0 * same[2] and same[0] < 2.0 * same[2]  # (measured: 0.98 x, 0.97 x the oracle-vs-oracle distance)


def test_what_separates_c2_from_the_oracle(tmp_path):
    """Where do the 9e-3 of relative L1 between the device and the oracle on the C2 stand-in come from (3e-4 at a sixth of the triangles
    and a quarter of the resolution; 1e-6 on the Cornell scenes)?  Rounds 1-2 said "path flips from fast math" (-ffp-contract=fast,
    approximate division / sqrt / functions).  Round 3 built the lean kernel with IEEE arithmetic and an exact triangle test
    (`make ieee`: -ffp-contract=off, correctly rounded division / sqrt, 1 / det instead of v_rcp_f32) and measured (gpurun_out r03g):

        room, 100 k triangles, 256^2, 8 spp       shipped build    IEEE build
        transforms baked into the vertices           2.58e-4          6.80e-5      <- both sides intersect the SAME fp32 vertices
        fixtures as scaled + rotated instances        3.46e-4          3.77e-4
        bench size (600 k triangles, 1024^2 window)   8.85e-3          8.57e-3

    So fast math is a quarter of the story where the geometry is identical on both sides, and none of it where the fixtures are
    instances.  There the difference is the DESIGN: the device intersects world-space triangles baked in fp32 (one-level BVH, no
    per-ray instance transform, dev_trace.h), the oracle -- like the reference's ray-tracing unit -- takes the ray into object space.
    A vertex at world scale carries 2.4e-7 of absolute rounding; on a fixture scaled to 0.1 whose triangles have 6 mm edges that is
    4e-5 of an edge, and the interpolated normal moves with it; the room's near-specular chains (GGX alpha 1e-4 mirrors, smooth glass,
    fixtures 5-15 cm across) multiply a direction error by ~20 per bounce, so after two or three of them a path takes another turn.
    No bias (mean 1e-4).  The test pins the two facts that can be pinned: with identical geometry the IEEE build agrees markedly
    better than the shipped one, with instances it does not."""
    from luisarender_amd.render import MegaPathRenderer
    import luisarender_amd._ffi as ffi
    ieee_lib = os.path.join(ffi.LIB_DIR, "variants", "liblrhip_ieee.so")
    kw = dict(target_triangles=100_000, resolution=(256, 256), spp=8)
    errs = {}
    for name, opt in (("baked", dict(bake_transforms=True)), ("instanced", dict(inline_meshes=True))):
        sc = Scene.load(generate_room_scene(str(tmp_path), name=name, **opt, **kw))
        c, _ = Oracle(sc).render(0, 8)
        for build, lib in (("shipped", None), ("ieee", ieee_lib)):
            r = MegaPathRenderer(0, lib_path=lib)
            r.upload(sc)
            r.render(0, 8, counters=False, sync=True)
            assert _variant(r) == 0
            g = r.download(False)
            r.close()
            assert np.array_equal(g[..., 3], c[..., 3])
            errs[name, build] = _rel_l1(g, c)
            assert abs(g[..., :3].mean() - c[..., :3].mean()) / c[..., :3].mean() < 2e-3
            print(f"C2-class room, {name}, {build} build: device vs oracle rel-L1 {errs[name, build]:.2e}")
    assert errs["baked", "ieee"] < 1.5e-4 and errs["baked", "ieee"] < 0.5 * errs["baked", "shipped"], errs
    assert errs["instanced", "ieee"] > 0.5 * errs["instanced", "shipped"] and errs["instanced", "shipped"] < 2e-3, errs


@pytest.mark.parametrize("config", ["c3", "c4", "c5"])
def test_full_size_c3_c4_c5_properties(renderer, tmp_path, config):
    """BASELINE C3 / C4 / C5 stand-ins at their FULL resolution AND the bench's own triangle counts (1280x720 / 3840x2160 / 1280x720, depth
    16, 0.6 M / 1 M / 0.6 M triangles: round 4 -- round 3 ran 150 k), 1-2 spp, with the SHIPPED kernels <4> / <20> / wavefront (counters off): size-independent properties -- every pixel got its
    samples, no NaN / Inf, per-sample clamp honoured -- and the oracle's estimate on a window of the frame (same seeded paths;
    C5 holds Layered and alpha-tested surfaces, statistical by construction: 8x8 block means there)."""
    gen, res, spp, variant, rect = {"c3": (generate_bedroom_scene, (1280, 720), 2, 4, (512, 232, 768, 488)),
                                    "c4": (generate_camera_scene, (3840, 2160), 1, 20, (1792, 952, 2048, 1208)),
                                    "c5": (generate_kitchen_scene, (1280, 720), 2, WF | 8 | 16 | 32 | 64, (512, 232, 768, 488))}[config]
    kw = {"texture_size": 1024} if config == "c4" else {}
    sc = Scene.load(gen(str(tmp_path), resolution=res, spp=spp, **kw))  # (the generator's default triangle count = the bench's)
    renderer.upload(sc)
    renderer.render(0, spp, counters=False, sync=True)
    assert _variant(renderer) == variant
    film = renderer.download(False)
    assert film.shape == (res[1], res[0], 4) and np.isfinite(film).all()
    assert (film[..., 3] <= spp).all() and (film[..., 3] == spp).mean() > 0.999  # (a NaN sample is rejected by the film, color.cpp:111)
    assert film[..., :3].max() <= spp * 256.0 + 1e-3 and film[..., :3].min() >= 0
    x0, y0, x1, y1 = rect
    sub, _ = Oracle(sc).render(0, spp, rect=rect)
    a, b = film[y0:y1, x0:x1], sub[y0:y1, x0:x1]
    if config == "c5":
        g, c = _blocks(a), _blocks(b)
        err = np.abs(g - c).sum() / np.abs(c).sum()
        print(f"{config}: block rel-L1 {err:.3e}, mean {abs(g.mean() - c.mean()) / c.mean():.2e}")
        assert err < 5.4e-2 and abs(g.mean() - c.mean()) / c.mean() < 4.5e-3  # (twice the measured 2.67e-2 / 2.2e-3 of 2 spp, profiles/r04_parity_bars.txt; the distance is the scene's own noise: test_kitchen_class_scene_converges_on_the_oracle)
    else:
        same = _same_geometry(sc, spp, film, rect=rect, cpu=sub)
        print(f"{config}: rel-L1 {_rel_l1(a, b):.3e}, mean {abs(a[..., :3].mean() - b[..., :3].mean()) / b[..., :3].mean():.2e}; same baked geometry: rel-L1 {same[0]:.3e}, mean {same[1]:.2e}; "
              f"oracle vs oracle (object-space vs baked geometry): {same[2]:.3e}")
        assert _rel_l1(a, b) < FULL_BAR[config] and abs(a[..., :3].mean() - b[..., :3].mean()) / b[..., :3].mean() < 4e-3  # (1-2 spp of a 65 k-pixel window: the means are noisy)
        assert same[0] < FULL_SAME_BAR[config] and same[1] < 6e-3


@pytest.mark.parametrize("case", ["lean", "environment", "alpha", "env_alpha", "disney", "env_disney", "mix_alpha", "layered", "nested", "direct", "vpt", "sobol",
                                  "mix_sobol", "layered_pcg", "nested_sobol", "direct_pcg", "vpt_pcg"])
def test_shipped_kernels_equal_their_counting_twins(renderer, tmp_path, case):
    """Every parity test above drives the COUNT twin of a kernel variant (it needs the ray counters); bench.py and the CLI launch
    the twin without counters.  The two are the same template with `if (COUNT)` blocks, but they are different BINARIES (round 1
    saw a build of <124> that kept its ray counts and emitted NaNs), so each shipped variant is held to its twin: the same
    paths (equal sample counts per pixel), the same film up to what a differently scheduled fp32 expression (contraction, a
    reassociated sum) does to a specular chain now and then -- measured: bit-identical for <16> <252> <256> <2>, rel-L1 below
    1e-3 for the others; Layered seeds its walk from position bits, so block means there.  With tests/test_ref_golden.py
    (shipped variants vs the reference's frames) this closes VERDICT r01 item 2."""
    from helpers import MATERIALS
    from test_environment import sky_image
    from luisarender_amd.scene import save_image

    def mat(*names):
        return "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in names)
    sky = str(tmp_path / "sky.exr")
    save_image(sky, sky_image())
    env = f'render {{\n  environment : Spherical {{ emission : Image {{ file {{ "{sky}" }} }} }}'
    alpha = "Surface cutout : Matte { Kd : Constant { v { 0.7, 0.6, 0.2 } } alpha : Checkerboard { on : Constant { v { 1 } } off : Constant { v { 0.2 } } scale { 3 } } }\n"
    text, variant = {
        "lean": (cornell_box(resolution=64, spp=8, short_box_surface="glass", tall_box_surface="metal", extra_surfaces=mat("glass", "metal")), 0),
        "environment": (cornell_box(resolution=64, spp=8).replace("render {", env), 4),
        "alpha": (cornell_box(resolution=64, spp=8, short_box_surface="cutout", tall_box_surface="glass", extra_surfaces=mat("glass") + alpha), 8),
        "env_alpha": (cornell_box(resolution=64, spp=8, short_box_surface="cutout", extra_surfaces=alpha).replace("render {", env), 12),
        "disney": (cornell_box(resolution=64, spp=8, short_box_surface="disney", tall_box_surface="disney_thin", extra_surfaces=mat("disney", "disney_thin")), 16),
        "env_disney": (cornell_box(resolution=64, spp=8, short_box_surface="disney", extra_surfaces=mat("disney")).replace("render {", env), 20),
        "mix_alpha": (cornell_box(resolution=64, spp=8, short_box_surface="mix_nested", tall_box_surface="cutout", extra_surfaces=mat("mix_nested") + alpha), WF | 8 | 32),
        "layered": (cornell_box(resolution=64, spp=8, short_box_surface="layered", tall_box_surface="layered_medium", extra_surfaces=mat("layered", "layered_medium")), WF | 16 | 64),
        "nested": (cornell_box(resolution=64, spp=8, short_box_surface="mix_layered", tall_box_surface="layered_mix", extra_surfaces=mat("mix_layered", "layered_mix")), WF | 16 | 32 | 64 | 512),
        "direct": (cornell_box(resolution=64, spp=8, short_box_surface="glass", extra_surfaces=mat("glass")).replace("integrator : MegaPath {", 'integrator : Direct { importance_sampling { "both" }'), 252),
        "vpt": (cornell_box(resolution=64, spp=8, extra_surfaces=FOG, short_box_surface="skin").replace("integrator : MegaPath {", "integrator : MegaVPTNaive {")
                .replace("render {", "render {\n  environment_medium { @fog }").replace("surface { @skin }", "surface { @skin } medium { @inner }"), 256),
        "sobol": (cornell_box(resolution=(96, 64), spp=8, sampler="Sobol"), 2),
        # the generic-sampler twins (| 2) of the variants that make real calls: every shipped binary of that kind is held to its
        # counting twin (they are the ones a compiler mishap has hit, Makefile: CALL_SAFE_FLAGS)
        "mix_sobol": (cornell_box(resolution=64, spp=8, short_box_surface="mix_nested", tall_box_surface="cutout", extra_surfaces=mat("mix_nested") + alpha, sampler="Sobol"), WF | 8 | 32 | 2),
        "layered_pcg": (cornell_box(resolution=64, spp=8, short_box_surface="layered", tall_box_surface="layered_medium", extra_surfaces=mat("layered", "layered_medium"), sampler="PCG32"), WF | 16 | 64 | 2),
        "nested_sobol": (cornell_box(resolution=64, spp=8, short_box_surface="mix_layered", tall_box_surface="layered_mix", extra_surfaces=mat("mix_layered", "layered_mix"), sampler="PaddedSobol"), WF | 16 | 32 | 64 | 512 | 2),
        "direct_pcg": (cornell_box(resolution=64, spp=8, short_box_surface="glass", extra_surfaces=mat("glass"), sampler="PCG32").replace("integrator : MegaPath {", 'integrator : Direct { importance_sampling { "both" }'), 254),
        "vpt_pcg": (cornell_box(resolution=64, spp=8, extra_surfaces=FOG, short_box_surface="skin", sampler="PCG32").replace("integrator : MegaPath {", "integrator : MegaVPTNaive {")
                    .replace("render {", "render {\n  environment_medium { @fog }").replace("surface { @skin }", "surface { @skin } medium { @inner }"), 258),
    }[case]
    sc = Scene.from_string(text)
    films = []
    for count in (True, False):
        renderer.upload(sc)
        renderer.render(0, 8, counters=count, sync=True)
        assert _variant(renderer) == (variant | (1 if count else 0)), (case, _variant(renderer))
        films.append(renderer.download(converted=False))
    a, b = films
    assert a[..., :3].sum() > 0 and np.isfinite(b).all() and np.array_equal(a[..., 3], b[..., 3]), case
    renderer.upload(sc)  # and the shipped binary is deterministic (a miscompiled <124> once was not: lost samples, another sum every run)
    renderer.render(0, 8, counters=False, sync=True)
    assert np.array_equal(renderer.download(converted=False), b), case
    if case in ("layered", "nested", "layered_pcg", "nested_sobol"):
        g, c = _blocks(b), _blocks(a)
        err, bias = np.abs(g - c).sum() / np.abs(c).sum(), abs(g.mean() - c.mean()) / c.mean()
        print(f"{case}: block rel-L1 {err:.3e}, mean {bias:.2e}")
        assert err < 0.15 and bias < 3e-2  # 8 spp of a firefly-prone estimator: the tight layered check is test_layered_closure / test_ref_golden
    else:
        err, bias = _rel_l1(b, a), abs(b[..., :3].mean() - a[..., :3].mean()) / a[..., :3].mean()
        print(f"{case}: rel-L1 {err:.3e}, mean {bias:.2e}")
        assert err < 3e-3 and bias < 1e-3, (case, err, bias)


def test_error_paths(renderer):
    import ctypes as C
    from luisarender_amd import _ffi
    lib = _ffi.hip_lib()
    sc = Scene.from_string(cornell_box(resolution=16, spp=1), build_accel=False)
    ctx = C.c_void_p()
    assert lib.lrhip_create(0, C.byref(ctx)) == 0
    v = sc.view()
    assert lib.lrhip_upload_scene(ctx, C.byref(v)) < 0 and b"accel" in lib.lrhip_last_error()
    p = _ffi.RenderParams()
    assert lib.lrhip_render(ctx, C.byref(p)) < 0  # no scene uploaded
    assert lib.lrhip_create(99, C.byref(C.c_void_p())) < 0
    lib.lrhip_destroy(ctx)


@pytest.mark.parametrize("integrator", ['Direct { importance_sampling { "both" }', 'Direct { importance_sampling { "light" }',
                                        'Direct { importance_sampling { "surface" }', "Normal {", "Normal { remap { false } shading { false }"])
def test_sibling_integrators(renderer, integrator):
    """DirectLighting (direct.cpp:66-200) and NormalVisualizer (normal.cpp:36-70) on the device — the
    all-features variant + kFeatAux — against their oracle restatements: same sampler streams, same paths."""
    from helpers import MATERIALS
    extra = MATERIALS["glass"].replace("Surface m ", "Surface probe ") + "\n"
    text = cornell_box(resolution=64, spp=16, short_box_surface="probe", extra_surfaces=extra).replace("integrator : MegaPath {", "integrator : " + integrator)
    text = text.replace("render {", "render {\n  environment : Spherical { emission : Constant { v { 0.3, 0.4, 0.6 } } }")
    sc = Scene.from_string(text)
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert _variant(renderer) == 252 | 1
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-4 * cc["closest_rays"]
    assert np.abs(cpu[..., :3]).mean() > 0.01 and _rel_l1(gpu, cpu) < 1e-3, integrator


def test_normal_visualiser_needs_no_light(renderer):
    """normal.cpp has no "no lights -> abort" branch: a light-less scene renders (mega_path.cpp:40-46 would return black)."""
    text = """
Surface s : Matte { Kd : Constant { v { 0.5 } } }
Shape quad : InlineMesh { positions { -5,0,-5, 5,0,-5, 5,0,5, -5,0,5 } indices { 0,2,1, 0,3,2 } surface { @s } }
Shape cube : InlineMesh {
  positions { -1,0,-1, 1,0,-1, 1,2,-1, -1,2,-1, -1,0,1, 1,0,1, 1,2,1, -1,2,1 }
  indices { 0,2,1, 0,3,2,  4,5,6, 4,6,7,  0,1,5, 0,5,4,  3,6,2, 3,7,6,  0,7,3, 0,4,7,  1,2,6, 1,6,5 }
  surface { @s } transform : SRT { rotate { 0, 1, 0, 30 } } }
Camera cam : Pinhole { fov { 40 } spp { 2 } film : Color { resolution { 32, 32 } } position { 0, 4, 8 } look_at { 0, 1, 0 } }
render { cameras { @cam } shapes { @quad, @cube } integrator : Normal { } }
"""
    sc = Scene.from_string(text)
    assert not sc.has_lighting
    gpu, gc, cpu, cc = _render_both(renderer, sc, 2)
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and _rel_l1(gpu, cpu) < 1e-5
    img = gpu[..., :3] / 2.0
    assert img.min() >= -1e-6 and img.max() <= 1 + 1e-6
    assert len(np.unique(np.round(img.reshape(-1, 3), 2), axis=0)) >= 4  # background, floor and at least two cube faces


FOG = """
Medium fog : Homogeneous { sigma_a : Constant { v { 0.0001, 0.0002, 0.0003 } } sigma_s : Constant { v { 0.0006 } } eta { 1 }
  phasefunction : HenyeyGreenstein { g { 0.4 } } }
Medium inner : Homogeneous { sigma_a : Constant { v { 0.004, 0.002, 0.001 } } sigma_s : Constant { v { 0.003 } } eta { 1.3 } priority { 0 }
  phasefunction : HenyeyGreenstein { g { -0.3 } } }
Surface skin : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.05 } } eta : Constant { v { 1.3 } } }
"""


def _vpt_scene(case, resolution=64, spp=64):
    """the Cornell box under MegaVPTNaive: `vacuum` (no medium), `fog_env` (an environment medium lit by a constant environment),
    `fog_lamp_and_medium_box` (the lamp-lit fog with a glass-skinned medium box, the whole box tilted)"""
    extra = "" if case == "vacuum" else FOG
    boxed = case == "fog_lamp_and_medium_box"
    text = cornell_box(resolution=resolution, spp=spp, depth=8, extra_surfaces=extra, short_box_surface="skin" if boxed else "white")
    text = text.replace("integrator : MegaPath {", "integrator : MegaVPTNaive {")
    if case != "vacuum":
        text = text.replace("render {", "render {\n  environment_medium { @fog }")
    if case == "fog_env":
        assert "light : Diffuse { emission : Constant { v { 17, 12, 4 } } }" in text
        text = text.replace("light : Diffuse { emission : Constant { v { 17, 12, 4 } } }", "")
        text = text.replace("render {", "render {\n  environment : Spherical { emission : Constant { v { 2, 2.5, 3 } } }")
    if boxed:
        assert "surface { @skin }" in text
        text = text.replace("surface { @skin }", "surface { @skin } medium { @inner }")
        import re  # tilt the whole box: on axis-aligned surfaces the lottery decides ~20 % of the directly seen emission
        shapes = re.search(r"shapes \{ (.*?) \}\n  integrator", text, re.S).group(1)
        text = text.replace(f"shapes {{ {shapes} }}", "shapes { @tilted }")
        text = text.replace("Camera cam", f"Shape tilted : Group {{ shapes {{ {shapes} }} transform : SRT {{ rotate {{ 0.2, 1, 0.1, 5 }} translate {{ -23, 0, 25 }} }} }}\nCamera cam")
    return text


@pytest.mark.parametrize("case", ["vacuum", "fog_env", "fog_lamp_and_medium_box"])
def test_volumetric_megakernel(renderer, case):
    """MegaVPTNaive (mega_vpt_naive.cpp:170-483) on the device against its oracle restatement.  The sampler and
    the PCG32 majorant streams are shared, so the two trace the same paths: `vacuum` (no medium) and `fog_env` (an
    environment medium lit by a constant environment) agree like the path tracer does.
    `fog_lamp_and_medium_box` is statistical BY CONSTRUCTION OF THE REFERENCE: after a medium "hit surface" event the ray
    origin is moved onto the surface (homogeneous.cpp:64) and the emitter is evaluated from there (mega_vpt_naive.cpp:331);
    the direction from that origin to the hit point lies in the surface up to rounding, so diffuse.cpp:84
    (|cos| < 1e-6 -> no emission) fires or not depending on the last bits — fused multiply-adds alone move the oracle's own
    mean by 0.5 % (measured), the device's arithmetic by 2.5 %.  Round 6 measured what that is (test_lamp_lit_fog_converges_on_the_oracle
    below): the estimator's mean depends on the intersector's rounding -- the oracle on the device's own baked triangles is 4.3 % darker
    than on the object-space ones, the device lies between the two.  The bars here are twice what the device shows at 64 spp (block
    rel-L1 3.97e-2, mean -2.4e-2); the case also checks that the medium box is really entered (it differs from the same scene without
    media by much more)."""
    boxed = case == "fog_lamp_and_medium_box"
    text = _vpt_scene(case)
    sc = Scene.from_string(text)
    assert sc.view().integrator.kind == 3 and sc.view().medium_count == {"vacuum": 0, "fog_env": 1, "fog_lamp_and_medium_box": 2}[case]
    gpu, gc, cpu, cc = _render_both(renderer, sc, 64)
    assert _variant(renderer) == 256 | 1
    # NaN samples of the lottery are rejected by the film on both sides, not necessarily the same ones
    assert np.isfinite(gpu).all() and np.abs(gpu[..., 3] - cpu[..., 3]).max() <= 8 and abs(gpu[..., 3].sum() - cpu[..., 3].sum()) <= 2e-3 * cpu[..., 3].sum()
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 5e-3 * cc["closest_rays"]
    g, c = _blocks(gpu), _blocks(cpu)
    err, bias = np.abs(g - c).sum() / np.abs(c).sum(), abs(g.mean() - c.mean()) / c.mean()
    print(f"vpt {case}: block rel-L1 {err:.5f}, mean {bias:.6f}, per-pixel rel-L1 {_rel_l1(gpu, cpu):.5f}, image mean {c.mean():.4f}, "
          f"scatter+surface vertices per path {cc['nee_samples'] / cc['paths']:.2f}")
    assert c.mean() > 0.02
    if not boxed:
        assert _rel_l1(gpu, cpu) < 5e-3 and bias < 1e-3
    else:
        assert err < 0.08 and bias < 0.05  # (twice the 3.97e-2 / 2.43e-2 measured in round 6)
        plain = Scene.from_string(text.replace(" medium { @inner }", "").replace("\n  environment_medium { @fog }", ""))
        renderer.upload(plain)
        renderer.render(0, 64, sync=True)
        p = _blocks(renderer.download(converted=False))
        assert np.abs(p - c).sum() / np.abs(c).sum() > 0.25


def test_lamp_lit_fog_converges_on_the_oracle(renderer):
    """the lamp-lit fog case's "statistical by construction" as a MEASUREMENT, the way the kitchen class got one in round 5.
    The reference moves the ray origin ONTO the surface it just reached -- origin + direction * t_hit, homogeneous.cpp:64 -- and evaluates the
    emitter from there (mega_vpt_naive.cpp:331), so diffuse.cpp:84's |cos| < 1e-6 test is decided by how far off the surface's plane the
    rounding of t_hit and of the reconstructed hit point leave that origin.  Four sides render the same samples [0, 64), [0, 256), [0, 1024):
    the device (volumetric kernel: no contraction, exact division, 1 / det in the triangle test), the oracle, the oracle BUILT WITH
    -ffp-contract=fast (oracle/liboracle_fma.so: what fused multiply-adds alone do to it), and the oracle intersecting the WORLD-space
    triangles the host bakes for the device instead of the reference's object-space ones (bake_instances).
    MEASURED (round 6, profiles/r06n_fog_convergence.txt): this is not a lottery with fixed odds.  The oracle on the baked triangles is
    4.3 % DARKER than the oracle on the object-space ones, at 64 spp and at 1024 spp alike -- t_hit from a world-space triangle test lands
    the origin closer to the plane, the test fires more often -- while contraction alone moves the oracle by 0.8 % -> 0.2 % (noise).  The
    estimator's MEAN is a function of the intersector's rounding; the reference's own intersector is LuisaCompute's (absent; DESIGN.md
    section 8: unpinnable), the object-space one of the oracle is the builder's stand-in for it.  The device intersects baked triangles (like
    the darker oracle) and reconstructs the hit point in object space (like the brighter one): it sits BETWEEN the two, -2.2 % ... -1.5 %
    against the object-space oracle and +3 % against the baked one, and what is left after that offset keeps falling with the samples, within
    twice the distance of the oracle's two builds.  Asserted: the device is no further from the reference-mode oracle than the oracle's own
    second geometry mode is, its mean lies between the two modes', and its distance with the mean offset taken out falls by > 1.25 per
    4 x spp and stays below twice the oracle pair's."""
    sc = Scene.from_string(_vpt_scene("fog_lamp_and_medium_box", resolution=64, spp=1024))
    plain, fused, baked = Oracle(sc), Oracle(sc, lib="liboracle_fma.so"), Oracle(sc, bake_instances=True)
    cpu = cpu_fused = cpu_baked = None
    renderer.upload(sc)
    rows, begin = {}, 0

    def blocks(f):  # block means of the per-pixel ESTIMATES (the film rejects NaN samples on every side, not necessarily the same ones)
        return _blocks(f[..., :3] / np.maximum(f[..., 3:4], 1.0))

    d = lambda x, y: float(np.abs(x - y).sum() / np.abs(y).sum())  # noqa: E731
    for spp in (64, 256, 1024):
        renderer.render(begin, spp, counters=False, sync=True)  # (progressive: the film carries on)
        gpu = renderer.download(converted=False)
        cpu, _ = plain.render(begin, spp, film=cpu)
        cpu_fused, _ = fused.render(begin, spp, film=cpu_fused)
        cpu_baked, _ = baked.render(begin, spp, film=cpu_baked)
        begin = spp
        assert _variant(renderer) == 256 and np.isfinite(gpu).all() and np.abs(gpu[..., 3] - cpu[..., 3]).max() <= max(8, spp // 8)
        g, c, f, b = blocks(gpu), blocks(cpu), blocks(cpu_fused), blocks(cpu_baked)
        rows[spp] = {"device_vs_oracle": d(g, c), "fma_oracle_vs_oracle": d(f, c), "baked_oracle_vs_oracle": d(b, c), "device_vs_baked_oracle": d(g, b),
                     "device_vs_oracle_without_offset": d(g * (c.mean() / g.mean()), c),
                     "bias": float((g.mean() - c.mean()) / c.mean()), "bias_fma_oracle": float((f.mean() - c.mean()) / c.mean()),
                     "bias_baked_oracle": float((b.mean() - c.mean()) / c.mean())}
        r = rows[spp]
        print(f"vpt lamp-lit fog 64 x 64 at {spp} spp: block rel-L1 to the oracle: device {r['device_vs_oracle']:.3e} (mean offset taken out {r['device_vs_oracle_without_offset']:.3e}), "
              f"fma oracle {r['fma_oracle_vs_oracle']:.3e}, baked-triangle oracle {r['baked_oracle_vs_oracle']:.3e}; device vs baked-triangle oracle {r['device_vs_baked_oracle']:.3e}; "
              f"mean bias device {r['bias']:+.2e}, fma oracle {r['bias_fma_oracle']:+.2e}, baked-triangle oracle {r['bias_baked_oracle']:+.2e}")
    for spp, r in rows.items():
        # (a) no further from the reference-mode oracle than the oracle's own second geometry mode
        assert r["device_vs_oracle"] < r["baked_oracle_vs_oracle"], (spp, rows)
        # (b) the device's mean lies between the two geometry modes' (1 % of slack either side: 64 spp of this estimator)
        assert r["bias_baked_oracle"] - 1e-2 < r["bias"] < 1e-2, (spp, rows)
    # (c) what is left of the distance once the mean offset is out keeps falling with the samples (measured 3.45e-2 / 2.15e-2 / 1.55e-2: the offset
    # is not uniform over the frame -- only the blocks that see lamp-lit surfaces carry it -- so it falls slower than the oracle's two builds approach
    # each other, 3.35e-2 / 1.57e-2 / 1.02e-2) and stays within twice the oracle pair's distance at every sample count
    for lo, hi in ((64, 256), (256, 1024)):
        assert rows[lo]["device_vs_oracle_without_offset"] / rows[hi]["device_vs_oracle_without_offset"] > 1.25, (lo, hi, rows)
    for spp, r in rows.items():
        assert r["device_vs_oracle_without_offset"] < 2.0 * r["fma_oracle_vs_oracle"], (spp, rows)


def test_edge_cases(renderer):
    """tiny and ragged films, one sample, a one-triangle scene (the BVH root is a leaf), an empty sample range"""
    for res, spp in (((1, 1), 1), ((3, 5), 2), ((9, 8), 1), ((17, 1), 3)):
        sc = Scene.from_string(cornell_box(resolution=res, spp=spp))
        gpu, gc, cpu, cc = _render_both(renderer, sc, spp)
        assert gpu.shape == (res[1], res[0], 4) and np.array_equal(gpu[..., 3], cpu[..., 3]) and (gpu[..., 3] == spp).mean() > 0.999
        assert gc["closest_rays"] == cc["closest_rays"] and _rel_l1(gpu, cpu) < 1e-4, res
    one = Scene.from_string("""
Shape tri : InlineMesh { positions { -1,-1,0, 1,-1,0, 0,1,0 } indices { 0,1,2 } light : Diffuse { emission : Constant { v { 2, 3, 4 } } two_sided { true } } }
Camera cam : Pinhole { fov { 50 } spp { 4 } film : Color { resolution { 24, 24 } } position { 0, 0, 3 } look_at { 0, 0, 0 } }
render { cameras { @cam } shapes { @tri } integrator : MegaPath { depth { 3 } } }
""")
    gpu, gc, cpu, cc = _render_both(renderer, one, 4)
    assert gc["closest_rays"] == cc["closest_rays"] and _rel_l1(gpu, cpu) < 1e-5 and gpu[12, 12, 0] > 0 and gpu[0, 0, 0] == 0
    renderer.upload(one)
    renderer.render(2, 2, sync=True)  # empty sample range: nothing happens
    assert not renderer.download(converted=False).any()


def test_sphere_loopsubdiv_and_jpeg_texture(renderer, tmp_path):
    """the shapes of csrc/host/subdiv.cpp and a (swizzled) texture read by csrc/host/image_codecs.cpp, through the device path"""
    PIL = pytest.importorskip("PIL.Image")
    y, x = np.mgrid[0:64, 0:64]
    pic = np.stack([128 + 100 * np.sin(x / 5.0), 128 + 100 * np.cos(y / 7.0), 4 * x], axis=-1).clip(0, 255).astype(np.uint8)
    PIL.fromarray(pic, "RGB").save(tmp_path / "tex.jpg", quality=90, subsampling=2)
    text = """
Shape ball : Sphere { subdivision { 3 } surface : Plastic { Kd : Constant { v { 0.7, 0.3, 0.2 } } roughness : Constant { v { 0.3 } } eta : Constant { v { 1.5 } } }
  transform : SRT { translate { -1.2, 1, 0 } } }
Shape tetra : InlineMesh { positions { 1,1,1, -1,-1,1, -1,1,-1, 1,-1,-1 } indices { 0,1,2, 0,3,1, 0,2,3, 1,3,2 } }
Shape blob : LoopSubdiv { mesh { @tetra } level { 3 } surface : Metal { eta { "Cu" } roughness : Constant { v { 0.25 } } } transform : SRT { translate { 1.2, 1, 0 } } }
Shape floor : InlineMesh { positions { -4,0,-4, 4,0,-4, 4,0,4, -4,0,4 } indices { 0,2,1, 0,3,2 } uvs { 0,0, 1,0, 1,1, 0,1 }
  surface : Matte { Kd : Swizzle { base : Image { file { "tex.jpg" } } swizzle { "bgr" } } } }
Shape lamp : InlineMesh { positions { -1,4,-1, 1,4,-1, 1,4,1, -1,4,1 } indices { 0,1,2, 0,2,3 } light : Diffuse { emission : Constant { v { 12, 11, 10 } } } }
Camera cam : Pinhole { fov { 45 } spp { 16 } film : Color { resolution { 96, 64 } } position { 0, 2.5, 7 } look_at { 0, 0.8, 0 } }
render { cameras { @cam } shapes { @ball, @blob, @floor, @lamp } integrator : MegaPath { depth { 6 } } }
"""
    (tmp_path / "scene.luisa").write_text(text)
    sc = Scene.load(str(tmp_path / "scene.luisa"))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 16)
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and abs(gc["closest_rays"] - cc["closest_rays"]) <= 4
    assert _rel_l1(gpu, cpu) < 2e-3 and gpu[..., :3].mean() > 0.01


@pytest.mark.parametrize("scale", ["2.5, -3", "170001.25, -290000.5"])
@pytest.mark.parametrize("filter_mode", ["point", "bilinear"])
@pytest.mark.parametrize("address", ["repeat", "mirror", "edge", "zero"])
def test_texture_address_modes(renderer, tmp_path, address, filter_mode, scale):
    """dev_shade.h: texel_wrap (floor-mod through a float reciprocal, integer path for huge coordinates) against the oracle's
    integer wrap, uvs outside [0, 1) on both sides; the second scale puts the texel coordinates beyond 2^20 (integer path).  The
    oracle's wrap is bit-exact with the reference's sampler (test_oracle_vs_ref.py::test_li_texture_address_modes_bit_exact)."""
    from test_oracle_vs_ref import ADDRESS_SCENE, address_mode_texture
    address_mode_texture(tmp_path / "tex.pfm")
    text = ADDRESS_SCENE.format(address=address, filter=filter_mode, spp=8, res="96, 64").replace("uv_scale { 2.5, -3 }", "uv_scale { %s }" % scale)
    (tmp_path / "scene.luisa").write_text(text)
    sc = Scene.load(str(tmp_path / "scene.luisa"))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 8)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    print(f"address {address} {filter_mode} scale {scale}: rel-L1 {_rel_l1(gpu, cpu):.3e}, rays {gc['closest_rays']} / {cc['closest_rays']}")
    if scale == "2.5, -3":
        assert gc["closest_rays"] == cc["closest_rays"]
        assert _rel_l1(gpu, cpu) < 1e-5 and gpu[..., :3].mean() > 0.01
    else:  # uv rounds to ~1e-2 of a texel up there and the device contracts u * scale + offset: a texel flips now and then
        assert _rel_l1(gpu, cpu) < 6e-2
        assert abs(float(gpu[..., :3].mean()) - float(cpu[..., :3].mean())) < 2e-3 * float(cpu[..., :3].mean()) + 1e-6


@pytest.mark.parametrize("gamma", ["2.2", "2", "3"])
def test_gamma_encoded_texture_with_zero_and_negative_texels(renderer, tmp_path, gamma):
    """image.cpp:151-152: `scale * pow(rgba, gamma)`.  The device takes x^g as exp2(g log2 x) for positive texels and the cases of IEEE
    pow otherwise (dev_shade.h: pow_nonpositive: 0^g = 0, (-x)^g = +-x^g for an integer g, NaN for a fractional one) instead of libm's
    powf: a picture with exact zeros everywhere and negative texels under the integer exponents."""
    from test_oracle_vs_ref import ADDRESS_SCENE, _write_pfm
    y, x = np.mgrid[0:16, 0:16]
    pic = np.stack([0.1 + x / 20.0, 0.9 - y / 20.0, 0.3 + 0.02 * ((x + y) % 5)], axis=-1).astype(np.float32)
    pic[::3, ::2] = 0.0
    if gamma != "2.2":
        pic[1::4, 1::3] *= -1.0
    _write_pfm(tmp_path / "tex.pfm", pic)
    text = ADDRESS_SCENE.format(address="repeat", filter="point", spp=8, res="96, 64").replace('encoding { "linear" }', 'encoding { "gamma" } gamma { %s }' % gamma)
    (tmp_path / "scene.luisa").write_text(text)
    sc = Scene.load(str(tmp_path / "scene.luisa"))
    gpu, gc, cpu, cc = _render_both(renderer, sc, 8)
    assert np.isfinite(gpu).all() and np.array_equal(gpu[..., 3], cpu[..., 3])
    assert abs(gc["closest_rays"] - cc["closest_rays"]) <= 2e-4 * cc["closest_rays"]
    assert _rel_l1(gpu, cpu) < 2e-4 and gpu[..., :3].mean() > 0.005


def test_film_reduce_through_the_c_abi_single_rank(renderer):
    """The product's collective end to end through include/lrhip.h: lrhip_comm_unique_id -> lrhip_comm_init_rank (world 1) ->
    lrhip_film_reduce (ncclReduce on the context's stream) -> lrhip_comm_destroy.  With one rank the sum-reduce must leave the
    film exactly as it was.  (World > 1 needs > 1 GPU: bench.py --gpus N runs exactly these calls, test_cli.py the C++ host.)"""
    from luisarender_amd import _ffi
    from luisarender_amd.parallel import FilmReducer
    sc = Scene.from_string(cornell_box(resolution=32, spp=4))
    renderer.upload(sc)
    renderer.render(0, 4, sync=True)
    before = renderer.download(converted=False)
    comm = renderer.comm_init_rank(1, 0, renderer.comm_unique_id())
    try:
        renderer.film_reduce(comm, 0)
        renderer.synchronize()
        after = renderer.download(converted=False)
        assert np.array_equal(before, after) and before[..., 3].min() == 4
        lib = _ffi.hip_lib()
        assert lib.lrhip_film_reduce(renderer._ctx, None, 0) < 0  # NULL communicator: error code, no crash
    finally:
        renderer.comm_destroy(comm)
    assert FilmReducer(renderer, 0, 1).comm is None  # one rank: no communicator, reduce() is a no-op


def test_strided_shards_are_diagonals_and_empty_shards_are_legal(renderer):
    """include/lrhip.h: tile numbers rotate every row by its index, so a strided shard is a set of diagonals (not of column stripes,
    which tile % world gives whenever the tile-row length is a multiple of the world size); rank >= tile count is an empty shard."""
    from luisarender_amd.parallel import owner_mask
    sc = Scene.from_string(cornell_box(resolution=64, spp=2))
    renderer.upload(sc)
    for world in (2, 4, 8):
        for rank in (0, world - 1):
            renderer.clear()
            renderer.render(0, 2, rank=rank, world=world, sync=True)
            owned = renderer.download(converted=False)[..., 3] == 2
            assert np.array_equal(owned, owner_mask(64, 64, rank, world)), (rank, world)
            cols = owned.any(axis=0)
            assert cols.all()  # every pixel column holds tiles of this rank: no stripes
    tiny = Scene.from_string(cornell_box(resolution=8, spp=1))  # one tile
    renderer.upload(tiny)
    renderer.render(0, 1, rank=5, world=8, sync=True)  # more ranks than tiles: nothing to do, no error
    assert not renderer.download(converted=False).any()


def test_bound_film_survives_an_upload(renderer):
    """lrhip_bind_film + MegaPathRenderer.render_frame (which uploads per shutter sample): the caller's buffer stays bound across
    uploads of the same resolution (round-1 advisor finding: it was silently dropped, the tensor stayed zero)"""
    import torch
    sc = Scene.from_string(cornell_box(resolution=32, spp=4))
    renderer.upload(sc)
    film = torch.zeros((32, 32, 4), dtype=torch.float32, device="cuda:0")
    renderer.bind_film(film.data_ptr())
    renderer.render_frame(sc)  # uploads again
    renderer.synchronize()
    torch.cuda.synchronize()
    got = film.cpu().numpy()
    assert (got[..., 3] == 4).all() and got[..., :3].sum() > 0
    assert np.array_equal(got, renderer.download(converted=False))
    renderer.bind_film(None)
    renderer.upload(Scene.from_string(cornell_box(resolution=16, spp=1)))  # another resolution: back to the library's film
    renderer.render(0, 1, sync=True)
    assert renderer.download(converted=False).shape == (16, 16, 4)


def test_nested_mix_layered_only_in_megapath(renderer):
    """Mix / Layered nested in each other live in their own kernel variant (636); the sibling and volumetric kernels refuse them."""
    from helpers import MATERIALS
    from luisarender_amd.render import DeviceError
    extra = MATERIALS["mix_layered"].replace("Surface m ", "Surface mix_layered ") + "\n"
    text = cornell_box(resolution=32, spp=4, short_box_surface="mix_layered", extra_surfaces=extra)
    renderer.upload(Scene.from_string(text))
    renderer.render(0, 4, sync=True)
    assert _variant(renderer) == WF | 16 | 32 | 64 | 512
    for integrator in ('Direct { importance_sampling { "both" }', "MegaVPTNaive {"):
        with pytest.raises(DeviceError, match="MegaPath integrator only"):
            renderer.upload(Scene.from_string(text.replace("integrator : MegaPath {", "integrator : " + integrator)))


def test_wavefront_mode_is_deterministic_shardable_and_agrees_with_the_all_in_one_kernel(renderer, tmp_path):
    """Scenes with Mix / Layered surfaces render in wavefront mode (lean megakernel that parks heavy hits -> heavy-closure kernel ->
    continuation pass, dev_scene.h: WfArgs).  Paths that were parked finish in whatever wave picked their record up, so their film
    adds are 64-bit fixed-point atomics: the film must still be bit-identical run to run and under tile sharding.  Slicing the frame
    differently (smaller queues) regroups float sums: equal to rounding.  And the all-in-one megakernel <124> (lrhip_set_wavefront
    off) renders the same estimator: every path draws the same random numbers, so the two films agree far below the noise."""
    sc = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(256, 144), spp=16, target_triangles=60_000))
    renderer.upload(sc)
    films = []
    for _ in range(2):
        renderer.clear()
        renderer.render(0, 16, sync=True)
        films.append(renderer.download(False))
    assert _variant(renderer) == WF | 16 | 32 | 64  # (the stand-in at this size holds no alpha-tested surface)
    assert np.array_equal(films[0], films[1]) and np.isfinite(films[0]).all() and (films[0][..., 3] == 16).all()
    total = np.zeros_like(films[0])
    for rank in range(3):
        renderer.clear()
        renderer.render(0, 16, rank=rank, world=3, sync=True)
        total += renderer.download(False)
    assert np.array_equal(total, films[0])
    try:
        renderer.set_wavefront(True, slice_paths=256 * 144 * 3)  # 3 spp per slice: six slices, the last one short
        renderer.clear()
        renderer.render(0, 16, sync=True)
        sliced = renderer.download(False)
        assert np.array_equal(sliced[..., 3], films[0][..., 3]) and np.allclose(sliced, films[0], rtol=2e-5, atol=1e-5)
        # slices are cut by the frame (a nominal shard's tiles), not by the tile range of the call or by the free memory: the shards of
        # the sliced frame still add up to it bit for bit, and so does the frame pushed through the queues eight tiles at a time
        total = np.zeros_like(sliced)
        for rank in range(3):
            renderer.clear()
            renderer.render(0, 16, rank=rank, world=3, sync=True)
            total += renderer.download(False)
        assert np.array_equal(total, sliced)
        renderer.set_wavefront(True, slice_paths=256 * 144 * 3, tiny_tile_groups=True)
        renderer.clear()
        renderer.render(0, 16, sync=True)
        assert np.array_equal(renderer.download(False), sliced)
        renderer.set_wavefront(False)
        renderer.clear()
        renderer.render(0, 16, sync=True)
        assert _variant(renderer) == 124
        mono = renderer.download(False)
    finally:
        renderer.set_wavefront(True)
    err = _rel_l1(films[0], mono)
    print(f"wavefront vs all-in-one <124>: rel-L1 {err:.2e}")
    assert np.array_equal(mono[..., 3], films[0][..., 3]) and err < 2e-2 and abs(films[0][..., :3].mean() - mono[..., :3].mean()) / mono[..., :3].mean() < 2e-3


def test_wavefront_mode_keeps_negative_samples(renderer, tmp_path):
    """Mitchell and Lanczos filters have negative lobes: the filter weight (f / pdf, filter.cpp:49-64) and with it a sample's radiance can
    be negative.  A path that finishes outside its tile's wave adds in fixed point, and round 3 clamped those adds at zero -- brighter
    frames in wavefront mode only.  Signed adds now: wavefront mode and the all-in-one kernel (plain float adds, like the reference's
    ColorFilmInstance::_accumulate) render the same frame, negative pixels included."""
    path = generate_kitchen_scene(str(tmp_path), resolution=(192, 108), spp=12, target_triangles=40_000)
    text = open(path).read()
    assert "filter : Gaussian { radius { 1 } }" in text
    mitchell = tmp_path / "kitchen_mitchell.luisa"
    mitchell.write_text(text.replace("filter : Gaussian { radius { 1 } }", "filter : Mitchell { radius { 2 } }"))
    sc = Scene.load(str(mitchell))
    renderer.upload(sc)
    renderer.render(0, 12, sync=True)
    wave = renderer.download(False)
    assert (_variant(renderer) & WF) != 0
    try:
        renderer.set_wavefront(False)
        renderer.clear()
        renderer.render(0, 12, sync=True)
        mono = renderer.download(False)
        assert (_variant(renderer) & WF) == 0
    finally:
        renderer.set_wavefront(True)
    neg = mono[..., :3] < 0.0
    print(f"Mitchell r = 2: {int(neg.sum())} negative channel sums of {neg.size}; wavefront vs all-in-one rel-L1 {_rel_l1(wave, mono):.2e}")
    assert neg.sum() > 0  # (the scene does produce negative sums: what round 3 lost)
    assert np.array_equal(wave[..., 3], mono[..., 3])
    assert abs(wave[..., :3].mean() - mono[..., :3].mean()) / abs(mono[..., :3].mean()) < 2e-3 and _rel_l1(wave, mono) < 2e-2
    assert abs(wave[..., :3][neg].sum() - mono[..., :3][neg].sum()) <= 0.05 * abs(mono[..., :3][neg].sum()) + 1e-3


def test_c_abi_rejects_closure_trees_the_interpreters_cannot_walk(renderer):
    """lrhip_upload_scene walks every Mix / Layered tree itself: what the C++ loader refuses -- more than two
    Layered levels on a path through the interfaces, a Mix tree whose recorded depth (u[2]) is not its depth, a cycle -- is an
    error for a C-ABI caller too, not silently wrong shading.  The host tables of a valid scene are tampered with in place."""
    from helpers import MATERIALS
    from luisarender_amd.render import DeviceError
    extra = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("mix_layered", "mix_nested"))
    sc = Scene.from_string(cornell_box(resolution=16, spp=1, short_box_surface="mix_layered", tall_box_surface="mix_nested", extra_surfaces=extra))
    view = sc.view(0)
    renderer.upload(sc)  # valid as loaded
    surfaces = [view.surfaces[i] for i in range(view.surface_count)]
    mixes = [i for i, x in enumerate(surfaces) if x.kind == 7]  # LR_SURFACE_MIX
    layered = [i for i, x in enumerate(surfaces) if x.kind == 8]  # LR_SURFACE_LAYERED
    assert mixes and layered, [x.kind for x in surfaces]
    deep = max(mixes, key=lambda i: surfaces[i].u[2])
    saved = surfaces[deep].u[2]
    view.surfaces[deep].u[2] = saved + 1  # a wrong depth
    with pytest.raises(DeviceError, match="u\\[2\\] is not the depth"):
        renderer.upload(sc)
    view.surfaces[deep].u[2] = saved
    lay = layered[0]
    top = view.surfaces[lay].u[0]
    holder = next(i for i in mixes if lay in (surfaces[i].u[0], surfaces[i].u[1]))  # Mix -> Layered ...
    view.surfaces[lay].u[0] = holder                                                 # ... -> Mix -> Layered: a cycle through a Layered interface
    with pytest.raises(DeviceError, match="Layered surfaces nested more than 2 levels deep|cyclic"):
        renderer.upload(sc)
    view.surfaces[lay].u[0] = top
    renderer.upload(sc)  # and valid again


def test_c_abi_rejects_malformed_environment_trees(renderer):
    """lrhip_upload_scene checks the tree of Combined environment records itself (lr_scene.h: children before parents, positive
    scales, at most LR_ENV_MAX_COMBINED_DEPTH levels): a C-ABI caller's cycle or dangling index is an error, not a device fault."""
    import ctypes as C
    from luisarender_amd.render import DeviceError
    sun = "Directional { emission : Constant { v { 1, 2, 3 } } angle { 10 } direction { 0.2, 1, 0 } }"
    env = f"Combined {{ a : {sun} b : Combined {{ a : {sun} b : {sun} scale_b {{ 2 }} }} }}"
    sc = Scene.from_string(ENV_SCENE.format(env=env, spp=1))
    view = sc.view(0)
    renderer.upload(sc)  # valid as loaded
    kids = C.cast(view.environment_children, C.POINTER(type(view.environment)))
    assert view.environment_child_count == 4 and kids[3].kind == 3
    kids[3].child[1] = 3  # a node that is its own child
    with pytest.raises(DeviceError, match="children precede their parents"):
        renderer.upload(sc)
    kids[3].child[1] = 1
    view.environment.child[0] = 4  # past the end of the array
    with pytest.raises(DeviceError, match="child index out of range"):
        renderer.upload(sc)
    view.environment.child[0] = 2
    kids[3].child_scale[0] = 0.0
    with pytest.raises(DeviceError, match="scales must be positive"):
        renderer.upload(sc)
    kids[3].child_scale[0] = 1.0
    renderer.upload(sc)  # and valid again


@pytest.mark.parametrize("sampler", ["Independent", "Sobol"])
def test_every_scene_feature_variant_renders_the_same_frame(renderer, sampler):
    """The SAME scene through every precompiled scene-feature variant (lrhip_set_diagnostics: a per-context override of the variant
    choice for tests and tools): a binary that holds more features than the scene needs must still render the scene -- same sample counts, the same
    image up to the rounding of a different instruction schedule, identical when run twice.  This reaches the shipped binaries
    that no BASELINE stand-in launches with this sampler (<12>, <62>, <126>, ...)."""
    from helpers import MATERIALS
    extra = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("glass", "metal"))
    sc = Scene.from_string(cornell_box(resolution=48, spp=8, short_box_surface="glass", tall_box_surface="metal", extra_surfaces=extra, sampler=sampler))
    generic = 0 if sampler == "Independent" else 2
    renderer.upload(sc)
    renderer.render(0, 8, counters=False, sync=True)
    assert _variant(renderer) == generic
    base = renderer.download(converted=False)
    try:
        for force in (4, 8, 12, 16, 20, 60, 124):
            renderer.set_diagnostics(force_features=force)
            films = []
            for _ in range(2):
                renderer.upload(sc)
                renderer.render(0, 8, counters=False, sync=True)
                assert _variant(renderer) == (force | generic), (force, _variant(renderer))
                films.append(renderer.download(converted=False))
            assert np.array_equal(films[0], films[1]), force
            assert np.array_equal(films[0][..., 3], base[..., 3]) and np.isfinite(films[0]).all(), force
            err = _rel_l1(films[0], base)
            print(f"variant {force | generic}: rel-L1 vs <{generic}> {err:.2e}")
            assert err < 3e-3, (force, err)
    finally:
        renderer.set_diagnostics()


def test_two_contexts_driven_from_two_host_threads_render_one_frame(tmp_path):
    """The C++ host's multi-GPU model (plugin_megapath.cpp: one context + one host thread per GPU) on ONE device: two lrhip_ctx on
    device 0, driven concurrently from two host threads, render complementary tile shards of a frame; their films, added, are bit-equal
    to one context's full frame (same balance_shards).  Exercises what a second rank exercises short of RCCL: per-context state
    (streams, work counters, spill areas, scene records), two persistent grids sharing the CUs, the free-memory-sized queues of
    wavefront mode under contention, and the thread-local error string."""
    import threading
    from luisarender_amd.render import DeviceError, MegaPathRenderer
    scenes = {"lean": Scene.load(generate_room_scene(str(tmp_path / "room"), target_triangles=60_000, resolution=(256, 160), spp=8)),
              "wavefront": Scene.load(generate_kitchen_scene(str(tmp_path / "kitchen"), resolution=(192, 112), spp=8, target_triangles=40_000))}
    for name, sc in scenes.items():
        single = MegaPathRenderer(0)
        single.upload(sc)
        single.render(0, 8, balance_shards=2, sync=True)
        full = single.download(False)
        single.close()
        films, errors, messages = [None, None], [], [None, None]
        barrier = threading.Barrier(2)

        def work(rank):
            try:
                r = MegaPathRenderer(0)
                r.upload(sc)
                barrier.wait()
                for _ in range(3):  # (several rounds: the two grids meet in different phases)
                    r.clear()
                    r.render(0, 8, rank=rank, world=2, balance_shards=2, sync=True)
                films[rank] = r.download(False)
                # an error of THIS thread's context: its text must not leak into (or come from) the other thread
                try:
                    r.render(5, 3, rank=rank, world=2)
                except DeviceError as e:
                    messages[rank] = str(e)
                barrier.wait()
                r.close()
            except Exception as e:  # noqa: BLE001
                errors.append((rank, repr(e)))
                barrier.abort()

        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        assert all(m is not None and "spp" in m for m in messages), messages
        total = films[0] + films[1]
        assert np.isfinite(total).all() and (total[..., 3] == 8).all(), name
        assert np.array_equal(total, full), name


def test_byte_texels_decode_to_the_floats_the_host_made(tmp_path):
    """lrhip_set_texture_storage (round 5): an image whose texels are 8-bit codes' floats stays 8 bits per channel on the device
    (lrhip.hip: pack_byte_textures; dev_shade.h: texel_at).  The decode must give back exactly the floats the host readers made --
    b * (1 / 255.f) from the PNG reader, b / 255.f from the JPEG / BMP / TGA readers and from a PFM that holds byte / 255 -- so the
    film of storage mode 2 is the film of mode 0, bit for bit; a float picture that is no 8-bit code stays float."""
    from luisarender_amd.render import MegaPathRenderer
    from luisarender_amd.scenes.configs import write_pfm, write_png
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:37, 0:53]
    rgb = np.stack([(x * 5 + y) % 256, (y * 7 + 3 * x) % 256, rng.integers(0, 256, x.shape)], axis=-1).astype(np.uint8)
    write_png(tmp_path / "rgb.png", rgb)                                                      # b * (1 / 255.f), sRGB by default
    write_png(tmp_path / "grey.png", ((x * 9 + y * 4) % 256).astype(np.uint8))                # one channel, replicated by the reader
    codes = np.stack([rng.integers(0, 256, (16, 16)) for _ in range(3)], axis=-1).reshape(16, 16, 3)
    codes[:4] = np.arange(256 * 3).reshape(-1)[:4 * 16 * 3].reshape(4, 16, 3) % 256           # every code at least once over the picture
    write_pfm(tmp_path / "quot.pfm", codes.astype(np.float32) / np.float32(255.0))            # b / 255.f
    write_pfm(tmp_path / "free.pfm", rng.random((9, 11, 3)).astype(np.float32))               # no 8-bit picture: stays float
    text = """
Shape floor : InlineMesh { positions { -4,0,-4, 4,0,-4, 4,0,4, -4,0,4 } indices { 0,2,1, 0,3,2 } uvs { 0,0, 1,0, 1,1, 0,1 }
  surface : Matte { Kd : Image { file { "rgb.png" } uv_scale { 2.5, -3 } filter { "bilinear" } } } }
Shape back : InlineMesh { positions { -4,0,-4, 4,0,-4, 4,5,-4, -4,5,-4 } indices { 0,1,2, 0,2,3 } uvs { 0,0, 1,0, 1,1, 0,1 }
  surface : Plastic { Kd : Image { file { "quot.pfm" } encoding { "linear" } filter { "point" } } roughness : Image { file { "grey.png" } encoding { "linear" } } eta : Constant { v { 1.5 } } } }
Shape side : InlineMesh { positions { -4,0,4, -4,0,-4, -4,5,-4, -4,5,4 } indices { 0,1,2, 0,2,3 } uvs { 0,0, 1,0, 1,1, 0,1 }
  surface : Matte { Kd : Image { file { "free.pfm" } encoding { "linear" } filter { "bilinear" } } } }
Shape lamp : InlineMesh { positions { -1,4.9,-1, 1,4.9,-1, 1,4.9,1, -1,4.9,1 } indices { 0,1,2, 0,2,3 } light : Diffuse { emission : Constant { v { 12, 11, 10 } } } }
Camera cam : Pinhole { fov { 55 } spp { 8 } film : Color { resolution { 96, 64 } } position { 2, 2.5, 7 } look_at { -0.5, 1.5, 0 } }
render { cameras { @cam } shapes { @floor, @back, @side, @lamp } integrator : MegaPath { depth { 5 } } }
"""
    (tmp_path / "scene.luisa").write_text(text)
    sc = Scene.load(str(tmp_path / "scene.luisa"))
    films, packed = {}, {}
    for mode in (0, 1, 2):
        r = MegaPathRenderer(0)
        r.set_texture_storage(mode)
        r.upload(sc)
        r.render(0, 8, sync=True)
        films[mode], packed[mode] = r.download(converted=False), r.packed_texels()
        r.close()
    assert packed[0] == 0 and packed[1] == 0                         # automatic: a few kB of texels stay float
    assert packed[2] == 37 * 53 * 2 + 16 * 16, packed                 # the two PNGs and the byte / 255 PFM, not the free-valued one
    assert np.isfinite(films[0]).all() and films[0][..., :3].mean() > 0.01
    assert np.array_equal(films[0], films[2]) and np.array_equal(films[0], films[1])
    with pytest.raises(Exception):
        MegaPathRenderer(0).set_texture_storage(3)
