"""the volumetric megakernel MegakernelVolumePathTracingNaive (src/integrators/mega_vpt_naive.cpp:68-483) with
Homogeneous / Vacuum media (src/media), the Henyey-Greenstein phase function and the priority MediumTracker
(src/util/medium_tracker.cpp) — oracle pins.  The integrator's MIS weights are "naive" (1 / sum of pdfs, :296,409), so closed
forms exist only where next-event estimation cannot contribute."""
import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box

SLAB = """
Medium fog : Homogeneous {{ sigma_a : Constant {{ v {{ {sa} }} }} sigma_s : Constant {{ v {{ {ss} }} }} eta {{ 1 }} priority {{ {prio} }}
  phasefunction : HenyeyGreenstein {{ g {{ {g} }} }} }}
Shape lamp : InlineMesh {{ positions {{ -50,-50,0, 50,-50,0, 50,50,0, -50,50,0 }} indices {{ 0,1,2, 0,2,3 }}
  light : Diffuse {{ emission : Constant {{ v {{ 2, 3, 4 }} }} two_sided {{ true }} }}
  transform : SRT {{ rotate {{ 0.3, 1, 0.1, 9 }} }} }}
{extra}
Camera cam : Pinhole {{ fov {{ 1 }} spp {{ 1 }} film : Color {{ resolution {{ 8, 8 }} clamp {{ 1000000 }} }}
  position {{ 0, 0, {dist} }} look_at {{ 0, 0, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @lamp{shapes} }} {env_medium}
  integrator : MegaVPTNaive {{ depth {{ 16 }} rr_depth {{ 1000 }} }} }}
"""


def _mean(scene, spp):
    o = Oracle(scene)
    film, counters = o.render(0, spp)
    # after a medium "hit surface" event the ray origin sits ON the surface (homogeneous.cpp:64), so evaluate_hit normalises a
    # zero vector now and then (diffuse.cpp:72): NaN radiance, which the film rejects (color.cpp:110) — as in the reference
    assert np.isfinite(film).all() and (film[..., 3] >= 0.99 * spp).all()
    return o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0), counters


def test_loader_registers_media_like_the_pipeline():
    sc = Scene.from_string(SLAB.format(sa="0.1, 0.2, 0.3", ss="0.5", g=0.3, prio=2, dist=2, extra="", shapes="",
                                       env_medium="environment_medium { @fog }"), build_accel=False)
    v = sc.view()
    assert v.integrator.kind == 3 and v.integrator.max_depth == 16 and v.medium_count == 1
    assert v.integrator.environment_medium_tag == 0
    m = v.media[0]
    assert m.kind == 1 and m.priority == 2 and abs(m.g - 0.3) < 1e-7 and abs(m.eta - 1) < 1e-7
    assert np.allclose(list(m.sigma_a), [0.1, 0.2, 0.3]) and np.allclose(list(m.sigma_s), [0.5] * 3)
    with pytest.raises(Exception, match="must be specified as constant"):
        Scene.from_string(SLAB.format(sa="0.1", ss="0.5", g=0, prio=0, dist=2, extra="", shapes="", env_medium="environment_medium { @fog }")
                          .replace("sigma_s : Constant { v { 0.5 } }", ""), build_accel=False)


@pytest.mark.parametrize("dist", [1.0, 2.5])
def test_absorbing_environment_medium_attenuates_by_beer_lambert(dist):
    """Camera and lamp inside a purely absorbing medium: L = L_e * exp(-sigma_a d) per channel (distance sampling with a
    random channel choice, homogeneous.cpp:48-80, is unbiased; the only NEE sample has weight ~1e-16)."""
    sigma = np.array([0.2, 0.5, 1.0])
    sc = Scene.from_string(SLAB.format(sa=", ".join(map(str, sigma)), ss="0", g=0, prio=0, dist=dist, extra="", shapes="",
                                       env_medium="environment_medium { @fog }"))
    img, _ = _mean(sc, 4096)
    # (the lamp is tilted: on an axis-aligned lamp the "hit surface" origin o + d * t often lands EXACTLY in the lamp's plane,
    # the direction to it is then perpendicular to the normal and diffuse.cpp:84 zeroes the emission — a rounding lottery)
    expect = np.array([2.0, 3.0, 4.0]) * np.exp(-sigma * dist)
    assert np.allclose(img, expect, rtol=0.04), (img, expect)


def test_vacuum_scene_matches_the_path_tracer():
    """Without media the transmittance walk degenerates to a visibility test and the naive weights to the balance heuristic:
    the volumetric integrator converges to MegaPath's image (different sample streams: compared by mean)."""
    vpt = Scene.from_string(cornell_box(resolution=32, spp=256, depth=6).replace("integrator : MegaPath {", "integrator : MegaVPTNaive {"))
    path = Scene.from_string(cornell_box(resolution=32, spp=256, depth=6))
    a, _ = Oracle(vpt).render(0, 256)
    b, _ = Oracle(path).render(0, 256)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) / b[..., :3].mean() < 0.02
    blocks = lambda f: f[..., :3].reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
    assert np.abs(blocks(a) - blocks(b)).sum() / np.abs(blocks(b)).sum() < 0.05


def test_medium_boundary_enter_and_exit_through_an_index_matched_box():
    """A box of absorbing medium behind an (almost) index-matched smooth Glass boundary between the camera and the lamp:
    enter -> distance sampling inside -> exit, i.e. the MediumTracker round trip; L ~ L_e * exp(-sigma_a * thickness)."""
    box = """
Surface skin : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } eta : Constant { v { 1.0001 } } }
Shape box : InlineMesh {
  positions { -20,-20,1, 20,-20,1, 20,20,1, -20,20,1, -20,-20,2, 20,-20,2, 20,20,2, -20,20,2 }
  indices { 0,2,1, 0,3,2,  4,5,6, 4,6,7,  0,1,5, 0,5,4,  3,6,2, 3,7,6,  0,7,3, 0,4,7,  1,2,6, 1,6,5 }
  surface { @skin } medium { @fog } }
"""
    sigma = np.array([0.3, 0.6, 0.9])
    sc = Scene.from_string(SLAB.format(sa=", ".join(map(str, sigma)), ss="0", g=0, prio=0, dist=4, extra=box, shapes=", @box", env_medium=""))
    v = sc.view()
    assert v.medium_count == 1 and v.integrator.environment_medium_tag == 0xFFFFFFFF
    assert (v.instances[1].handle.x & 16) != 0 and (v.instances[1].handle.y >> 24) == 0  # LR_SHAPE_HAS_MEDIUM, medium tag 0
    img, _ = _mean(sc, 4096)
    expect = np.array([2.0, 3.0, 4.0]) * np.exp(-sigma * 1.0)
    assert np.allclose(img, expect, rtol=0.05), (img, expect)


def test_scattering_medium_keeps_energy_between_the_absorbing_bounds():
    """sigma_s > 0: more than the unscattered beam arrives (in-scattering), never more than the lamp's radiance."""
    sc = Scene.from_string(SLAB.format(sa="0.1", ss="0.6", g=0.5, prio=0, dist=2, extra="", shapes="", env_medium="environment_medium { @fog }"))
    img, counters = _mean(sc, 2048)
    beam = np.array([2.0, 3.0, 4.0]) * np.exp(-0.7 * 2.0)
    assert (img > beam * 0.98).all() and (img < np.array([2.0, 3.0, 4.0])).all()
    assert counters["closest_rays"] > 2 * counters["paths"]  # scattered paths + transmittance walks
