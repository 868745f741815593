"""motion blur — animated transforms (src/transforms/lerp.cpp, src/util/xform.cpp), the camera shutter
(src/base/camera.cpp:22-131,150-203) and the per-shutter-sample scene update (src/base/pipeline.cpp:101-113,
src/base/geometry.cpp:194-216, src/base/integrator.cpp:86-107).  CPU pins: closed forms of the transform interpolation, the
invariants of Camera::shutter_samples, the refitted BVH against a rebuilt one, the time-averaged coverage of a moving emitter;
the device parity test renders the same frame through lrhip_update_scene + shutter weights."""
import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle

# an emissive strip of width 0.5 sweeps from x = -2 to x = 2 in front of a black backdrop while the shutter is open;
# the orthographic camera looks down -z
MOVING_STRIP = """
Shape strip : InlineMesh { positions { -0.25,-4,0, 0.25,-4,0, 0.25,4,0, -0.25,4,0 } indices { 0,1,2, 0,2,3 }
  light : Diffuse { emission : Constant { v { 3, 2, 1 } } two_sided { true } }
  transform : Lerp { time_points { 0, 1 }
    transforms { SRT { translate { -2, 0, 0 } }, SRT { translate { 2, 0, 0 } } } } }
Shape backdrop : InlineMesh { positions { -9,-9,-1, 9,-9,-1, 9,9,-1, -9,9,-1 } indices { 0,1,2, 0,2,3 }
  surface : Matte { Kd : Constant { v { 0 } } } }
Camera cam : Ortho { zoom { -1 } spp { SPP } film : Color { resolution { 32, 8 } } filter : Box { radius { 0.5 } }
  position { 0, 0, 5 } look_at { 0, 0, 0 } shutter_span { 0, 1 } shutter_samples { SHUTTER } }
render { cameras { @cam } shapes { @strip, @backdrop } integrator : MegaPath { depth { 2 } sampler : Independent { seed { 7 } } } }
"""


def _strip_scene(spp=64, shutter=32):
    return Scene.from_string(MOVING_STRIP.replace("SPP", str(spp)).replace("SHUTTER", str(shutter)))


def _instance_matrix(scene, i=0):
    v = scene.view()
    return np.array(v.instances[i].object_to_world[:], np.float32).reshape(4, 4).T  # column-major -> rows


def test_lerp_translation_is_linear_and_clamped():
    sc = _strip_scene()
    for t, x in [(-1.0, -2.0), (0.0, -2.0), (0.25, -1.0), (0.5, 0.0), (1.0, 2.0), (3.0, 2.0)]:
        assert sc.set_time(t)
        m = _instance_matrix(sc)
        assert np.allclose(m[:3, 3], [x, 0, 0], atol=1e-6), (t, m)
        assert np.allclose(m[:3, :3], np.eye(3), atol=1e-6)


def test_lerp_rotation_is_a_slerp_and_scaling_a_lerp():
    """decompose (polar iteration, xform.cpp:12-43) + slerp (:90-99): half way between 0 and 90 degrees about z is 45
    degrees; scale 1 -> 3 passes 2; a Stack applies its children first to last (stack.cpp:38-47)"""
    sc = Scene.from_string("""
Shape a : InlineMesh { positions { 0,0,0, 1,0,0, 0,1,0 } indices { 0,1,2 } light : Diffuse { emission : Constant { v { 1 } } }
  transform : Stack { transforms {
    Lerp { time_points { 2, 0 } transforms { SRT { rotate { 0, 0, 1, 90 } scale { 3 } }, SRT { scale { 1 } } } },
    SRT { translate { 0, 0, 1 } } } } }
Camera cam : Pinhole { spp { 1 } film : Color { resolution { 4, 4 } } position { 0, 0, 5 } look_at { 0, 0, 0 } }
render { cameras { @cam } shapes { @a } integrator : MegaPath { } }
""")
    sc.set_time(1.0)
    m = _instance_matrix(sc)
    c = np.cos(np.pi / 4) * 2.0
    assert np.allclose(m[:3, :3], [[c, -c, 0], [c, c, 0], [0, 0, 2]], atol=1e-5), m
    assert np.allclose(m[:3, 3], [0, 0, 1], atol=1e-6)
    sc.set_time(2.0)  # the key at the later time point (time points are sorted, lerp.cpp:41-66)
    assert np.allclose(_instance_matrix(sc)[:3, :3], [[0, -3, 0], [3, 0, 0], [0, 0, 3]], atol=1e-5)


def test_shutter_samples_invariants():
    sc = _strip_scene(spp=70, shutter=32)
    s = sc.shutter_samples()
    assert len(s) == 32
    assert sum(n for _, _, n in s) == 70 and {n for _, _, n in s} == {2, 3}  # spp spread over the buckets, remainder shuffled
    for b, (t, w, n) in enumerate(s):
        assert b / 32 <= t <= (b + 1) / 32  # one jittered time per bucket (camera.cpp:172-178)
        assert w == pytest.approx(1.0, abs=1e-6)  # uniform curve: sum(w * spp) = spp
    # a static camera has ONE shutter sample at shutter_span.x
    static = Scene.from_string(MOVING_STRIP.replace("SPP", "5").replace("shutter_span { 0, 1 } shutter_samples { SHUTTER }", "shutter_span { 0.25 }"))
    assert static.shutter_samples() == [(0.25, 1.0, 5)]
    assert np.allclose(_instance_matrix(static)[:3, 3], [-1, 0, 0], atol=1e-6)  # built at the initial time, pipeline.cpp:50-56
    # a tent curve: weights follow the curve and are normalised to sum(w * spp) = spp (camera.cpp:189-201)
    tent = Scene.from_string(MOVING_STRIP.replace("SPP", "64").replace("shutter_samples { SHUTTER }",
                                                                       "shutter_samples { 16 } shutter_time_points { 0, 0.5, 1 } shutter_weights { 0, 1, 0 }"))
    ts = tent.shutter_samples()
    assert sum(w * n for _, w, n in ts) == pytest.approx(64.0, rel=1e-5)


def test_refit_equals_rebuild():
    """the refitted BVH (accel.cpp: refit_accel) answers closest-hit queries like the oracle's BVH built from scratch at that time"""
    sc = _strip_scene()
    sc.set_time(0.7)
    v = sc.view()
    nodes = v.accel.nodes
    tri = v.accel.triangles
    # every leaf's box holds its (moved) triangle and every inner child box holds its child's boxes
    for ni in range(v.accel.node_count):
        n = nodes[ni]
        for k in range(4):
            c = n.child[k]
            if c == 0xFFFFFFFF:
                continue
            lo = np.array([n.lo_x[k], n.lo_y[k], n.lo_z[k]])
            hi = np.array([n.hi_x[k], n.hi_y[k], n.hi_z[k]])
            if c & 0x80000000:
                t = tri[c & 0x7FFFFFF]
                p0 = np.array(t.v0[:])
                pts = np.stack([p0, p0 + np.array(t.e1[:]), p0 + np.array(t.e2[:])])
            else:
                ch = nodes[c]
                pts = np.array([[ch.lo_x[j], ch.lo_y[j], ch.lo_z[j]] for j in range(4) if ch.child[j] != 0xFFFFFFFF] +
                               [[ch.hi_x[j], ch.hi_y[j], ch.hi_z[j]] for j in range(4) if ch.child[j] != 0xFFFFFFFF])
            assert (pts >= lo - 1e-6).all() and (pts <= hi + 1e-6).all()
    o = Oracle(sc)
    inst, prim, u, vv, t = o.trace_closest([0.8, 0.0, 5.0], [0.0, 0.0, -1.0])  # the strip is centred at x = 0.8 now
    assert inst == 0 and t == pytest.approx(5.0, abs=1e-5)
    inst, _, _, _, t = o.trace_closest([-1.9, 0.0, 5.0], [0.0, 0.0, -1.0])  # where it was at time 0: backdrop
    assert inst == 1 and t == pytest.approx(6.0, abs=1e-5)


def test_moving_emitter_time_average():
    """A strip of width w crossing the view at constant speed over a distance D covers a point for w / D of the exposure:
    the blurred radiance along its path is L * w / D (here 0.5 / 4 = 1/8)."""
    sc = _strip_scene(spp=64, shutter=64)
    film, counters = Oracle.render_frame(sc)
    assert np.all(film[..., 3] == 64)
    img = film[..., :3] / film[..., 3:4]
    view_half = 32 / 8  # Ortho zoom -1: |x| <= 2^1 * aspect... measured below instead of assumed
    cols = img.mean(axis=0)  # [32, 3]
    lit = cols[:, 0] > 0.01
    assert lit.sum() >= 8
    inner = np.where(lit)[0][2:-2]  # away from the ends of the sweep, where the strip enters / leaves
    assert np.allclose(cols[inner].mean(axis=0), np.array([3, 2, 1]) / 8.0, rtol=0.08), cols[inner].mean(axis=0)
    del view_half


def test_frame_is_the_weighted_sum_of_static_renders():
    """integrator.cpp:86-107: sample ids run on across shutter samples; each one renders the STATIC scene of its time"""
    sc = _strip_scene(spp=8, shutter=4)
    film, _ = Oracle.render_frame(sc)
    ref = np.zeros_like(film)
    begin = 0
    for time, weight, spp in sc.shutter_samples():
        static = Scene.from_string(MOVING_STRIP.replace("SPP", "8").replace("shutter_span { 0, 1 } shutter_samples { SHUTTER }", f"shutter_span {{ {time!r} }}"))
        part, _ = Oracle(static).render(begin, begin + spp)
        ref[..., :3] += np.float32(weight) * part[..., :3]
        ref[..., 3] += part[..., 3]
        begin += spp
    assert np.allclose(film, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_device_motion_blur_matches_oracle():
    from luisarender_amd.render import MegaPathRenderer
    sc = _strip_scene(spp=32, shutter=8)
    r = MegaPathRenderer(0)
    r.render_frame(sc)
    gpu = r.download(converted=False)
    cpu, _ = Oracle.render_frame(sc)
    assert np.array_equal(gpu[..., 3], cpu[..., 3])
    rel = float(np.abs(gpu[..., :3] - cpu[..., :3]).sum() / np.abs(cpu[..., :3]).sum())
    assert rel < 1e-4, rel
    # an animated camera over a static scene: only the camera record changes between shutter samples
    cam = Scene.from_string(MOVING_STRIP.replace("SPP", "16").replace("SHUTTER", "4")
                            .replace("transform : Lerp { time_points { 0, 1 }\n    transforms { SRT { translate { -2, 0, 0 } }, SRT { translate { 2, 0, 0 } } } }", "")
                            .replace("position { 0, 0, 5 } look_at { 0, 0, 0 }",
                                     "transform : Lerp { time_points { 0, 1 } transforms { View { position { -1, 0, 5 } front { 0, 0, -1 } }, View { position { 1, 0, 5 } front { 0, 0, -1 } } } }"))
    assert len(cam.shutter_samples()) == 4
    r.render_frame(cam)
    gpu = r.download(converted=False)
    cpu, _ = Oracle.render_frame(cam)
    rel = float(np.abs(gpu[..., :3] - cpu[..., :3]).sum() / np.abs(cpu[..., :3]).sum())
    assert np.array_equal(gpu[..., 3], cpu[..., 3]) and rel < 1e-4, rel
    r.close()


def test_animated_camera_and_environment_follow_the_time():
    """Pipeline::update re-evaluates EVERY registered animated transform (pipeline.cpp:101-113): cameras and the environment too"""
    text = """
Shape floor : InlineMesh { positions { -5,0,-5, 5,0,-5, 5,0,5, -5,0,5 } indices { 0,2,1, 0,3,2 } surface : Matte { } }
Camera cam : Pinhole { spp { 4 } film : Color { resolution { 8, 8 } } shutter_span { 0, 2 } shutter_samples { 2 }
  transform : Lerp { time_points { 0, 2 } transforms { View { position { 0, 1, 4 } front { 0, 0, -1 } }, View { position { 2, 1, 4 } front { 0, 0, -1 } } } } }
render { cameras { @cam } shapes { @floor } integrator : MegaPath { }
  environment : Directional { emission : Constant { v { 5 } } angle { 5 }
    transform : Lerp { time_points { 0, 2 } transforms { SRT { rotate { 0, 0, 1, 0 } }, SRT { rotate { 0, 0, 1, 90 } } } } } }
"""
    sc = Scene.from_string(text)
    assert sc.set_time(1.0)
    v = sc.view()
    c2w = np.array(v.camera.camera_to_world[:], np.float32).reshape(4, 4).T
    assert np.allclose(c2w[:3, 3], [1, 1, 4], atol=1e-6)  # half way between the two key positions
    e2w = np.array(v.environment.env_to_world[:], np.float32).reshape(3, 3).T
    c = np.cos(np.pi / 4)
    assert np.allclose(e2w, [[c, -c, 0], [c, c, 0], [0, 0, 1]], atol=1e-5)  # 45 degrees about z
    assert np.allclose(np.array(v.environment.world_to_env[:]).reshape(3, 3).T, e2w.T, atol=1e-6)
    sc.set_time(0.0)
    assert np.allclose(np.array(sc.view().environment.env_to_world[:]).reshape(3, 3), np.eye(3), atol=1e-6)
    # the frame renders (two shutter samples, camera and light moved in between) and differs from the static frame at t = 0
    film, _ = Oracle.render_frame(sc)
    static = Scene.from_string(text.replace("shutter_span { 0, 2 } shutter_samples { 2 }", "shutter_span { 0 }"))
    ref, _ = Oracle.render_frame(static)
    assert np.all(film[..., 3] == 4) and np.isfinite(film).all() and not np.allclose(film[..., :3], ref[..., :3])
