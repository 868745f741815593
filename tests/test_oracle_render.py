"""Render-level pins for the oracle: closed-form scenes (the reference has no golden images,
SURVEY §8c iii-iv), reference quirks that must be reproduced, and the repo's own golden fixtures."""
import os

import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

FURNACE = """
Surface s : Matte {{ Kd : Constant {{ v {{ {rho} }} }} }}
Shape box : InlineMesh {{
  positions {{ -1,-1,-1, 1,-1,-1, 1,1,-1, -1,1,-1, -1,-1,1, 1,-1,1, 1,1,1, -1,1,1 }}
  indices {{ 0,1,2, 0,2,3,  4,6,5, 4,7,6,  0,4,5, 0,5,1,  3,2,6, 3,6,7,  0,3,7, 0,7,4,  1,5,6, 1,6,2 }}
  surface {{ @s }}
  light : Diffuse {{ emission : Constant {{ v {{ {e} }} }} two_sided {{ true }} }}
}}
Camera cam : Pinhole {{ fov {{ 60 }} spp {{ 1 }} film : Color {{ resolution {{ 16, 16 }} clamp {{ 1000000 }} }}
  position {{ 0.1, 0.05, 0.2 }} look_at {{ 0.3, 0.2, -1 }} }}
render {{ cameras {{ @cam }} shapes {{ @box }} integrator : MegaPath {{ depth {{ 64 }} rr_depth {{ {rr} }} }} }}
"""


@pytest.mark.parametrize("rr_depth", [0, 5])
def test_white_furnace_closed_form(rr_depth):
    """Closed diffuse box, every wall emits E with albedo rho: L = E / (1 - rho) everywhere."""
    rho, e = 0.5, 0.75
    sc = Scene.from_string(FURNACE.format(rho=rho, e=e, rr=rr_depth))
    o = Oracle(sc)
    film, counters = o.render(0, 64)
    img = o.convert(film)[..., :3]
    expect = e / (1 - rho)
    assert abs(img.mean() - expect) / expect < 0.01, (img.mean(), expect)
    assert np.abs(img.mean(axis=2) - expect).max() / expect < 0.4  # every pixel fluctuates around the same value
    assert (film[..., 3] == 64).all()
    assert counters["closest_rays"] >= counters["surface_hits"] == counters["nee_samples"]


ENV_QUAD = """
Surface s : Matte { Kd : Constant { v { 0.6, 0.4, 0.2 } } }
Shape quad : InlineMesh { positions { -50,0,-50, 50,0,-50, 50,0,50, -50,0,50 } indices { 0,2,1, 0,3,2 } surface { @s } }
Camera cam : Pinhole { fov { 30 } spp { 1 } film : Color { resolution { 16, 16 } }
  position { 0, 5, 0 } look_at { 0, 0, -3 } }
render { cameras { @cam } shapes { @quad }
  environment : Spherical { emission : Constant { v { 2, 3, 4 } } }
  integrator : MegaPath { depth { 4 } } }
"""


def test_diffuse_plane_under_uniform_environment():
    """Convex receiver under a constant environment: L = rho * L_env (SURVEY §8c iii)."""
    sc = Scene.from_string(ENV_QUAD)
    o = Oracle(sc)
    film, _ = o.render(0, 256)
    img = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
    expect = np.array([0.6 * 2, 0.4 * 3, 0.2 * 4])
    assert np.allclose(img, expect, rtol=0.02), (img, expect)
    assert sc.view().integrator.env_prob == 1.0  # environment only (uniform.cpp:39-41)


def test_alpha_tested_plane_closed_form():
    """Geometry::_alpha_skip (geometry.cpp:165-192): a plane of opacity a under a constant environment shows
    a * rho * L_env + (1 - a) * L_env; a checkerboard opacity of 0 / 1 averages the two cases."""
    for alpha, a_mean in (("Constant { v { 0.25 } }", 0.25),
                          ("Checkerboard { on : Constant { v { 1 } } off : Constant { v { 0 } } scale { 64 } }", 0.5)):
        sc = Scene.from_string(ENV_QUAD.replace("Surface s : Matte {", "Surface s : Matte { alpha : " + alpha))
        assert sc.view().any_non_opaque == 1
        o = Oracle(sc)
        film, _ = o.render(0, 1024)
        img = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
        env = np.array([2.0, 3.0, 4.0])
        expect = a_mean * np.array([0.6, 0.4, 0.2]) * env + (1 - a_mean) * env
        assert np.allclose(img, expect, rtol=0.02), (alpha, img, expect)
    opaque = Scene.from_string(ENV_QUAD.replace("Surface s : Matte {", "Surface s : Matte { alpha : Constant { v { 1 } }"))
    assert opaque.view().any_non_opaque == 0  # OpacitySurfaceWrapper::maybe_non_opaque, surface.h:177-181


def test_directly_visible_light_is_its_emission():
    sc = Scene.from_string(cornell_box(resolution=64, spp=2))
    o = Oracle(sc)
    # the lamp quad is seen by the centre-top pixels; camera rays carry weight 1 and pdf_bsdf = 1e16 -> MIS weight 1
    v = sc.view()
    found = 0
    for py in range(0, 20):
        for px in range(24, 40):
            ray = o.camera_ray(px, py, 0)
            inst, prim, *_ = o.trace_closest(ray[:3], ray[3:6])
            if inst != 0xffffffff and (v.instances[inst].handle.x & 8):
                L = o.li(px, py, 0)
                assert (L >= np.array([17, 12, 4]) - 1e-4).all()  # emission + whatever the bounce adds
                found += 1
    assert found > 4


def test_no_lights_renders_black():
    """mega_path.cpp:40-47: no lights -> "Rendering aborted", the image stays black."""
    text = cornell_box(16, 2).replace('light : Diffuse { emission : Constant { v { 17, 12, 4 } } }', "")
    sc = Scene.from_string(text)
    assert not sc.has_lighting
    film, _ = Oracle(sc).render(0, 2)
    assert (film == 0).all()


def test_sample_streams_are_independent_of_render_partition():
    """Counter-based sampler (independent.cpp:57-59): a pixel's result does not depend on which
    rectangle / thread / spp batch rendered it -> tile sharding is exact."""
    sc = Scene.from_string(cornell_box(resolution=24, spp=4))
    o = Oracle(sc)
    full, _ = o.render(0, 4, threads=3)
    parts = np.zeros_like(full)
    o.render(0, 2, rect=(0, 0, 24, 9), threads=1, film=parts)
    o.render(0, 2, rect=(0, 9, 24, 24), threads=2, film=parts)
    o.render(2, 4, rect=(0, 0, 11, 24), threads=1, film=parts)
    o.render(2, 4, rect=(11, 0, 24, 24), threads=1, film=parts)
    assert np.array_equal(full, parts)


def test_film_clamp_and_sample_count():
    text = cornell_box(resolution=16, spp=4).replace("resolution { 16, 16 }", "resolution { 16, 16 } clamp { 1 }")
    sc = Scene.from_string(text)
    o = Oracle(sc)
    film, _ = o.render(0, 4)
    assert film[..., :3].max() <= 4.0 + 1e-5  # each sample clamped to max component 1 (color.cpp:111-115)
    assert (film[..., 3] == 4).all()


def test_golden_cornell_fixture():
    """tests/golden/make_golden.py: oracle film of the Cornell box, fixed seed (regression pin)."""
    ref = np.load(os.path.join(GOLDEN, "cornell_32_8spp.npz"))
    sc = Scene.from_string(cornell_box(resolution=32, spp=8))
    film, counters = Oracle(sc).render(0, 8)
    assert int(ref["closest_rays"]) == counters["closest_rays"]
    assert np.allclose(film, ref["film"], rtol=1e-5, atol=1e-6)


def test_degenerate_geometry_keeps_a_shallow_bvh():
    """stacks of identical triangles and collinear (zero-area) ones: every split / every reinsertion place costs the same; the
    builder must keep a balanced tree (the device's traversal stack bounds the depth, lrhip_upload_scene rejects deeper trees)"""
    n = 3000
    pos, idx = [], []
    for i in range(n):
        b = len(pos) // 3
        pos += [0, 0, 0, 1, 0, 0, 0, 1, 0]
        idx += [b, b + 1, b + 2]
    for i in range(n):
        b = len(pos) // 3
        pos += [i * 1e-3, 2, 0, i * 1e-3 + 1e-3, 2, 0, i * 1e-3 + 2e-3, 2, 0]
        idx += [b, b + 1, b + 2]
    text = f"""
Shape s : InlineMesh {{ positions {{ {", ".join(map(str, pos))} }} indices {{ {", ".join(map(str, idx))} }} light : Diffuse {{ emission : Constant {{ v {{ 1 }} }} two_sided {{ true }} }} }}
Camera cam : Pinhole {{ spp {{ 1 }} film : Color {{ resolution {{ 8, 8 }} }} position {{ 0.3, 0.3, 3 }} look_at {{ 0.3, 0.3, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @s }} integrator : MegaPath {{ }} }}
"""
    sc = Scene.from_string(text)
    v = sc.view()
    nodes, depth, stack = v.accel.nodes, 0, [(0, 1)]
    leaves = 0
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        for k in range(4):
            c = nodes[i].child[k]
            if c == 0xFFFFFFFF:
                continue
            if c & 0x80000000:
                leaves += 1
            else:
                stack.append((c, d + 1))
    assert leaves == 2 * n and depth <= 20, depth  # log4(6000) = 6.3; the stack allows 34
    film, _ = Oracle(sc).render(0, 1)
    assert film[4, 4, 0] == 1.0


def test_swizzle_texture(tmp_path):
    """src/textures/swizzle.cpp: a swizzled constant folds to a constant (evaluate_static); a swizzled image permutes the
    base's channels per lookup — a Matte surface with Kd = image.bgr under a white sky shows the image's colours swapped"""
    from luisarender_amd.scene import save_image
    img = np.zeros((4, 4, 4), np.float32)
    img[..., 0], img[..., 1], img[..., 2], img[..., 3] = 0.8, 0.4, 0.1, 1.0
    save_image(str(tmp_path / "rgb.exr"), img)
    text = """
Texture base : Image { file { "rgb.exr" } encoding { "linear" } }
Texture swapped : Swizzle { base { @base } swizzle { "bgr" } }
Texture red_as_grey : Swizzle { base { @base } swizzle { 0 } }
Texture folded : Swizzle { base : Constant { v { 0.1, 0.2, 0.3 } } swizzle { "zy" } }
Surface s : Matte { Kd { @KD } }
Shape quad : InlineMesh { positions { -50,0,-50, 50,0,-50, 50,0,50, -50,0,50 } indices { 0,2,1, 0,3,2 } uvs { 0,0, 1,0, 1,1, 0,1 } surface { @s } }
Camera cam : Pinhole { fov { 30 } spp { 1 } film : Color { resolution { 8, 8 } } position { 0, 5, 0 } look_at { 0, 0, -3 } }
render { cameras { @cam } shapes { @quad } environment : Spherical { emission : Constant { v { 1 } } } integrator : MegaPath { depth { 2 } } }
"""
    def mean(kd):
        (tmp_path / "s.luisa").write_text(text.replace("@KD", "@" + kd))
        sc = Scene.load(str(tmp_path / "s.luisa"))
        film, _ = Oracle(sc).render(0, 64)
        return sc, (film[..., :3] / film[..., 3:4]).reshape(-1, 3).mean(axis=0)
    _, plain = mean("base")
    sc, swapped = mean("swapped")
    assert np.allclose(plain, [0.8, 0.4, 0.1], rtol=0.03) and np.allclose(swapped, [0.1, 0.4, 0.8], rtol=0.03)
    _, grey = mean("red_as_grey")
    assert np.allclose(grey, [0.8, 0.8, 0.8], rtol=0.03)  # one channel: extended to rgb (texture.cpp:14-18)
    sc, folded = mean("folded")
    v = sc.view()
    t = [v.textures[i] for i in range(v.texture_count) if v.textures[i].kind == 0 and v.textures[i].channels == 2]
    assert any(np.allclose(x.v[:2], [0.3, 0.2]) for x in t)  # evaluate_static: (z, y) of the constant
    assert np.allclose(folded, [0.3, 0.2, 1.0], rtol=0.03)  # two channels extend to (x, y, 1)


def test_bvh_build_is_deterministic(tmp_path):
    """the sweep build runs its big subtrees as concurrent tasks; the tree that comes out (topology, node order, triangle
    order) must not depend on how they were scheduled: every rank of a multi-GPU render builds its own copy"""
    import ctypes as C
    from luisarender_amd.scenes import generate_room_scene
    path = generate_room_scene(str(tmp_path), target_triangles=60_000, resolution=(32, 32), spp=1)
    blobs = []
    for _ in range(3):
        sc = Scene.load(path)
        v = sc.view()
        nodes = C.string_at(C.addressof(v.accel.nodes.contents), v.accel.node_count * 128)
        tris = C.string_at(C.addressof(v.accel.triangles.contents), v.accel.triangle_count * 48)
        blobs.append((v.accel.node_count, nodes, tris))
    assert blobs[0] == blobs[1] == blobs[2]


def test_bvh4_is_a_valid_tree_over_every_triangle(tmp_path):
    """structural check of what lrhip_upload_scene receives: every baked triangle sits in exactly one leaf, every child box
    encloses its child (the triangle, or the child node's boxes), child indices are larger than their parent's (refit and the
    collapse rely on it), depth within the traversal stack's bound"""
    from luisarender_amd.scenes import generate_room_scene
    sc = Scene.load(generate_room_scene(str(tmp_path), target_triangles=30_000, resolution=(32, 32), spp=1))
    v = sc.view()
    n_nodes, n_tris = v.accel.node_count, v.accel.triangle_count
    import ctypes as C
    raw = np.frombuffer(C.string_at(C.addressof(v.accel.nodes.contents), n_nodes * 128), np.uint8).reshape(n_nodes, 128)
    boxes = raw[:, :96].copy().view(np.float32).reshape(n_nodes, 6, 4)   # lo_x lo_y lo_z hi_x hi_y hi_z, 4 children each
    child = raw[:, 96:112].copy().view(np.uint32).reshape(n_nodes, 4)
    tri = np.frombuffer(C.string_at(C.addressof(v.accel.triangles.contents), n_tris * 48), np.float32).reshape(n_tris, 12)
    p0, e1, e2 = tri[:, 0:3], tri[:, 4:7], tri[:, 8:11]
    tlo = np.minimum(np.minimum(p0, p0 + e1), p0 + e2)
    thi = np.maximum(np.maximum(p0, p0 + e1), p0 + e2)
    valid = child != 0xFFFFFFFF
    leaf = valid & ((child & 0x80000000) != 0)
    inner = valid & ~leaf
    # leaves: each triangle exactly once, inside its box
    refs = (child[leaf] & 0x7FFFFFF).astype(np.int64)
    assert np.array_equal(np.sort(refs), np.arange(n_tris))
    node_of, slot_of = np.nonzero(leaf)
    lo = boxes[node_of, 0:3, slot_of]
    hi = boxes[node_of, 3:6, slot_of]
    assert (tlo[refs] >= lo - 1e-6).all() and (thi[refs] <= hi + 1e-6).all()
    # inner children: larger index than the parent, box encloses the child's boxes, every node except the root referenced once
    node_of, slot_of = np.nonzero(inner)
    kids = child[inner].astype(np.int64)
    assert (kids > node_of).all() and np.array_equal(np.sort(kids), np.arange(1, n_nodes))
    big = 3e38
    klo = np.where(valid[kids][:, None, :], boxes[kids, 0:3, :], big).min(axis=2)
    khi = np.where(valid[kids][:, None, :], boxes[kids, 3:6, :], -big).max(axis=2)
    assert (klo >= boxes[node_of, 0:3, slot_of] - 1e-6).all() and (khi <= boxes[node_of, 3:6, slot_of] + 1e-6).all()
    depth = np.zeros(n_nodes, np.int64)
    depth[0] = 1
    for parent, kid in sorted(zip(node_of.tolist(), kids.tolist())):
        depth[kid] = depth[parent] + 1
    assert depth.max() * 3 <= 16 + 88  # lrhip_upload_scene's bound: kStackLds + kSpillEntries
