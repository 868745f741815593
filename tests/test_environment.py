"""Row a11 / f1: image-based Spherical environments (importance tables built by lrhost, restating
src/environments/spherical.cpp:144-235) and Directional environments (directional.cpp), on the CPU oracle.
The reference ships no environment fixtures, so the pins are closed forms: table normalisation, sampled
frequencies vs the pdf table, and a diffuse plane whose radiance is the cosine-weighted integral of the map."""
import ctypes as C

import numpy as np
import pytest

from luisarender_amd import Scene, _ffi
from oracle.check import Oracle
from luisarender_amd.scene import save_image

W, H = 2048, 1024  # Spherical::sample_map_size


def sky_image(w=256, h=128):
    """smooth sky gradient + a bright 'sun' blob + a dim coloured ground, float RGBA"""
    v, u = np.meshgrid((np.arange(h) + .5) / h, (np.arange(w) + .5) / w, indexing="ij")
    img = np.zeros((h, w, 4), np.float32)
    up = np.clip(1 - 2 * v, 0, 1)
    img[..., 0] = 0.3 + 0.5 * up
    img[..., 1] = 0.4 + 0.6 * up
    img[..., 2] = 0.6 + 1.0 * up
    img[v > 0.5] = (0.15, 0.12, 0.08, 0)
    sun = np.exp(-(((u - 0.3) / 0.02) ** 2 + ((v - 0.25) / 0.02) ** 2))
    img[..., :3] += 60.0 * sun[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
    img[..., 3] = 1
    return img


PLANE = """
Surface s : Matte {{ Kd : Constant {{ v {{ 0.6, 0.4, 0.2 }} }} }}
Shape quad : InlineMesh {{ positions {{ -50,0,-50, 50,0,-50, 50,0,50, -50,0,50 }} indices {{ 0,2,1, 0,3,2 }} surface {{ @s }} }}
Camera cam : Pinhole {{ fov {{ 30 }} spp {{ 1 }} film : Color {{ resolution {{ 16, 16 }} clamp {{ 100000 }} }}
  position {{ 0, 5, 0 }} look_at {{ 0, 0, -3 }} }}
render {{ cameras {{ @cam }} shapes {{ @quad }}
  environment : {env}
  integrator : MegaPath {{ depth {{ 4 }} }} }}
"""


@pytest.fixture(scope="module")
def sky(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("env") / "sky.exr")
    img = sky_image()
    save_image(path, img)
    return path, img


def _env_scene(path, extra=""):
    env = f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} {extra} }}'
    return Scene.from_string(PLANE.format(env=env), build_accel=False)


def _tables(view):
    env = view.environment
    assert (env.map_width, env.map_height) == (W, H)
    pdf = np.ctypeslib.as_array(C.cast(env.pdf, C.POINTER(C.c_float)), shape=(H, W))
    alias = np.ctypeslib.as_array(C.cast(env.alias, C.POINTER(C.c_uint32)), shape=(H + H * W, 2))
    prob = alias[:, 0].copy().view(np.float32)
    return pdf, prob, alias[:, 1]


def test_importance_tables_are_normalised_and_follow_the_map(sky):
    path, img = sky
    sc = _env_scene(path)
    pdf, prob, alias = _tables(sc.view())
    assert abs(pdf.mean() - 1.0) < 1e-3  # pdf = p_row * p_col * pixel_count (spherical.cpp:216-222)
    assert (pdf >= 0).all() and (prob >= 0).all() and (prob <= 1.0 + 1e-5).all()
    assert (alias[:H] < H).all() and (alias[H:] < W).all()
    # MIS compensation (spherical.cpp:187-192) removes the average: the dim ground gets pdf 0, the sun the maximum
    assert pdf[H * 3 // 4].max() == 0.0
    sy, sx = np.unravel_index(pdf.argmax(), pdf.shape)
    assert abs(sx / W - 0.3) < 0.02 and abs(sy / H - 0.25) < 0.02
    # without compensation the table is the filtered luminance * sin(theta)
    raw, _, _ = _tables(_env_scene(path, "compensate_mis { false }").view())
    lum = (img[..., :3] * np.array([0.212671, 0.715160, 0.072169])).sum(-1) * np.sin(np.pi * (np.arange(128) + .5) / 128)[:, None]
    lum = lum / lum.mean()
    coarse = raw.reshape(128, 8, 256, 8).mean(axis=(1, 3))
    assert abs(raw.mean() - 1.0) < 1e-3
    assert np.abs(coarse - lum).sum() / lum.sum() < 0.03  # up to the radius-1 Gaussian filter and bilinear taps


def test_alias_sampling_reproduces_the_pdf_table(sky):
    """sample_alias_table over the marginal + conditional tables (spherical.cpp:123-133) draws texel (ix, iy) with
    probability pdf / pixel_count: block frequencies of 2^22 random draws."""
    sc = _env_scene(sky[0])
    pdf, prob, alias = _tables(sc.view())
    u = np.random.default_rng(5).random((1 << 22, 2))
    def pick(p, a, count, uu):
        x = (uu * count).astype(np.float32)
        i = np.minimum(x.astype(np.int64), count - 1)
        r = x - np.floor(x)
        return np.where(r < p[i], i, a[i])
    iy = pick(prob[:H], alias[:H], H, u[:, 1].astype(np.float32))
    base = H + iy * W
    x = (u[:, 0].astype(np.float32) * W)
    i = np.minimum(x.astype(np.int64), W - 1)
    r = x - np.floor(x)
    ix = np.where(r < prob[base + i], i, alias[base + i])
    hist = np.zeros((16, 32))
    np.add.at(hist, (iy * 16 // H, ix * 32 // W), 1.0)
    hist /= hist.sum()
    want = pdf.reshape(16, H // 16, 32, W // 32).sum(axis=(1, 3))
    want /= want.sum()
    assert np.abs(hist - want).sum() < 0.03


def _plane_radiance(img):
    """rho / pi * int L cos over the upper hemisphere (+y) of the lat-long map, per channel"""
    h, w = img.shape[:2]
    theta = np.pi * (np.arange(h) + .5) / h
    d_omega = (np.pi / h) * (2 * np.pi / w) * np.sin(theta)
    cos = np.clip(np.cos(theta), 0, None)
    irradiance = (img[..., :3] * (d_omega * cos)[:, None, None]).sum(axis=(0, 1))
    return np.array([0.6, 0.4, 0.2]) / np.pi * irradiance


@pytest.mark.parametrize("extra", ["", "compensate_mis { false }", "scale { 0.5 } transform : SRT { rotate { 0, 1, 0, 75 } }"])
def test_diffuse_plane_under_image_environment(sky, extra):
    """Closed form: a convex diffuse receiver sees rho / pi * irradiance; NEE through the alias tables and BSDF-sampled
    misses through the pdf table must agree (a uv <-> direction mismatch between sample and evaluate would bias MIS).
    Rotating the map about the plane normal does not change the answer."""
    path, img = sky
    sc = Scene.from_string(PLANE.format(env=f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} {extra} }}'))
    o = Oracle(sc)
    film, _ = o.render(0, 512)
    got = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
    expect = _plane_radiance(img) * (0.5 if "scale" in extra else 1.0)
    assert np.allclose(got, expect, rtol=0.03), (got, expect)


def test_directional_environment_closed_form():
    """directional.cpp: a cone of half-angle a/2 around `direction`, normalised so that scale = irradiance / (4 pi ... ):
    L_cone = 2 scale / (1 - cos) -> irradiance on a plane facing it = L_cone * pi * sin^2(a/2) -> rho / pi * E."""
    for angle, visible in ((10.0, "true"), (40.0, "false")):
        env = f"Directional {{ emission : Constant {{ v {{ 1, 2, 3 }} }} scale {{ 0.5 }} angle {{ {angle} }} direction {{ 0, 1, 0 }} visible {{ {visible} }} }}"
        sc = Scene.from_string(PLANE.format(env=env))
        view = sc.view()
        assert view.environment.kind == 2 and view.environment.visible == (1 if visible == "true" else 0)
        o = Oracle(sc)
        film, _ = o.render(0, 512)
        got = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
        half = np.radians(angle / 2)
        l_cone = np.array([1.0, 2.0, 3.0]) * 2 * 0.5 / (1 - np.cos(half))
        expect = np.array([0.6, 0.4, 0.2]) / np.pi * l_cone * np.pi * np.sin(half) ** 2
        if visible == "false":
            # BSDF-sampled rays that escape see nothing (evaluate() returns zero, directional.cpp:82) while NEE keeps its
            # MIS weight: the estimate loses the BSDF-sampling share — only an upper bound holds
            assert (got < expect * 1.03).all() and (got > 0.3 * expect).all()
        else:
            assert np.allclose(got, expect, rtol=0.03), (got, expect)


def test_combined_environment(sky):
    """combined.cpp: L = a * scale_a + b * scale_b under the Combined node's transform; a Combined with one live child is
    flattened by the host into that child (composed transform, scaled radiance)."""
    path, img = sky
    sun = "Directional { emission : Constant { v { 1, 2, 3 } } scale { 0.5 } angle { 10 } direction { 0, 1, 0 } }"
    dome = f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} }}'
    both = f"Combined {{ a : {dome} b : {sun} scale_a {{ 0.5 }} scale_b {{ 2 }} transform : SRT {{ rotate {{ 0, 1, 0, 40 }} }} }}"
    sc = Scene.from_string(PLANE.format(env=both))
    view = sc.view()
    assert view.environment.kind == 3 and view.environment_child_count == 2
    o = Oracle(sc)
    film, _ = o.render(0, 1024)
    got = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
    half = np.radians(5.0)
    sun_part = np.array([0.6, 0.4, 0.2]) / np.pi * (np.array([1.0, 2.0, 3.0]) * 2 * 0.5 / (1 - np.cos(half))) * np.pi * np.sin(half) ** 2
    expect = 0.5 * _plane_radiance(img) + 2.0 * sun_part
    assert np.allclose(got, expect, rtol=0.03), (got, expect)
    # one black child -> flattened into the other
    lone = f"Combined {{ a : {dome} b : Spherical {{ emission : Constant {{ v {{ 0 }} }} }} scale_a {{ 0.25 }} transform : SRT {{ rotate {{ 0, 1, 0, 40 }} }} }}"
    sc = Scene.from_string(PLANE.format(env=lone))
    view = sc.view()
    assert view.environment.kind == 1 and view.environment_child_count == 0 and view.environment.map_width == W
    o = Oracle(sc)
    film, _ = o.render(0, 512)
    got = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
    assert np.allclose(got, 0.25 * _plane_radiance(img), rtol=0.03)


def test_nested_combined_environments(sky):
    """combined.cpp composes freely: a Combined node may be the child of a Combined node.  The host lays the tree out children before
    parents (lr_scene.h), flattens nodes with one live child at any level, and refuses more than LR_ENV_MAX_COMBINED_DEPTH levels."""
    path, img = sky
    sun = "Directional { emission : Constant { v { 1, 2, 3 } } scale { 0.5 } angle { 10 } direction { 0, 1, 0 } }"
    dome = f'Spherical {{ emission : Image {{ file {{ "{path}" }} }} }}'
    black = "Spherical { emission : Constant { v { 0 } } }"
    inner = f"Combined {{ a : {sun} b : {sun} scale_a {{ 0.5 }} scale_b {{ 1.5 }} }}"
    both = f"Combined {{ a : {dome} b : {inner} scale_a {{ 0.5 }} scale_b {{ 1 }} transform : SRT {{ rotate {{ 0, 1, 0, 40 }} }} }}"
    sc = Scene.from_string(PLANE.format(env=both))
    view = sc.view()
    root, kids = view.environment, view.environment_children
    assert root.kind == 3 and view.environment_child_count == 4
    kids = C.cast(kids, C.POINTER(type(root)))
    assert [kids[i].kind for i in range(4)] == [2, 2, 1, 3]            # the inner node's suns, then the root's dome and the inner node
    assert list(root.child) == [2, 3] and list(kids[3].child) == [0, 1]  # children precede their parents
    assert list(kids[3].child_scale) == [0.5, 1.5] and kids[2].map_width == W and kids[2].alias and not kids[0].alias
    o = Oracle(sc)
    film, _ = o.render(0, 1024)
    got = o.convert(film)[..., :3].reshape(-1, 3).mean(axis=0)
    half = np.radians(5.0)
    sun_part = np.array([0.6, 0.4, 0.2]) / np.pi * (np.array([1.0, 2.0, 3.0]) * 2 * 0.5 / (1 - np.cos(half))) * np.pi * np.sin(half) ** 2
    assert np.allclose(got, 0.5 * _plane_radiance(img) + 2.0 * sun_part, rtol=0.03)  # the inner node is worth (0.5 + 1.5) suns
    # an inner node with one live child is that child (scaled); a root whose only live child is a Combined node is that node (scaled)
    lone_inner = f"Combined {{ a : {black} b : {sun} scale_b {{ 3 }} }}"
    view = Scene.from_string(PLANE.format(env=f"Combined {{ a : {dome} b : {lone_inner} scale_b {{ 2 }} }}")).view()
    kids = C.cast(view.environment_children, C.POINTER(type(root)))
    assert view.environment_child_count == 2 and kids[1].kind == 2 and np.isclose(kids[1].scale, 3 * (2 * 0.5 / (1 - np.cos(half))))
    view = Scene.from_string(PLANE.format(env=f"Combined {{ a : {black} b : {inner} scale_b {{ 2 }} }}")).view()
    assert view.environment.kind == 3 and view.environment_child_count == 2 and list(view.environment.child_scale) == [1.0, 3.0]
    deep = sun
    for _ in range(5):
        deep = f"Combined {{ a : {sun} b : {deep} }}"
    with pytest.raises(Exception, match="nested more than 4 deep"):
        Scene.from_string(PLANE.format(env=deep))
    four = sun
    for _ in range(4):
        four = f"Combined {{ a : {sun} b : {four} }}"
    assert Scene.from_string(PLANE.format(env=four)).view().environment_child_count == 8
