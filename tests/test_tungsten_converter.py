"""SURVEY §8 f4, asset pipeline: tools/tungsten2luisa.py (the repository's own Tungsten -> LuisaRender converter; the
reference ships tools/tungsten2luisa.py for the same job).  A hand-written Tungsten scene covering every bsdf / primitive
kind the converter maps goes through it, is loaded by the host library and rendered by the oracle."""
import importlib.util
import json
import os

import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("tungsten2luisa", os.path.join(ROOT, "tools", "tungsten2luisa.py"))
t2l = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t2l)

CUBE_OBJ = """v -1 -1 -1\nv 1 -1 -1\nv 1 1 -1\nv -1 1 -1\nv -1 -1 1\nv 1 -1 1\nv 1 1 1\nv -1 1 1
f 1 3 2\nf 1 4 3\nf 5 6 7\nf 5 7 8\nf 1 2 6\nf 1 6 5\nf 4 7 3\nf 4 8 7\nf 1 8 4\nf 1 5 8\nf 2 3 7\nf 2 7 6\n"""

TUNGSTEN = {
    "bsdfs": [
        {"name": "floor", "type": "lambert", "albedo": {"type": "checker", "on_color": [0.7, 0.7, 0.7], "off_color": [0.2, 0.2, 0.2], "res_u": 4, "res_v": 4}},
        {"name": "red", "type": "oren_nayar", "albedo": [0.6, 0.1, 0.1]},
        {"name": "shiny", "type": "rough_plastic", "albedo": 0.5, "ior": 1.5, "roughness": 0.04},
        {"name": "gold", "type": "rough_conductor", "material": "Au", "albedo": 1.0, "roughness": 0.09},
        {"name": "chrome", "type": "conductor", "eta": 2.0, "k": 3.0, "albedo": [0.9, 0.9, 0.9]},
        {"name": "window", "type": "dielectric", "albedo": 1.0, "ior": 1.5},
        {"name": "looking_glass", "type": "mirror", "albedo": 0.9},
        {"name": "veil", "type": "transparency", "alpha": 0.5, "base": {"type": "lambert", "albedo": [0.2, 0.6, 0.3]}},
        {"name": "strange", "type": "phong", "albedo": 0.5},
    ],
    "primitives": [
        {"type": "quad", "bsdf": "floor", "transform": {"scale": [8, 1, 8]}},
        {"type": "quad", "bsdf": "red", "transform": {"position": [0, 2, -4], "rotation": [90, 0, 0], "scale": [8, 1, 4]}},
        {"type": "cube", "bsdf": "shiny", "transform": {"position": [-1.5, 0.5, 0], "rotation": [0, 30, 0]}},
        {"type": "cube", "bsdf": "gold", "transform": {"position": [0, 0.5, -1]}},
        {"type": "cube", "bsdf": "chrome", "transform": {"position": [1.5, 0.5, 0], "scale": 0.8}},
        {"type": "cube", "bsdf": "window", "transform": {"position": [0, 0.4, 1.5], "scale": [1.5, 0.8, 0.1]}},
        {"type": "quad", "bsdf": "looking_glass", "transform": {"position": [-3.9, 1.5, 0], "rotation": [0, 0, -90], "scale": [3, 1, 3]}},
        {"type": "quad", "bsdf": "veil", "transform": {"position": [2.5, 1, 1], "rotation": [90, 0, 0], "scale": 2}},
        {"type": "mesh", "file": "models/block.wo3", "bsdf": {"type": "lambert", "albedo": 0.4}, "transform": {"position": [3, 0.25, -2], "scale": 0.25}},
        {"type": "quad", "bsdf": {"type": "null"}, "power": 400, "transform": {"position": [0, 3.9, 0], "rotation": [180, 0, 0], "scale": [2, 1, 2]}},
        {"type": "infinite_sphere", "emission": [0.3, 0.35, 0.4]},
    ],
    "camera": {"type": "pinhole", "resolution": [64, 36], "fov": 60,
               "transform": {"position": [0, 2, 7], "look_at": [0, 1, 0], "up": [0, 1, 0]}},
}


def _convert(tmp_path, scene=TUNGSTEN, spp=4, sampler="Independent"):
    os.makedirs(tmp_path / "models", exist_ok=True)
    for name in ("cube", "block"):
        (tmp_path / "models" / f"{name}.obj").write_text(CUBE_OBJ)
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    text = t2l.convert(scene, spp, sampler)
    (tmp_path / "scene.luisa").write_text(text)
    return text


def test_converted_scene_loads_and_renders(tmp_path):
    text = _convert(tmp_path)
    for expected in ("mat_floor : Matte", "Checkerboard", "mat_shiny : Plastic", 'eta { "Au" }', "360, 2.0, 3.0, 830, 2.0, 3.0",
                     "mat_window : Glass", "mat_looking_glass : Mirror", "alpha : Constant { v { 0.5 } }", "mat_strange : Matte",
                     "mat_shape_8 : Matte", "@mat_Null", "Env env : Spherical", "rotate { 0, 1, 0, -90 }", "sampler : Independent"):
        assert expected in text, expected
    sc = Scene.load(str(tmp_path / "scene.luisa"))
    v = sc.view()
    assert (v.camera.width, v.camera.height, v.camera.spp) == (64, 36, 4)
    # horizontal 60 degrees -> vertical: tan(v/2) = tan(30 deg) * 36 / 64
    assert v.camera.tan_half_fov == pytest.approx(np.tan(np.radians(30)) * 36 / 64, rel=1e-5)
    assert v.instance_count == 10 and v.light_count == 1 and v.environment.kind != 0 and v.any_non_opaque == 1
    # roughness: alpha 0.04 -> sqrt -> LuisaRender squares it again (roughness_to_alpha)
    plastic = [s for s in (v.surfaces[i] for i in range(v.surface_count)) if s.kind == 3]
    assert plastic, [v.surfaces[i].kind for i in range(v.surface_count)]
    film, counters = Oracle(sc).render(0, 4)
    img = film[..., :3] / np.maximum(film[..., 3:4], 1)
    assert np.isfinite(img).all() and img.mean() > 0.02 and counters["shadow_rays"] > 0
    # the power-400 quad light: emission = power / (sx * sz * pi) = 400 / (4 pi)
    assert f"{400 / (4 * np.pi)!r}"[:8] in text


def test_quad_and_rotation_conventions():
    """Tungsten's quad spans [-0.5, 0.5]^2 in its local xz plane facing +y; rotations compose y, x, z"""
    out = []
    t2l.primitive(0, {"type": "quad", "bsdf": "m", "transform": {"position": [1, 2, 3], "scale": [4, 1, 6]}}, out)
    rows = [list(map(float, line.strip().rstrip(",").split(","))) for line in out[0].split("m {")[1].split("}")[0].strip().splitlines()]
    m = np.array(rows)
    corners = np.array([[1, 1, 0, 1], [-1, 1, 0, 1], [-1, -1, 0, 1], [1, -1, 0, 1]], np.float64) @ m.T
    assert np.allclose(corners[:, 1], 2) and np.allclose(sorted(set(np.round(corners[:, 0], 6))), [-1, 3]) and np.allclose(sorted(set(np.round(corners[:, 2], 6))), [0, 6])
    n = np.cross(corners[1, :3] - corners[0, :3], corners[2, :3] - corners[0, :3])
    assert n[1] > 0  # faces +y
    r = t2l.rotation_yxz(np.radians([0.0, 90.0, 0.0]))
    assert np.allclose(r @ [0, 0, 1], [-1, 0, 0], atol=1e-12)  # (Tungsten's rotY sends +z to -x)
    r = t2l.rotation_yxz(np.radians([90.0, 0.0, 0.0]))
    assert np.allclose(r @ [0, 1, 0], [0, 0, 1], atol=1e-12)


def test_reference_default_sampler_is_rejected_by_name(tmp_path):
    """the reference converter writes `sampler : PMJ02BN {}`; its tables are not in the snapshot, the loader says so"""
    from luisarender_amd.scene import HostError
    _convert(tmp_path, sampler="PMJ02BN")
    with pytest.raises(HostError, match="PMJ02BN"):
        Scene.load(str(tmp_path / "scene.luisa"))
