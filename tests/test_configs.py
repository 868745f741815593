"""CPU checks of the BASELINE C3..C5 stand-in generators (luisarender_amd/scenes/configs.py): the scenes parse and
flatten through liblrhost, hold the features the configs name, and the oracle renders them to finite, lit images."""
import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle
from luisarender_amd.scenes import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene

SMALL = dict(target_triangles=60_000, resolution=(64, 36), spp=2)


def _surface_kinds(view):
    return {int(view.surfaces[i].kind) for i in range(view.surface_count)}


@pytest.mark.parametrize("gen, kw", [
    (generate_bedroom_scene, dict(env_resolution=(256, 128))),
    (generate_camera_scene, dict(env_resolution=(256, 128), texture_size=128)),
    (generate_kitchen_scene, {}),
])
def test_config_stand_ins_render_on_the_oracle(tmp_path, gen, kw):
    sc = Scene.load(gen(str(tmp_path), **SMALL, **kw))
    v = sc.view()
    assert v.accel.triangle_count >= 60_000 and (v.camera.width, v.camera.height) == (64, 36)
    film, counters = Oracle(sc).render(0, 2, threads=4)
    assert np.isfinite(film).all() and (film[..., 3] == 2).all()
    assert film[..., :3].mean() > 1e-3
    assert counters["paths"] == 64 * 36 * 2 and counters["closest_rays"] > counters["paths"]


def test_bedroom_has_an_image_environment_and_camera_has_textures(tmp_path):
    sc = Scene.load(generate_bedroom_scene(str(tmp_path / "c3"), env_resolution=(256, 128), **SMALL))
    v = sc.view()
    assert v.environment.kind != 0 and v.light_instance_count == 0  # lit through the windows only
    sc4 = Scene.load(generate_camera_scene(str(tmp_path / "c4"), env_resolution=(256, 128), texture_size=128, **SMALL))
    v4 = sc4.view()
    assert v4.texel_count >= 8 * 128 * 128 and v4.camera.kind != 0  # 8 images resident, thin lens


def test_kitchen_holds_the_full_closure_set(tmp_path):
    sc = Scene.load(generate_kitchen_scene(str(tmp_path), resolution=(64, 36), spp=2))  # the full ~600 k triangle population
    v = sc.view()
    assert _surface_kinds(v) >= set(range(1, 9))  # LR_SURFACE_MATTE .. LR_SURFACE_LAYERED (include/lr_scene.h)
    assert v.any_non_opaque != 0                  # the alpha-tested "lace" fixtures
