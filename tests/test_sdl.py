"""Scene description layer: grammar and semantics of the reference's src/sdl (scene_parser.cpp:71-451,
scene_parser_json.cpp, scene_node_desc.h getters) as re-created in csrc/host/sdl.cpp."""
import json
import os

import pytest

from luisarender_amd import HostError, Scene
from luisarender_amd.scenes import cornell_box

MINI = """
// comment line
define RES 32
Surface white : Matte { Kd : Constant { v { 0.5, 0.5, 0.5 } } }
Shape quad : InlineMesh {
  positions { -1, 0, -1,  1, 0, -1,  1, 0, 1,  -1, 0, 1 }
  indices { 0, 1, 2, 0, 2, 3 }
  surface { @white }
  light : Diffuse { emission : Constant { v { 1, 1, 1 } } scale { 2 } }
}
Camera cam : Pinhole {
  fov { 40 } spp { #SPP }
  film : Color { resolution { #RES } }
  position { 0, 3, 0 } look_at { 0, 0, 0 } up { 0, 0, -1 }
}
render { cameras { @cam } shapes { @quad } integrator : MegaPath { depth { 3 } } }
"""


def test_text_grammar_macros_and_defaults():
    sc = Scene.from_string(MINI, macros={"SPP": 7}, build_accel=False)
    v = sc.view()
    assert (v.camera.width, v.camera.height, v.camera.spp) == (32, 32, 7)
    assert v.integrator.max_depth == 3 and v.integrator.rr_depth == 0
    assert abs(v.integrator.rr_threshold - 0.95) < 1e-7
    assert v.sampler.seed == 19980810 and v.sampler.kind == 0  # default Independent sampler
    assert v.filter.radius == 0.5  # default Box filter
    assert v.instance_count == 1 and v.triangle_count == 2 and v.light_count == 1
    assert abs(v.lights[0].scale - 2.0) < 1e-7
    # legacy position/look_at synthesises a View transform (camera.cpp:31-50)
    assert abs(v.camera.camera_to_world[13] - 3.0) < 1e-6


def test_cli_macro_overrides_local_define():
    sc = Scene.from_string(MINI, macros={"SPP": 1, "RES": 16}, build_accel=False)
    assert sc.resolution() == (16, 16)


def test_undefined_macro_is_an_error():
    with pytest.raises(HostError, match="Undefined macro"):
        Scene.from_string(MINI, build_accel=False)


def test_json_form_matches_text_form(tmp_path):
    doc = {
        "white": {"type": "Surface", "impl": "Matte", "prop": {"Kd": {"impl": "Constant", "prop": {"v": [0.5, 0.5, 0.5]}}}},
        "quad": {"type": "Shape", "impl": "InlineMesh", "prop": {
            "positions": [-1, 0, -1, 1, 0, -1, 1, 0, 1, -1, 0, 1], "indices": [0, 1, 2, 0, 2, 3], "surface": "@white",
            "light": {"impl": "Diffuse", "prop": {"emission": {"impl": "Constant", "prop": {"v": [1, 1, 1]}}, "scale": 2}}}},
        "cam": {"type": "Camera", "impl": "Pinhole", "prop": {
            "fov": 40, "spp": 7, "film": {"impl": "Color", "prop": {"resolution": 32}},
            "position": [0, 3, 0], "look_at": [0, 0, 0], "up": [0, 0, -1]}},
        "render": {"cameras": ["@cam"], "shapes": ["@quad"], "integrator": {"impl": "MegaPath", "prop": {"depth": 3}}},
    }
    a = Scene.from_string(json.dumps(doc), json=True, build_accel=False).view()
    b = Scene.from_string(MINI, macros={"SPP": 7}, build_accel=False).view()
    assert bytes(a.camera) == bytes(b.camera) and bytes(a.filter) == bytes(b.filter)
    assert a.triangle_count == b.triangle_count and bytes(a.instances[0]) == bytes(b.instances[0])


def test_import_base_inheritance_and_group_override(tmp_path):
    (tmp_path / "mats.luisa").write_text(
        "Surface base_mat : Matte { Kd : Constant { v { 0.2, 0.3, 0.4 } } }\n"
        "Surface derived : Matte (@base_mat) { sigma : Constant { v { 0.5 } } }\n"
        "Surface red : Matte { Kd : Constant { v { 1, 0, 0 } } }\n")
    (tmp_path / "main.luisa").write_text("""
import "mats.luisa"
Shape tri : InlineMesh { positions { 0,0,0, 1,0,0, 0,1,0 } indices { 0,1,2 } surface { @derived } }
Shape lamp : InlineMesh { positions { 0,2,0, 1,2,0, 0,2,1 } indices { 0,1,2 } light : Diffuse { emission : Constant { v { 1 } } } }
Shape grp : Group { shapes { @tri } surface { @red } transform : SRT { translate { 0, 0, 5 } } }
Camera cam : Pinhole { film : Color { resolution { 8, 4 } } spp { 1 } }
render { cameras { @cam } shapes { @tri, @grp, @lamp } integrator : MegaPath { } }
""")
    v = Scene.load(str(tmp_path / "main.luisa"), build_accel=False).view()
    assert v.instance_count == 3 and v.mesh_count == 2  # `tri` is shared by both instances
    tag0 = (v.instances[0].handle.y >> 12) & 4095
    tag1 = (v.instances[1].handle.y >> 12) & 4095
    assert tag0 != tag1  # an ancestor's surface overrides the child's own (geometry.cpp:36-39)
    s0 = v.surfaces[tag0]
    kd = v.textures[s0.tex[0]]  # Kd inherited from @base_mat through `(@base_mat)`
    assert [round(kd.v[i], 3) for i in range(3)] == [0.2, 0.3, 0.4] and s0.tex[1] >= 0
    assert abs(v.instances[1].object_to_world[14] - 5.0) < 1e-6  # group transform applied
    assert (v.camera.width, v.camera.height) == (8, 4)
    assert os.path.basename(Scene.load(str(tmp_path / "main.luisa"), build_accel=False).camera_file()) == "render.exr"


def test_errors_are_reported_not_aborted():
    with pytest.raises(HostError, match="out of scope"):
        Scene.from_string(cornell_box(8, 1).replace("MegaPath", "WavePath"), build_accel=False)
    with pytest.raises(HostError, match="Redefinition"):
        Scene.from_string("Surface a : Matte { }\nSurface a : Matte { }\n" + MINI, macros={"SPP": 1}, build_accel=False)
    with pytest.raises(HostError):
        Scene.from_string("render { cameras { } }", build_accel=False)


def test_cornell_box_layout():
    v = Scene.from_string(cornell_box(64, 4)).view()
    assert v.instance_count == 16 and v.triangle_count == 32 and v.light_instance_count == 1
    assert v.surface_count == 3 and v.accel.triangle_count == 32
    assert v.integrator.light_count == 1 and v.integrator.env_prob == 0.0


def test_image_texture_default_encoding_follows_the_extension_rule_of_the_reference(tmp_path):
    """src/textures/image.cpp:85-90: only `.exr` and `.hdr` default to linear, every other extension -- `.pfm` included --
    to sRGB.  (Round 3: the host had `.pfm` on the linear side; found by running the reduced C3 generator through the reference's
    own code, tests/test_oracle_vs_ref.py::test_li_baseline_stand_ins.)"""
    import numpy as np
    from luisarender_amd.scenes.configs import write_pfm
    from luisarender_amd.scene import save_image
    img = np.full((4, 4, 4), 0.5, np.float32)
    write_pfm(str(tmp_path / "a.pfm"), img)
    save_image(str(tmp_path / "a.exr"), img)
    save_image(str(tmp_path / "a.hdr"), img)
    for ext, srgb in ((".pfm", True), (".exr", False), (".hdr", False)):
        text = MINI.replace("#SPP", "1").replace("Kd : Constant { v { 0.5, 0.5, 0.5 } }", f'Kd : Image {{ file {{ "a{ext}" }} }}')
        view = Scene.from_string(text, virtual_path=str(tmp_path / "scene.luisa")).view(0)
        encodings = [view.textures[i].encoding for i in range(view.texture_count) if view.textures[i].kind == 1]
        assert encodings == [1 if srgb else 0], (ext, encodings)
