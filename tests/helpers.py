import ctypes as C

import numpy as np

from luisarender_amd import Scene, _ffi
from oracle.check import oracle_lib

MATERIALS = {
    "matte": "Surface m : Matte { Kd : Constant { v { 0.6, 0.5, 0.4 } } }",
    "oren": "Surface m : Matte { Kd : Constant { v { 0.6, 0.5, 0.4 } } sigma : Constant { v { 0.4 } } }",
    "mirror": "Surface m : Mirror { color : Constant { v { 0.9, 0.8, 0.7 } } roughness : Constant { v { 0.3 } } }",
    "glass": "Surface m : Glass { Kr : Constant { v { 0.9, 0.9, 0.9 } } Kt : Constant { v { 0.95, 0.9, 0.85 } } roughness : Constant { v { 0.25 } } eta : Constant { v { 1.5 } } }",
    "plastic": "Surface m : Plastic { Kd : Constant { v { 0.5, 0.3, 0.2 } } roughness : Constant { v { 0.2 } } sigma_a : Constant { v { 0.1, 0.2, 0.3 } } eta : Constant { v { 1.5 } } thickness : Constant { v { 0.5 } } }",
    "disney": "Surface m : Disney { color : Constant { v { 0.7, 0.4, 0.3 } } metallic : Constant { v { 0.3 } } roughness : Constant { v { 0.4 } } "
              "specular_tint : Constant { v { 0.5 } } anisotropic : Constant { v { 0.4 } } sheen : Constant { v { 0.5 } } sheen_tint : Constant { v { 0.5 } } "
              "clearcoat : Constant { v { 0.6 } } clearcoat_gloss : Constant { v { 0.8 } } eta : Constant { v { 1.5 } } }",
    "disney_trans": "Surface m : Disney { color : Constant { v { 0.8, 0.85, 0.9 } } roughness : Constant { v { 0.3 } } "
                    "specular_trans : Constant { v { 0.7 } } flatness : Constant { v { 0.3 } } eta : Constant { v { 1.45 } } }",
    "disney_thin": "Surface m : Disney { thin { true } color : Constant { v { 0.6, 0.7, 0.5 } } roughness : Constant { v { 0.35 } } "
                   "specular_trans : Constant { v { 0.4 } } diffuse_trans : Constant { v { 0.8 } } flatness : Constant { v { 0.5 } } eta : Constant { v { 1.3 } } }",
    "mix": "Surface mix_a : Matte { Kd : Constant { v { 0.7, 0.2, 0.2 } } } "
           "Surface mix_b : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } roughness : Constant { v { 0.2 } } } "
           "Surface m : Mix { a { @mix_a } b { @mix_b } ratio : Constant { v { 0.3 } } }",
    "mix_glass": "Surface mg_a : Glass { Kr : Constant { v { 0.9 } } Kt : Constant { v { 0.9 } } roughness : Constant { v { 0.2 } } eta : Constant { v { 1.5 } } } "
                 "Surface mg_b : Disney { color : Constant { v { 0.5, 0.6, 0.7 } } roughness : Constant { v { 0.5 } } } "
                 "Surface m : Mix { a { @mg_a } b { @mg_b } ratio : Constant { v { 0.6 } } }",
    "mix_nested": "Surface mn_a : Matte { Kd : Constant { v { 0.7, 0.2, 0.2 } } } "
                  "Surface mn_b : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } roughness : Constant { v { 0.3 } } } "
                  "Surface mn_c : Plastic { Kd : Constant { v { 0.2, 0.3, 0.8 } } roughness : Constant { v { 0.3 } } eta : Constant { v { 1.5 } } } "
                  "Surface mn_d : Glass { Kr : Constant { v { 0.9 } } Kt : Constant { v { 0.9 } } roughness : Constant { v { 0.2 } } eta : Constant { v { 1.5 } } } "
                  "Surface mn_in : Mix { a { @mn_a } b { @mn_b } ratio : Constant { v { 0.25 } } } "
                  "Surface mn_in2 : Mix { a { @mn_d } b { @mn_in } ratio : Constant { v { 0.5 } } } "
                  "Surface m : Mix { a { @mn_in2 } b { @mn_c } ratio : Constant { v { 0.6 } } }",
    "layered": "Surface lay_t : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.15 } } eta : Constant { v { 1.5 } } } "
               "Surface lay_b : Matte { Kd : Constant { v { 0.7, 0.5, 0.3 } } } "
               "Surface m : Layered { top { @lay_t } bottom { @lay_b } thickness : Constant { v { 0.05 } } }",
    "layered_medium": "Surface lm_t : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.2 } } eta : Constant { v { 1.4 } } } "
                      "Surface lm_b : Metal { eta { \"Au\" } roughness : Constant { v { 0.3 } } } "
                      "Surface m : Layered { top { @lm_t } bottom { @lm_b } thickness : Constant { v { 0.3 } } g : Constant { v { 0.4 } } "
                      "albedo : Constant { v { 0.8, 0.6, 0.4 } } max_depth { 12 } samples { 2 } two_sided { true } }",
    # free composition of Mix and Layered (mix.cpp:82-212 and layered.cpp:195-500 hold arbitrary child closures)
    "mix_layered": "Surface ml_t : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.15 } } eta : Constant { v { 1.5 } } } "
                   "Surface ml_b : Matte { Kd : Constant { v { 0.3, 0.5, 0.7 } } } "
                   "Surface ml_l : Layered { top { @ml_t } bottom { @ml_b } thickness : Constant { v { 0.05 } } } "
                   "Surface ml_p : Plastic { Kd : Constant { v { 0.6, 0.3, 0.2 } } roughness : Constant { v { 0.3 } } eta : Constant { v { 1.5 } } } "
                   "Surface m : Mix { a { @ml_l } b { @ml_p } ratio : Constant { v { 0.6 } } }",
    "layered_mix": "Surface lx_g1 : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.1 } } eta : Constant { v { 1.5 } } } "
                   "Surface lx_g2 : Glass { Kr : Constant { v { 0.9 } } Kt : Constant { v { 0.9, 0.95, 1 } } roughness : Constant { v { 0.4 } } eta : Constant { v { 1.5 } } } "
                   "Surface lx_top : Mix { a { @lx_g1 } b { @lx_g2 } ratio : Constant { v { 0.5 } } } "
                   "Surface lx_m : Matte { Kd : Constant { v { 0.7, 0.5, 0.3 } } } "
                   "Surface lx_c : Metal { eta { \"Cu\" } roughness : Constant { v { 0.3 } } } "
                   "Surface lx_bot : Mix { a { @lx_m } b { @lx_c } ratio : Constant { v { 0.7 } } } "
                   "Surface m : Layered { top { @lx_top } bottom { @lx_bot } thickness : Constant { v { 0.1 } } }",
    # a Mix tree of six Mix levels (round 3: the kernels walk Mix trees with an explicit stack, lr_scene.h LR_MIX_MAX_DEPTH = 7), Mix nodes
    # on both sides of their parents, a different leaf kind at every level
    "mix_deep": "Surface md_a : Matte { Kd : Constant { v { 0.7, 0.2, 0.2 } } } "
                "Surface md_b : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } roughness : Constant { v { 0.3 } } } "
                "Surface md_c : Plastic { Kd : Constant { v { 0.2, 0.3, 0.8 } } roughness : Constant { v { 0.3 } } eta : Constant { v { 1.5 } } } "
                "Surface md_d : Glass { Kr : Constant { v { 0.9 } } Kt : Constant { v { 0.9 } } roughness : Constant { v { 0.2 } } eta : Constant { v { 1.5 } } } "
                "Surface md_e : Metal { eta { \"Au\" } roughness : Constant { v { 0.25 } } } "
                "Surface md_f : Disney { color : Constant { v { 0.5, 0.6, 0.7 } } roughness : Constant { v { 0.5 } } metallic : Constant { v { 0.3 } } } "
                "Surface md_1 : Mix { a { @md_a } b { @md_b } ratio : Constant { v { 0.25 } } } "
                "Surface md_2 : Mix { a { @md_d } b { @md_1 } ratio : Constant { v { 0.5 } } } "
                "Surface md_3 : Mix { a { @md_2 } b { @md_c } ratio : Constant { v { 0.6 } } } "
                "Surface md_s : Mix { a { @md_e } b { @md_f } ratio : Constant { v { 0.4 } } } "
                "Surface md_4 : Mix { a { @md_3 } b { @md_s } ratio : Constant { v { 0.7 } } } "
                "Surface md_5 : Mix { a { @md_b } b { @md_4 } ratio : Constant { v { 0.35 } } } "
                "Surface m : Mix { a { @md_5 } b { @md_a } ratio : Constant { v { 0.8 } } }",
    # a Layered surface as the bottom interface of a Layered surface (a clear coat over a tinted coat over paint), round 3
    "layered_layered": "Surface ll_t : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.1 } } eta : Constant { v { 1.5 } } } "
                       "Surface ll_t2 : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1, 0.8, 0.6 } } roughness : Constant { v { 0.3 } } eta : Constant { v { 1.3 } } } "
                       "Surface ll_b : Matte { Kd : Constant { v { 0.6, 0.6, 0.6 } } } "
                       "Surface ll_in : Layered { top { @ll_t2 } bottom { @ll_b } thickness : Constant { v { 0.05 } } } "
                       "Surface m : Layered { top { @ll_t } bottom { @ll_in } thickness : Constant { v { 0.02 } } g : Constant { v { 0.2 } } albedo : Constant { v { 0.5, 0.6, 0.7 } } }",
    "metal": 'Surface m : Metal { eta { "Cu" } roughness : Constant { v { 0.3, 0.15 } } Kd : Constant { v { 0.9, 0.9, 0.9 } } }',
}

# rejected by the loader with a clear error: a Mix tree deeper than LR_MIX_MAX_DEPTH = 7 levels below its root
def mix_too_deep():
    text = "Surface mz_a : Matte { Kd : Constant { v { 0.7, 0.2, 0.2 } } } Surface mz_b : Mirror { color : Constant { v { 0.9 } } } Surface mz_0 : Mix { a { @mz_a } b { @mz_b } } "
    for i in range(1, 8):
        text += f"Surface mz_{i} : Mix {{ a {{ @mz_{i - 1} }} b {{ @mz_b }} }} "
    return text + "Surface m : Mix { a { @mz_a } b { @mz_7 } }"


# rejected by the loader with a clear error (the kernels bound their call graph: at most two Layered levels, lr_scene.h LR_LAYERED_MAX_LEVELS)
LAYERED_THREE_DEEP = (MATERIALS["layered_layered"].replace("Surface m ", "Surface ll_mid ") +
                      " Surface m : Layered { top { @ll_t } bottom { @ll_mid } thickness : Constant { v { 0.01 } } }")

_PATCH = """
{surface}
Shape quad : InlineMesh {{ positions {{ -1,0,-1, 1,0,-1, 1,0,1, -1,0,1 }} indices {{ 0,1,2, 0,2,3 }} surface {{ @m }}
  light : Diffuse {{ emission : Constant {{ v {{ 1 }} }} }} }}
Camera cam : Pinhole {{ film : Color {{ resolution {{ 8, 8 }} }} spp {{ 1 }} position {{ 0, 3, 0 }} look_at {{ 0, 0, 0 }} up {{ 0, 0, -1 }} }}
render {{ cameras {{ @cam }} shapes {{ @quad }} integrator : MegaPath {{ }} }}
"""


def material_scene(name):
    return Scene.from_string(_PATCH.format(surface=MATERIALS[name]), build_accel=False)


def sph(theta, phi):
    return np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], np.float32)


class SurfaceProbe:
    """evaluate/sample of the patch surface on a flat patch (ng = +z) through the oracle hooks"""

    def __init__(self, scene, ns=(0.0, 0.0, 1.0)):
        self.o = oracle_lib()
        self.scene = scene
        self.view = scene.view()
        self.ns = np.array(ns, np.float32)
        self.tag = self.view.surface_count - 1  # the shape's surface is registered after a Mix's children

    def evaluate(self, wo, wi):
        wo, wi = np.ascontiguousarray(wo, np.float32), np.ascontiguousarray(wi, np.float32)
        out = np.zeros(4, np.float32)
        self.o.oracle_surface_evaluate(C.byref(self.view), self.tag, self.ns.ctypes.data, wo.ctypes.data, wi.ctypes.data, out.ctypes.data)
        return out[:3].copy(), float(out[3])

    def sample(self, wo, u_lobe, ux, uy):
        wo = np.ascontiguousarray(wo, np.float32)
        out = np.zeros(8, np.float32)
        self.o.oracle_surface_sample(C.byref(self.view), self.tag, self.ns.ctypes.data, wo.ctypes.data, u_lobe, ux, uy, out.ctypes.data)
        return out[:3].copy(), float(out[3]), out[4:7].copy(), int(out[7])
