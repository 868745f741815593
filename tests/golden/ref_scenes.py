"""Scenes of the reference-rendered fixtures tests/golden/ref_*.npz (made by make_ref_golden.py from oracle/_ref/libref.so, the
reference's own code; compared with the oracle on the CPU and with the HIP path on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_pfm(path, img):
    img = np.asarray(img, np.float32)
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img[::-1, :, :3].tobytes())


def sky(w=256, h=128):
    """a small deterministic lat-long sky: gradient + sun lobe"""
    y, x = np.mgrid[0:h, 0:w]
    theta, phi = (y + 0.5) / h * np.pi, (x + 0.5) / w * 2 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1)
    sun = np.array([0.5, 0.6, 0.3]) / np.linalg.norm([0.5, 0.6, 0.3])
    lobe = np.exp(60.0 * (d @ sun - 1.0))[..., None]
    base = np.stack([0.3 + 0.2 * d[..., 1], 0.4 + 0.3 * d[..., 1], 0.6 + 0.35 * d[..., 1]], -1).clip(0.02, None)
    return (base + 40.0 * lobe * np.array([1.0, 0.9, 0.7])).astype(np.float32)


ENV = """
Surface ground : Matte {{ Kd : Constant {{ v {{ 0.5, 0.5, 0.5 }} }} }}
Surface shiny : Plastic {{ Kd : Constant {{ v {{ 0.7, 0.2, 0.1 }} }} roughness : Constant {{ v {{ 0.15 }} }} }}
Shape quad : InlineMesh {{ positions {{ -20,0,-20, 20,0,-20, 20,0,20, -20,0,20 }} indices {{ 0,2,1, 0,3,2 }} surface {{ @ground }} }}
Shape cube : InlineMesh {{
  positions {{ -1,0,-1, 1,0,-1, 1,2,-1, -1,2,-1, -1,0,1, 1,0,1, 1,2,1, -1,2,1 }}
  indices {{ 0,2,1, 0,3,2,  4,5,6, 4,6,7,  0,1,5, 0,5,4,  3,6,2, 3,7,6,  0,7,3, 0,4,7,  1,2,6, 1,6,5 }}
  surface {{ @shiny }} transform : SRT {{ rotate {{ 0, 1, 0, 30 }} }} }}
Camera cam : Pinhole {{ fov {{ 40 }} spp {{ {spp} }} film : Color {{ resolution {{ 40, 30 }} clamp {{ 64 }} }}
  position {{ 0, 4, 9 }} look_at {{ 0, 1, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @quad, @cube }}
  environment : {env}
  integrator : MegaPath {{ depth {{ 6 }} rr_depth {{ 2 }} }} }}
"""

FOG = """
Medium fog : Homogeneous { sigma_a : Constant { v { 0.0001, 0.0002, 0.0003 } } sigma_s : Constant { v { 0.0006 } } eta { 1 }
  phasefunction : HenyeyGreenstein { g { 0.4 } } }
Medium inner : Homogeneous { sigma_a : Constant { v { 0.004, 0.002, 0.001 } } sigma_s : Constant { v { 0.003 } } eta { 1.3 } priority { 0 }
  phasefunction : HenyeyGreenstein { g { -0.3 } } }
Surface skin : Glass { Kr : Constant { v { 1 } } Kt : Constant { v { 1 } } roughness : Constant { v { 0.05 } } eta : Constant { v { 1.3 } } }
"""


def scenes(directory):
    """name -> (scene text, spp); image files the scenes read are written into `directory`"""
    from helpers import MATERIALS
    from luisarender_amd.scenes.cornell import cornell_box
    write_pfm(os.path.join(directory, "sky.pfm"), sky())
    out = {}
    out["cornell"] = (cornell_box(resolution=32, spp=8), 8)
    extra = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("mirror", "glass", "plastic", "metal"))
    out["materials"] = (cornell_box(resolution=40, spp=8, short_box_surface="glass", tall_box_surface="metal", extra_surfaces=extra), 8)
    extra = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("disney", "mix_nested"))
    out["disney_mix_sobol"] = (cornell_box(resolution=(40, 32), spp=8, short_box_surface="disney", tall_box_surface="mix_nested", extra_surfaces=extra,
                                           sampler="PaddedSobol", filter_impl="Gaussian", filter_radius=1.0, rr_depth=2), 8)
    out["thin_lens_plastic"] = (cornell_box(resolution=32, spp=8, short_box_surface="plastic", tall_box_surface="mirror",
                                            extra_surfaces="".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("plastic", "mirror")))
                                .replace("Camera cam : Pinhole {", "Camera cam : ThinLens {\n  aperture { 1.4 } focal_length { 50 } focus_distance { 900 }"), 8)
    img = 'Image { file { "sky.pfm" } encoding { "linear" } }'
    out["env_image"] = (ENV.format(spp=8, env=f"Spherical {{ emission : {img} transform : SRT {{ rotate {{ 0, 1, 0, 40 }} }} }}"), 8)
    out["env_combined"] = (ENV.format(spp=8, env=f"Combined {{ a : Spherical {{ emission : {img} }} "
                                                 "b : Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 8 } direction { 0.4, 1, 0.3 } } scale_a { 0.7 } scale_b { 1.5 } }"), 8)
    glass = MATERIALS["glass"].replace("Surface m ", "Surface probe ") + "\n"
    direct = cornell_box(resolution=32, spp=8, short_box_surface="probe", extra_surfaces=glass).replace("integrator : MegaPath {", 'integrator : Direct { importance_sampling { "both" }')
    out["direct_both"] = (direct.replace("render {", "render {\n  environment : Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 40 } direction { 0.1, 0.2, -1 } }"), 8)
    vpt = cornell_box(resolution=32, spp=8, depth=8, extra_surfaces=FOG, short_box_surface="skin").replace("integrator : MegaPath {", "integrator : MegaVPTNaive {")
    vpt = vpt.replace("render {", "render {\n  environment_medium { @fog }").replace("surface { @skin }", "surface { @skin } medium { @inner }")
    out["vpt_fog_medium_box"] = (vpt, 8)
    # the remaining precompiled kernel variants (csrc/hip/variants.h): Disney alone (16), environment + Disney (20), Layered (124)
    # and the generic-sampler twin of the lean kernel (2)
    dis = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("disney", "disney_trans"))
    out["disney"] = (cornell_box(resolution=32, spp=8, short_box_surface="disney", tall_box_surface="disney_trans", extra_surfaces=dis), 8)
    out["env_disney"] = (ENV.format(spp=8, env=f"Spherical {{ emission : {img} }}").replace(
        "Surface shiny : Plastic { Kd : Constant { v { 0.7, 0.2, 0.1 } } roughness : Constant { v { 0.15 } } }",
        MATERIALS["disney"].replace("Surface m ", "Surface shiny ")), 8)
    out["cornell_sobol"] = (cornell_box(resolution=(40, 24), spp=8, sampler="Sobol"), 8)
    lay = MATERIALS["layered"].replace("Surface m ", "Surface layered ") + "\n"
    out["layered"] = (cornell_box(resolution=32, spp=256, short_box_surface="layered", tall_box_surface="layered", extra_surfaces=lay), 256)
    # Mix and Layered nested in each other (kernel variant 636): a Mix with a Layered leaf, a Layered surface with Mix interfaces
    nest = "".join(MATERIALS[k].replace("Surface m ", f"Surface {k} ") + "\n" for k in ("mix_layered", "layered_mix"))
    out["nested"] = (cornell_box(resolution=32, spp=256, short_box_surface="mix_layered", tall_box_surface="layered_mix", extra_surfaces=nest), 256)
    # round 3: a Layered surface as the bottom interface of a Layered surface (dev_layered.h: two Layered levels), and Combined
    # environments three deep (dev_shade.h: env_evaluate_tree)
    ll = MATERIALS["layered_layered"].replace("Surface m ", "Surface layered_layered ") + "\n"
    out["layered_layered"] = (cornell_box(resolution=32, spp=256, short_box_surface="layered_layered", tall_box_surface="layered_layered", extra_surfaces=ll), 256)
    sun2 = "Directional { emission : Constant { v { 1, 2, 4 } } angle { 12 } direction { -0.6, 0.5, 0.2 } }"
    dome2 = f"Spherical {{ emission : {img} scale {{ 0.4 }} compensate_mis {{ false }} transform : SRT {{ rotate {{ 0.3, 1, 0, 200 }} }} }}"
    inner = f"Combined {{ a : {dome2} b : {sun2} scale_a {{ 1.2 }} scale_b {{ 0.8 }} transform : SRT {{ rotate {{ 0, 0, 1, 25 }} }} }}"
    sun = "Directional { emission : Constant { v { 3, 2.5, 2 } } angle { 8 } direction { 0.4, 1, 0.3 } }"
    middle = f"Combined {{ a : {sun} b : {inner} scale_a {{ 1.5 }} scale_b {{ 0.6 }} transform : SRT {{ rotate {{ 1, 0, 0, -15 }} }} }}"
    out["env_combined_nested"] = (ENV.format(spp=8, env=f"Combined {{ a : Spherical {{ emission : {img} transform : SRT {{ rotate {{ 1, 0, 0, 20 }} }} }} b : {middle} "
                                                        "scale_a { 0.7 } scale_b { 1.1 } transform : SRT { rotate { 0, 1, 0, 60 } } }"), 8)
    # the same without an area light to run into: lit through the open front by a Directional + image environment.  No emitter
    # is ever evaluated from a ray origin lying IN its surface (mega_vpt_naive.cpp:331 after homogeneous.cpp:64), which is what
    # makes the lamp-lit case above chaotic in the last bit
    lamp = "\n  light : Diffuse { emission : Constant { v { 17, 12, 4 } } }"
    assert lamp in vpt
    env = (f"Combined {{ a : Spherical {{ emission : {img} }} b : Directional {{ emission : Constant {{ v {{ 30, 25, 20 }} }} angle {{ 30 }} "
           "direction { 0, 0.3, -1 } } scale_a { 0.5 } scale_b { 1 } }")
    out["vpt_fog_env_medium_box"] = (vpt.replace(lamp, "").replace("render {", "render {\n  environment : " + env), 8)
    return out
