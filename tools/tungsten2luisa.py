#!/usr/bin/env python3
"""Tungsten scene (.json) -> LuisaRender scene description (.luisa): the asset pipeline of SURVEY §8 f4.

The reference ships a converter with the same name and job (tools/tungsten2luisa.py: the README scenes are Benedikt
Bitterli's Tungsten scenes run through it); this one is written from the two formats, on numpy instead of PyGLM, and
emits the same nodes with the same property names and conventions:

  bsdfs        lambert / oren_nayar -> Matte; plastic / rough_plastic -> Plastic; dielectric / rough_dielectric /
               thinsheet -> Glass (Kr 1, Kt albedo); mirror -> Mirror; conductor / rough_conductor -> Metal (named
               material, or a flat (n, k) spectrum 360..830 nm); transparency -> the base surface + `alpha`; null -> Null;
               anything else -> white Matte.  Roughness: alpha -> sqrt(alpha) (tools/tungsten2luisa.py:7-8).
  textures     a file name -> Image; {"type": "checker"} -> Checkerboard; a colour / number -> Constant
  primitives   mesh (.wo3 -> the .obj next to it), quad (InlineMesh), cube / disk / sphere (models/*.obj);
               infinite_sphere -> Spherical environment (rotated -90 degrees about y), infinite_sphere_cap -> Directional,
               skydome -> Spherical on textures/sky.exr; emission or power -> a Diffuse light (power / (100 pi), a quad's
               power / (sx sz pi)); transform = T * R_yxz * S (Tungsten composes rotations y, x, z)
  camera       pinhole; Tungsten's horizontal fov -> LuisaRender's vertical fov; Gaussian filter of radius 1
  render       integrator MegaPath with the sampler named by --sampler (the reference converter writes PMJ02BN, whose
               sample tables are not part of the reference snapshot: this loader rejects it, so the default here is PaddedSobol)

    python tools/tungsten2luisa.py scene.json 1024 [--sampler PaddedSobol] [-o scene.luisa]
"""
from __future__ import annotations

import argparse
import json
import math
import os

import numpy as np


def _vec3(v, default=0.0):
    if v is None:
        v = default
    a = np.asarray(v, np.float64).reshape(-1)
    return np.full(3, a[0]) if a.size == 1 else a[:3].copy()


def _fmt(x) -> str:
    return repr(float(x))


def _list(v) -> str:
    return ", ".join(_fmt(x) for x in v)


def texture(value) -> str:
    """an inline Texture node for an albedo / emission entry"""
    if isinstance(value, str):
        return f'Image {{ file {{ "{value}" }} }}'
    if isinstance(value, dict):
        if value.get("type") != "checker":
            raise ValueError(f"unsupported texture {value}")
        on, off = _vec3(value["on_color"]), _vec3(value["off_color"])
        return (f'Checkerboard {{ on : Constant {{ v {{ {_list(on)} }} }} off : Constant {{ v {{ {_list(off)} }} }} '
                f'scale {{ {_fmt(value["res_u"])}, {_fmt(value["res_v"])} }} }}')
    return f"Constant {{ v {{ {_list(_vec3(value))} }} }}"


def _roughness(material) -> str:
    return f'roughness : Constant {{ v {{ {_fmt(math.sqrt(material.get("roughness", 1e-6)))} }} }}'


def surface(name: str, material: dict, alpha: str = "") -> list[str]:
    """the Surface node(s) of one Tungsten bsdf"""
    kind = material["type"]
    head = f"Surface mat_{name}"
    if kind in ("plastic", "rough_plastic"):
        return [f'{head} : Plastic {{ Kd : {texture(material["albedo"])} eta : Constant {{ v {{ {_fmt(material["ior"])} }} }} {_roughness(material)}{alpha} }}']
    if kind in ("dielectric", "rough_dielectric", "thinsheet"):  # (a thin sheet is rendered as glass, like the reference converter)
        return [f'{head} : Glass {{ Kr : Constant {{ v {{ 1, 1, 1 }} }} Kt : {texture(material.get("albedo", 1.0))} '
                f'eta : Constant {{ v {{ {_fmt(material.get("ior", 1.5))} }} }} {_roughness(material)}{alpha if kind != "thinsheet" else ""} }}']
    if kind == "mirror":
        return [f'{head} : Mirror {{ color : {texture(material["albedo"])}{alpha} }}']
    if kind in ("conductor", "rough_conductor"):
        if "material" in material:
            eta = f'"{material["material"]}"'
        else:
            n, k = _fmt(np.mean(_vec3(material["eta"]))), _fmt(np.mean(_vec3(material["k"])))
            eta = f"360, {n}, {k}, 830, {n}, {k}"
        albedo = material.get("albedo", 1.0)
        kd = "" if np.allclose(_vec3(albedo) if not isinstance(albedo, (str, dict)) else 0.0, 1.0) else f" Kd : {texture(albedo)}"
        return [f"{head} : Metal {{ eta {{ {eta} }} {_roughness(material)}{alpha}{kd} }}"]
    if kind in ("lambert", "oren_nayar"):
        return [f'{head} : Matte {{ Kd : {texture(material["albedo"])}{alpha} }}']
    if kind == "transparency":
        a = material["alpha"]
        if isinstance(a, (int, float)):
            wrapped = f" alpha : Constant {{ v {{ {_fmt(a)} }} }}"
        else:  # the alpha channel lives in a side file "<name>-alpha.<ext>"
            stem, ext = os.path.splitext(a)
            wrapped = f' alpha : Image {{ file {{ "{stem}-alpha{ext}" }} encoding {{ "linear" }} }}'
        return surface(name, dict(material["base"]), wrapped)
    if kind == "null":
        return [f"{head} : Null {{}}"]
    print(f"warning: unsupported bsdf '{kind}' ({name}): white Matte instead")
    return [f"{head} : Matte {{ Kd : Constant {{ v {{ 1, 1, 1 }} }} }}"]


def rotation_yxz(r):
    """Tungsten's Mat4f::rotYXZ: angles (x, y, z) in radians, applied y, then x, then z"""
    cx, cy, cz = np.cos(r)
    sx, sy, sz = np.sin(r)
    return np.array([[cy * cz - sy * sx * sz, -cy * sz - sy * sx * cz, -sy * cx],
                     [cx * sz, cx * cz, -sx],
                     [sy * cz + cy * sx * sz, -sy * sz + cy * sx * cz, cy * cx]])


def rotation_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def transform_matrix(t: dict) -> np.ndarray:
    m = np.eye(4)
    m[:3, :3] = rotation_yxz(np.radians(_vec3(t.get("rotation"), 0.0))) @ np.diag(_vec3(t.get("scale"), 1.0))
    m[:3, 3] = _vec3(t.get("position"), 0.0)
    return m


def _matrix_node(m: np.ndarray) -> str:
    rows = ",\n      ".join(_list(row) for row in m)  # row-major text (src/transforms/matrix.cpp)
    return f"transform : Matrix {{\n    m {{\n      {rows}\n    }}\n  }}"


def primitive(index: int, shape: dict, out: list[str]) -> str | None:
    """appends the node(s) of one primitive; returns the shape reference, or None for environments"""
    kind = shape["type"]
    t = shape.get("transform", {})
    m = transform_matrix(t)
    if kind == "infinite_sphere":
        out.append(f'Env env : Spherical {{\n  emission : {texture(shape["emission"])}\n  transform : SRT {{ rotate {{ 0, 1, 0, -90 }} }}\n}}')
        return None
    if kind == "infinite_sphere_cap":
        e = _vec3(shape["power"]) / (100.0 * math.pi)
        out.append(f'Env dir : Directional {{\n  emission : Constant {{ v {{ {_list(e)} }} }}\n  angle {{ {_fmt(shape["cap_angle"])} }}\n  '
                   f"{_matrix_node(m)}\n  scale {{ {_fmt(4.0 * math.pi)} }}\n}}")
        return None
    if kind == "skydome":
        c, s = math.cos(math.radians(-90)), math.sin(math.radians(-90))
        sky = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float64)
        out.append(f'Env sky : Spherical {{\n  emission : Image {{ file {{ "textures/sky.exr" }} }}\n  {_matrix_node(sky)}\n  scale {{ {_fmt(shape.get("intensity", 1.0))} }}\n}}')
        print("warning: a skydome is converted to a Spherical environment on textures/sky.exr")
        return None
    impl, power_scale = "Mesh", 100.0 * math.pi
    if kind == "mesh":
        stem, ext = os.path.splitext(shape["file"])
        body = f'file {{ "{stem}.obj" }}' if ext == ".wo3" else f'file {{ "{shape["file"]}" }}'
    elif kind == "quad":  # Tungsten's quad spans [-0.5, 0.5]^2 in its xz plane, facing +y
        impl = "InlineMesh"
        body = "positions { 1, 1, 0, -1, 1, 0, -1, -1, 0, 1, -1, 0 }\n  indices { 0, 1, 2, 0, 2, 3 }"
        s = _vec3(t.get("scale"), 1.0)
        power_scale = s[0] * s[2] * math.pi
        local = np.eye(4)
        local[:3, :3] = rotation_x(math.radians(-90)) * 0.5
        m = m @ local
    elif kind == "cube":
        body = 'file { "models/cube.obj" }'
        local = np.eye(4)
        local[:3, :3] = rotation_x(math.radians(-90)) * 0.5
        m = m @ local
    elif kind in ("disk", "sphere"):
        body = f'file {{ "models/{kind}.obj" }}'
    else:
        raise NotImplementedError(f"unsupported primitive '{kind}'")
    bsdf = shape.get("bsdf", "Null")
    if isinstance(bsdf, dict):  # an inline bsdf
        if bsdf["type"] == "null":
            bsdf = "Null"
        else:
            out.extend(surface(f"shape_{index}", dict(bsdf)))
            bsdf = f"shape_{index}"
    emission = shape.get("emission")
    e = _vec3(emission) if emission is not None else _vec3(shape.get("power", 0.0)) / power_scale
    light = "" if not e.any() else f"\n  light : Diffuse {{ emission : Constant {{ v {{ {_list(e)} }} }} }}"
    out.append(f"Shape shape_{index} : {impl} {{\n  {body}\n  surface {{ @mat_{bsdf} }}{light}\n  {_matrix_node(m)}\n}}")
    return f"@shape_{index}"


def camera(cam: dict, spp: int) -> str:
    res = np.asarray(cam["resolution"], np.float64).reshape(-1)
    w, h = (float(res[0]), float(res[0])) if res.size == 1 else (float(res[0]), float(res[1]))
    fov = math.degrees(2.0 * math.atan(h * math.tan(0.5 * math.radians(cam["fov"])) / w))  # horizontal -> vertical
    t = cam["transform"]
    position, look_at, up = _vec3(t["position"]), _vec3(t["look_at"]), _vec3(t.get("up", [0, 1, 0]))
    front = (look_at - position) / np.linalg.norm(look_at - position)
    return (f"Camera camera : Pinhole {{\n  fov {{ {_fmt(fov)} }}\n  spp {{ {int(spp)} }}\n  filter : Gaussian {{ radius {{ 1 }} }}\n"
            f"  film : Color {{ resolution {{ {int(w)}, {int(h)} }} }}\n  file {{ \"render.exr\" }}\n"
            f"  transform : View {{\n    position {{ {_list(position)} }}\n    front {{ {_list(front)} }}\n    up {{ {_list(up)} }}\n  }}\n}}")


def convert(scene: dict, spp: int, sampler: str = "PaddedSobol") -> str:
    out: list[str] = []
    for material in scene.get("bsdfs", []):
        out.extend(surface(material["name"], dict(material)))
    out.append("Surface mat_Null : Null {}")
    refs, kinds = [], []
    for i, shape in enumerate(scene.get("primitives", [])):
        ref = primitive(i, shape, out)
        kinds.append(shape["type"])
        if ref is not None:
            refs.append(ref)
    out.append(camera(scene["camera"], spp))
    env = "environment : Null {}"
    if "infinite_sphere" in kinds:
        env = "environment { @env }"
    elif "infinite_sphere_cap" in kinds:
        env = "environment { @dir }"
    shapes = ",\n    ".join(refs)
    out.append(f"render {{\n  cameras {{ @camera }}\n  integrator : MegaPath {{\n    sampler : {sampler} {{}}\n  }}\n  shapes {{\n    {shapes}\n  }}\n  {env}\n}}")
    return "\n\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("scene", help="Tungsten scene.json")
    ap.add_argument("spp", type=int)
    ap.add_argument("--sampler", default="PaddedSobol", help="Independent | PCG32 | Sobol | PaddedSobol (the reference converter writes PMJ02BN)")
    ap.add_argument("-o", "--output", default=None)
    args = ap.parse_args()
    with open(args.scene) as f:
        scene = json.load(f)
    text = convert(scene, args.spp, args.sampler)
    path = args.output or os.path.splitext(args.scene)[0] + ".luisa"
    with open(path, "w") as f:
        f.write(text)
    print(f"wrote {path}")


if __name__ == "__main__":
    main()
