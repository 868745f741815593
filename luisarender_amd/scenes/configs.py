"""Procedural stand-ins for BASELINE.json configs C3..C5 (the README assets are external
downloads, so every workload is generated from a seed).

  C3 "Bedroom-class"  1280x720, 4096 spp: the C2 room generator with window openings, 10 % Glass fixtures and an
                      image-based `Spherical` environment (procedural 2048x1024 sun + sky gradient, written as EXR).
  C4 "Camera-class"   3840x2160, 1024 spp: ~1 M triangles, Disney / Plastic / Matte fixtures driven by 8 procedural
                      2048^2 albedo (8-bit sRGB PNG) and roughness (8-bit grey PNG, linear) images, environment light.
  C5 "Kitchen-class"  1280x720, 65536 spp: every closure of SURVEY row a14 — Matte (Lambert + Oren-Nayar), Mirror,
                      Glass, Plastic, Metal, Disney thick + thin, Mix, Layered — plus NormalMap / Opacity wrappers.

All three reuse the meshes and the room of `bathroom.py` (icosphere / torus / tessellated box OBJ files + InlineMesh
walls) with UV coordinates added where textures need them; every generator is seeded (default 19980810).
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

from .bathroom import _quad_shape, icosphere, inline_mesh, tess_box, torus


def write_png(path, img8):
    """8-bit grey / RGB PNG (filter 0), enough for the procedural textures; read back by csrc/host/image_io.cpp"""
    img8 = np.ascontiguousarray(img8, np.uint8)
    if img8.ndim == 2:
        img8 = img8[..., None]
    h, w, c = img8.shape
    color_type = {1: 0, 3: 2, 4: 6}[c]

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    raw = np.concatenate([np.zeros((h, 1), np.uint8), img8.reshape(h, w * c)], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 3)) + chunk(b"IEND", b""))


def write_pfm(path, img):
    """float RGB PFM (bottom row first)"""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1, :, :3]).tobytes())


def _image(out_dir, stem, img8, inline, srgb):
    """an 8-bit procedural image as PNG -- or, `inline`, as a float PFM holding byte / 255 with the encoding spelled out (the
    reference's code compiled on the scalar stand-in reads PFM only, oracle/ref_shim); returns (file name, encoding property)"""
    if not inline:
        write_png(os.path.join(out_dir, stem + ".png"), img8)
        return stem + ".png", "" if srgb else 'encoding { "linear" }'
    img = np.asarray(img8, np.float32) / 255.0
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=-1)
    write_pfm(os.path.join(out_dir, stem + ".pfm"), img)
    return stem + ".pfm", 'encoding { "sRGB" }' if srgb else 'encoding { "linear" }'


def _sky(out_dir, img, inline):
    from ..scene import save_image
    if inline:
        write_pfm(os.path.join(out_dir, "sky.pfm"), img)
        return 'sky.pfm" } encoding { "linear'  # (only .exr / .hdr default to linear, src/textures/image.cpp:85-90)
    save_image(os.path.join(out_dir, "sky.exr"), img)
    return "sky.exr"


def sun_sky_image(w=2048, h=1024, sun_dir=(0.55, 0.62, 0.3), sun_power=900.0, sun_sharpness=600.0):
    """lat-long HDR in the reference's Spherical parameterisation (spherical.cpp:88-95: uv = (phi / 2pi, theta / pi),
    theta from +y... the host side owns the convention; this only has to be a smooth, strongly peaked map)"""
    v = (np.arange(h) + 0.5) / h
    u = (np.arange(w) + 0.5) / w
    theta = v[:, None] * np.pi
    phi = u[None, :] * 2.0 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) + 0 * phi, np.sin(theta) * np.sin(phi)], -1)
    s = np.asarray(sun_dir, np.float64)
    s /= np.linalg.norm(s)
    up = np.clip(d[..., 1], -1.0, 1.0)
    sky = np.where(up[..., None] > 0.0,
                   np.array([0.35, 0.55, 1.0]) * (0.35 + 0.65 * (1.0 - up[..., None]) ** 2) + np.array([0.9, 0.8, 0.7]) * 0.25 * (1.0 - up[..., None]) ** 8,
                   np.array([0.12, 0.11, 0.10]) * (1.0 + up[..., None]))
    lobe = np.exp(sun_sharpness * (np.clip((d * s).sum(-1), -1.0, 1.0) - 1.0))
    img = sky + sun_power * lobe[..., None] * np.array([1.0, 0.92, 0.8])
    rgba = np.ones((h, w, 4), np.float32)
    rgba[..., :3] = img
    return rgba


def albedo_image(rng, n=2048):
    """smooth band-limited colour pattern + tiles (8-bit sRGB)"""
    y, x = np.meshgrid(np.linspace(0, 1, n, endpoint=False), np.linspace(0, 1, n, endpoint=False), indexing="ij")
    img = np.zeros((n, n, 3))
    for c in range(3):
        acc = np.zeros((n, n))
        for _ in range(5):
            fx, fy = rng.integers(1, 9, 2)
            acc += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * (fx * x + fy * y) + rng.uniform(0, 2 * np.pi))
        img[..., c] = acc
    img = (img - img.min()) / (img.max() - img.min())
    tiles = ((np.floor(x * 16) + np.floor(y * 16)) % 2) * 0.25
    img = np.clip(0.12 + 0.7 * img + tiles[..., None] * 0.3, 0.0, 1.0)
    return (img * 255.0 + 0.5).astype(np.uint8)


def roughness_image(rng, n=2048, lo=0.08, hi=0.6):
    y, x = np.meshgrid(np.linspace(0, 1, n, endpoint=False), np.linspace(0, 1, n, endpoint=False), indexing="ij")
    acc = np.zeros((n, n))
    for _ in range(4):
        fx, fy = rng.integers(1, 12, 2)
        acc += np.sin(2 * np.pi * (fx * x + fy * y) + rng.uniform(0, 2 * np.pi))
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    return ((lo + (hi - lo) * acc) * 255.0 + 0.5).astype(np.uint8)


def normal_image(rng, n=512, strength=0.35):
    """tangent-space normal map of a sum of sines (8-bit, linear encoding requested in the scene)"""
    y, x = np.meshgrid(np.linspace(0, 1, n, endpoint=False), np.linspace(0, 1, n, endpoint=False), indexing="ij")
    dx, dy = np.zeros((n, n)), np.zeros((n, n))
    for _ in range(3):
        fx, fy = rng.integers(2, 10, 2)
        ph = rng.uniform(0, 2 * np.pi)
        dx += fx * np.cos(2 * np.pi * (fx * x + fy * y) + ph)
        dy += fy * np.cos(2 * np.pi * (fx * x + fy * y) + ph)
    nrm = np.stack([-strength * dx / 10.0, -strength * dy / 10.0, np.ones((n, n))], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return ((nrm * 0.5 + 0.5) * 255.0 + 0.5).astype(np.uint8)


def _write_obj_uv(path, v, f, n, uv):
    with open(path, "w") as out:
        for p in v:
            out.write(f"v {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
        for t in uv:
            out.write(f"vt {t[0]:.7g} {t[1]:.7g}\n")
        if n is not None:
            for p in n:
                out.write(f"vn {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
            for t in f + 1:
                out.write(f"f {t[0]}/{t[0]}/{t[0]} {t[1]}/{t[1]}/{t[1]} {t[2]}/{t[2]}/{t[2]}\n")
        else:
            for t in f + 1:
                out.write(f"f {t[0]}/{t[0]} {t[1]}/{t[1]} {t[2]}/{t[2]}\n")


def _uv_meshes(levels=(3, 4), torus_res=(48, 24), box_n=8):
    """the C2 meshes with a UV parameterisation (lat-long on spheres, (u, v) on tori, planar per box face)"""
    meshes = {}
    for lv in levels:
        v, f, n = icosphere(lv)
        uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(v[:, 1], -1, 1)) / np.pi], -1)
        meshes[f"ico{lv}"] = (v, f, n, uv)
    nu, nv = torus_res
    v, f, n = torus(nu, nv)
    uu, ww = np.meshgrid(np.arange(nu) / nu, np.arange(nv) / nv, indexing="ij")
    meshes["torus"] = (v, f, n, np.stack([uu.reshape(-1) * 2.0, ww.reshape(-1)], -1))
    v, f, _ = tess_box(box_n)
    per_face = (box_n + 1) ** 2
    uv = np.zeros((len(v), 2))
    for face in range(6):
        axis = face // 2
        p = v[face * per_face:(face + 1) * per_face]
        uv[face * per_face:(face + 1) * per_face, 0] = p[:, (axis + 1) % 3] * 0.5 + 0.5
        uv[face * per_face:(face + 1) * per_face, 1] = p[:, (axis + 2) % 3] * 0.5 + 0.5
    meshes[f"box{box_n}"] = (v, f, None, uv)
    return meshes


def _room(out, shapes, open_windows, X=4.0, Y=3.0, Z=4.0, lamps=True, wall="wall", floor="floor_s"):
    out.append(_quad_shape("floor", [(0, 0, 0), (0, 0, Z), (X, 0, Z), (X, 0, 0)], floor))
    out.append(_quad_shape("ceil", [(0, Y, 0), (X, Y, 0), (X, Y, Z), (0, Y, Z)], wall))
    out.append(_quad_shape("wall_back", [(0, 0, 0), (X, 0, 0), (X, Y, 0), (0, Y, 0)], wall))
    out.append(_quad_shape("wall_front", [(0, 0, Z), (0, Y, Z), (X, Y, Z), (X, 0, Z)], wall))
    out.append(_quad_shape("wall_left", [(0, 0, 0), (0, Y, 0), (0, Y, Z), (0, 0, Z)], wall))
    shapes += ["@floor", "@ceil", "@wall_back", "@wall_front", "@wall_left"]
    if not open_windows:
        out.append(_quad_shape("wall_right", [(X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0)], wall))
        shapes.append("@wall_right")
    else:
        strips = [[(X, 0, 0), (X, 0, Z), (X, 0.9, Z), (X, 0.9, 0)], [(X, 2.3, 0), (X, 2.3, Z), (X, Y, Z), (X, Y, 0)],
                  [(X, 0.9, 0), (X, 0.9, 0.6), (X, 2.3, 0.6), (X, 2.3, 0)], [(X, 0.9, 1.8), (X, 0.9, 2.2), (X, 2.3, 2.2), (X, 2.3, 1.8)],
                  [(X, 0.9, 3.4), (X, 0.9, Z), (X, 2.3, Z), (X, 2.3, 3.4)]]
        for i, q in enumerate(strips):
            out.append(_quad_shape(f"wall_right{i}", q, wall))
            shapes.append(f"@wall_right{i}")
    if lamps:
        for i, (cx, cz) in enumerate([(1.0, 1.0), (3.0, 1.2), (2.0, 3.0)]):
            h, y = 0.35, Y - 0.02
            out.append(_quad_shape(f"lamp{i}", [(cx - h, y, cz - h), (cx + h, y, cz - h), (cx + h, y, cz + h), (cx - h, y, cz + h)],
                                   wall, light="18, 16, 13"))
            shapes.append(f"@lamp{i}")


def _scatter(out, shapes, rng, tri_counts, target_triangles, pick_surface, X=4.0, Z=4.0, weights=None):
    names = list(tri_counts)
    weights = np.array(weights if weights is not None else [0.3, 0.2, 0.25, 0.25])
    total, i = 0, 0
    while total < target_triangles:
        k = names[rng.choice(len(names), p=weights)]
        surf = pick_surface(rng)
        s = rng.uniform(0.05, 0.16)
        x, z = rng.uniform(0.3, X - 0.3), rng.uniform(0.3, Z - 1.2)
        y = rng.choice([s, rng.uniform(0.4, 2.2)], p=[0.6, 0.4])
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 360)
        out.append(f"Shape obj{i} : Instance {{ shape {{ @mesh_{k} }} surface {{ @{surf} }} transform : SRT {{ "
                   f"scale {{ {s:.4f}, {s:.4f}, {s:.4f} }} rotate {{ {axis[0]:.4f}, {axis[1]:.4f}, {axis[2]:.4f}, {ang:.2f} }} "
                   f"translate {{ {x:.4f}, {y:.4f}, {z:.4f} }} }} }}\n")
        shapes.append(f"@obj{i}")
        total += tri_counts[k]
        i += 1
    return total


def _tail(out, shapes, env, resolution, spp, depth, seed, sampler, file, fov=55, filter_="Gaussian { radius { 1 } }", camera="Pinhole", camera_extra=""):
    out.append(f"""Camera cam : {camera} {{
  fov {{ {fov} }}  spp {{ {spp} }}  file {{ "{file}" }} {camera_extra}
  film : Color {{ resolution {{ {resolution[0]}, {resolution[1]} }} }}
  filter : {filter_}
  transform : View {{ position {{ 2.0, 1.6, 3.85 }}  front {{ 0, -0.15, -1 }}  up {{ 0, 1, 0 }} }}
}}
render {{
  cameras {{ @cam }}
  shapes {{ {", ".join(shapes)} }}
{env}  integrator : MegaPath {{ depth {{ {depth} }}  sampler : {sampler} {{ seed {{ {seed} }} }} }}
}}
""")


def _write_meshes(out_dir, out, meshes, inline=False):
    for k, (v, f, n, uv) in meshes.items():
        if inline:
            out.append(inline_mesh(f"mesh_{k}", v, f, n, uv))
        else:
            _write_obj_uv(os.path.join(out_dir, f"{k}.obj"), v, f, n, uv)
            out.append(f'Shape mesh_{k} : Mesh {{ file {{ "{k}.obj" }} }}\n')
    return {k: len(m[1]) for k, m in meshes.items()}


def _const(v):
    v = np.atleast_1d(v)
    return "Constant { v { " + ", ".join(f"{x:.4f}" for x in v) + " } }"


def generate_bedroom_scene(out_dir, target_triangles=600_000, resolution=(1280, 720), spp=4096, depth=16, seed=19980810,
                           env_resolution=(2048, 1024), sampler="Independent", file="render.exr", name="bedroom", inline=False,
                           mesh_levels=(3, 4), torus_res=(48, 24), box_n=8):
    """C3: window openings + image environment (sun/sky) + 10 % Glass; no interior lamps (the room is lit through the
    windows and by the sun lobe, as the reference's Bedroom is)."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    sky = _sky(out_dir, sun_sky_image(*env_resolution, sun_dir=(1.0, 0.55, 0.1)), inline)
    out, shapes = [], []
    palette = {"matte": [], "plastic": [], "metal": [], "glass": [], "mirror": []}
    for i in range(12):
        out.append(f"Surface matte{i} : Matte {{ Kd : {_const(rng.uniform(0.15, 0.85, 3))} }}\n")
        palette["matte"].append(f"matte{i}")
    for i in range(6):
        out.append(f"Surface plastic{i} : Plastic {{ Kd : {_const(rng.uniform(0.1, 0.8, 3))} roughness : {_const(rng.uniform(0.05, 0.4))} eta : {_const(1.5)} }}\n")
        palette["plastic"].append(f"plastic{i}")
    for i, m in enumerate(["Al", "Cu", "Au", "Ag"]):
        out.append(f'Surface metal{i} : Metal {{ eta {{ "{m}" }} roughness : {_const(rng.uniform(0.1, 0.4))} }}\n')
        palette["metal"].append(f"metal{i}")
    for i, (r, eta) in enumerate([(0.0, "1.5"), (0.1, "1.5"), (0.0, "1.33"), (0.05, "1.52, 1.53, 1.55")]):  # incl. a dispersive glass
        rough = f"roughness : {_const(r)} " if r > 0 else ""
        out.append(f"Surface glass{i} : Glass {{ {rough}eta : Constant {{ v {{ {eta} }} }} }}\n")
        palette["glass"].append(f"glass{i}")
    out.append("Surface mirror0 : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } }\n")
    palette["mirror"].append("mirror0")
    out.append("Surface wall : Matte { Kd : Constant { v { 0.75, 0.73, 0.7 } } }\n")
    out.append("Surface floor_s : Matte { Kd : Constant { v { 0.4, 0.35, 0.3 } } }\n")
    tri_counts = _write_meshes(out_dir, out, _uv_meshes(levels=mesh_levels, torus_res=torus_res, box_n=box_n), inline)
    _room(out, shapes, open_windows=True, lamps=False)
    classes = ["matte", "plastic", "metal", "glass", "mirror"]
    probs = np.array([0.58, 0.19, 0.08, 0.10, 0.05])

    def pick(r):
        c = classes[r.choice(len(classes), p=probs)]
        return palette[c][r.integers(len(palette[c]))]

    _scatter(out, shapes, rng, tri_counts, target_triangles, pick)
    env = f'  environment : Spherical {{ emission : Image {{ file {{ "{sky}" }} }} transform : SRT {{ rotate {{ 0, 1, 0, 15 }} }} }}\n'
    _tail(out, shapes, env, resolution, spp, depth, seed, sampler, file)
    path = os.path.join(out_dir, f"{name}.luisa")
    with open(path, "w") as f:
        f.write("".join(out))
    return path


def generate_camera_scene(out_dir, target_triangles=1_000_000, resolution=(3840, 2160), spp=1024, depth=16, seed=19980810,
                          texture_size=2048, env_resolution=(2048, 1024), sampler="Independent", file="render.exr", name="camera",
                          inline=False, mesh_levels=(3, 4), torus_res=(48, 24), box_n=8):
    """C4: Disney / Plastic / Matte mix on 8 procedural images (4 albedo sRGB PNG + 4 roughness grey PNG), env + lamps."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    sky = _sky(out_dir, sun_sky_image(*env_resolution, sun_power=300.0), inline)
    out, shapes = [], []
    for i in range(4):
        fa, ea = _image(out_dir, f"albedo{i}", albedo_image(rng, texture_size), inline, srgb=True)
        fr, er = _image(out_dir, f"rough{i}", roughness_image(rng, texture_size), inline, srgb=False)
        out.append(f'Texture albedo{i} : Image {{ file {{ "{fa}" }} {ea} uv_scale {{ {1 + i % 2}, {1 + i // 2} }} }}\n')
        out.append(f'Texture rough{i} : Image {{ file {{ "{fr}" }} {er} }}\n')
    palette = {"disney": [], "plastic": [], "matte": []}
    for i in range(8):
        met, cc = rng.uniform(0, 0.8), rng.uniform(0, 1)
        out.append(f"Surface disney{i} : Disney {{ color {{ @albedo{i % 4} }} roughness {{ @rough{(i + 1) % 4} }} metallic : {_const(met)} "
                   f"clearcoat : {_const(cc)} sheen : {_const(rng.uniform(0, 0.5))} eta : {_const(1.5)} }}\n")
        palette["disney"].append(f"disney{i}")
    for i in range(6):
        out.append(f"Surface plastic{i} : Plastic {{ Kd {{ @albedo{i % 4} }} roughness {{ @rough{i % 4} }} eta : {_const(1.5)} }}\n")
        palette["plastic"].append(f"plastic{i}")
    for i in range(6):
        kd = f"Kd {{ @albedo{i % 4} }}" if i < 4 else f"Kd : {_const(rng.uniform(0.2, 0.8, 3))}"
        out.append(f"Surface matte{i} : Matte {{ {kd} }}\n")
        palette["matte"].append(f"matte{i}")
    out.append("Surface wall : Matte { Kd { @albedo0 } }\n")
    out.append("Surface floor_s : Plastic { Kd { @albedo1 } roughness { @rough2 } eta : Constant { v { 1.5 } } }\n")
    tri_counts = _write_meshes(out_dir, out, _uv_meshes(levels=mesh_levels, torus_res=torus_res, box_n=box_n), inline)
    _room(out, shapes, open_windows=True, lamps=True)
    classes = ["disney", "plastic", "matte"]
    probs = np.array([0.45, 0.25, 0.30])

    def pick(r):
        c = classes[r.choice(len(classes), p=probs)]
        return palette[c][r.integers(len(palette[c]))]

    _scatter(out, shapes, rng, tri_counts, target_triangles, pick)
    env = f'  environment : Spherical {{ emission : Image {{ file {{ "{sky}" }} }} scale {{ 0.5 }} }}\n'
    _tail(out, shapes, env, resolution, spp, depth, seed, sampler, file, camera="ThinLens",
          camera_extra="aperture { 4 } focal_length { 35 } focus_distance { 2.5 }")
    path = os.path.join(out_dir, f"{name}.luisa")
    with open(path, "w") as f:
        f.write("".join(out))
    return path


def generate_kitchen_scene(out_dir, target_triangles=600_000, resolution=(1280, 720), spp=65536, depth=16, seed=19980810,
                           sampler="Independent", file="render.exr", name="kitchen", inline=False, mesh_levels=(3, 4), torus_res=(48, 24), box_n=8,
                           environment="Spherical { emission : Constant { v { 0.05, 0.06, 0.08 } } }"):
    """C5: the full surface closure set of SURVEY row a14 + the NormalMap / Opacity wrappers + image textures."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    out, shapes = [], []
    fa, ea = _image(out_dir, "albedo", albedo_image(rng, 1024), inline, srgb=True)
    fr, er = _image(out_dir, "rough", roughness_image(rng, 1024), inline, srgb=False)
    fn, en = _image(out_dir, "normal", normal_image(rng, 512), inline, srgb=False)
    out.append(f'Texture albedo : Image {{ file {{ "{fa}" }} {ea} }}\n')
    out.append(f'Texture rough : Image {{ file {{ "{fr}" }} {er} }}\n')
    out.append(f'Texture nmap : Image {{ file {{ "{fn}" }} {en} }}\n')
    out.append("Texture chk : Checkerboard { on : Constant { v { 1 } } off : Constant { v { 0.15 } } scale { 6 } }\n")
    palette = []
    for i in range(4):
        out.append(f"Surface matte{i} : Matte {{ Kd : {_const(rng.uniform(0.15, 0.85, 3))} }}\n")
        palette.append((f"matte{i}", 0.30 / 4))
    out.append(f"Surface oren : Matte {{ Kd {{ @albedo }} sigma : {_const(0.5)} }}\n")
    palette.append(("oren", 0.08))
    out.append("Surface mirror0 : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } }\n")
    out.append(f"Surface mirror1 : Mirror {{ color : {_const([0.8, 0.85, 0.9])} roughness : {_const(0.15)} }}\n")
    palette += [("mirror0", 0.03), ("mirror1", 0.03)]
    out.append("Surface glass0 : Glass { eta : Constant { v { 1.5 } } }\n")
    out.append(f"Surface glass1 : Glass {{ roughness : {_const(0.12)} eta : Constant {{ v {{ 1.51, 1.52, 1.54 }} }} }}\n")
    palette += [("glass0", 0.04), ("glass1", 0.04)]
    out.append(f"Surface plastic0 : Plastic {{ Kd {{ @albedo }} roughness {{ @rough }} eta : {_const(1.5)} }}\n")
    out.append(f"Surface plastic1 : Plastic {{ Kd : {_const([0.6, 0.2, 0.15])} roughness : {_const(0.1)} sigma_a : {_const([0.1, 0.3, 0.4])} thickness : {_const(0.4)} eta : {_const(1.5)} }}\n")
    palette += [("plastic0", 0.06), ("plastic1", 0.06)]
    for i, m in enumerate(["Al", "Cu", "Au"]):
        out.append(f'Surface metal{i} : Metal {{ eta {{ "{m}" }} roughness : {_const(rng.uniform(0.08, 0.35))} }}\n')
        palette.append((f"metal{i}", 0.03))
    out.append(f"Surface disney0 : Disney {{ color {{ @albedo }} roughness {{ @rough }} metallic : {_const(0.4)} clearcoat : {_const(0.7)} "
               f"sheen : {_const(0.3)} anisotropic : {_const(0.3)} eta : {_const(1.5)} }}\n")
    out.append(f"Surface disney1 : Disney {{ color : {_const([0.8, 0.85, 0.9])} roughness : {_const(0.25)} specular_trans : {_const(0.75)} eta : {_const(1.45)} }}\n")
    out.append(f"Surface disney2 : Disney {{ thin {{ true }} color : {_const([0.6, 0.7, 0.5])} roughness : {_const(0.35)} specular_trans : {_const(0.4)} "
               f"diffuse_trans : {_const(0.8)} flatness : {_const(0.5)} eta : {_const(1.3)} }}\n")
    palette += [("disney0", 0.06), ("disney1", 0.03), ("disney2", 0.03)]
    out.append(f"Surface mix0 : Mix {{ a {{ @matte0 }} b {{ @mirror1 }} ratio {{ @chk }} }}\n")
    out.append(f"Surface mix1 : Mix {{ a {{ @glass1 }} b {{ @disney0 }} ratio : {_const(0.6)} }}\n")
    palette += [("mix0", 0.03), ("mix1", 0.03)]
    out.append(f"Surface lay_t : Glass {{ Kr : {_const(1.0)} Kt : {_const(1.0)} roughness : {_const(0.15)} eta : {_const(1.5)} }}\n")
    out.append(f"Surface layered0 : Layered {{ top {{ @lay_t }} bottom {{ @matte1 }} thickness : {_const(0.05)} }}\n")
    out.append(f"Surface layered1 : Layered {{ top {{ @lay_t }} bottom {{ @metal2 }} thickness : {_const(0.3)} g : {_const(0.4)} "
               f"albedo : {_const([0.8, 0.6, 0.4])} max_depth {{ 8 }} samples {{ 1 }} }}\n")
    palette += [("layered0", 0.02), ("layered1", 0.02)]
    out.append(f"Surface bumpy : Plastic {{ Kd : {_const([0.3, 0.5, 0.7])} roughness : {_const(0.2)} eta : {_const(1.5)} normal_map {{ @nmap }} }}\n")
    out.append(f"Surface lace : Matte {{ Kd : {_const([0.8, 0.75, 0.6])} alpha {{ @chk }} }}\n")
    palette += [("bumpy", 0.04), ("lace", 0.02)]
    out.append("Surface wall : Matte { Kd : Constant { v { 0.75, 0.73, 0.7 } } }\n")
    out.append("Surface floor_s : Plastic { Kd { @albedo } roughness : Constant { v { 0.25 } } eta : Constant { v { 1.5 } } }\n")
    tri_counts = _write_meshes(out_dir, out, _uv_meshes(levels=mesh_levels, torus_res=torus_res, box_n=box_n), inline)
    _room(out, shapes, open_windows=False, lamps=True)
    names = [n for n, _ in palette]
    probs = np.array([p for _, p in palette])
    probs /= probs.sum()
    _scatter(out, shapes, rng, tri_counts, target_triangles, lambda r: names[r.choice(len(names), p=probs)])
    # (environment=None: a constant Spherical is the one node the reference's own code cannot be asked about -- it dereferences an
    # empty optional there, src/environments/spherical.cpp:97-105 -- so the reference-pinned reduced scene goes without)
    env = f"  environment : {environment}\n" if environment else ""
    _tail(out, shapes, env, resolution, spp, depth, seed, sampler, file)
    path = os.path.join(out_dir, f"{name}.luisa")
    with open(path, "w") as f:
        f.write("".join(out))
    return path
