"""Procedural stand-ins for the README benchmark scenes (assets are external downloads that are not
available offline): "Bathroom-class" (BASELINE config C2) and friends, per SURVEY §8(d).

A closed 4 x 3 x 4 m room filled with instanced, procedurally tessellated fixtures (icospheres,
tori, tessellated boxes written as Wavefront OBJ), a material mix by object count of about
60 % Matte, 20 % Plastic (eta 1.5, roughness U[0.05, 0.4]), 8 % Metal (Al, Cu), 7 % Glass
(eta 1.5, roughness 0 / 0.1), 5 % Mirror, and three one-sided Diffuse area lights under the
ceiling.  Every generator is seeded (default 19980810) and the output is a scene in the reference's
text grammar that uses Mesh / Instance / SRT nodes exactly like converter output
(tools/tungsten2luisa.py in the reference).
"""
from __future__ import annotations

import os

import numpy as np


def _write_obj(path, v, f, n=None):
    with open(path, "w") as out:
        for p in v:
            out.write(f"v {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
        if n is not None:
            for p in n:
                out.write(f"vn {p[0]:.7g} {p[1]:.7g} {p[2]:.7g}\n")
            for t in f + 1:
                out.write(f"f {t[0]}//{t[0]} {t[1]}//{t[1]} {t[2]}//{t[2]}\n")
        else:
            for t in f + 1:
                out.write(f"f {t[0]} {t[1]} {t[2]}\n")


def icosphere(level):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(level):
        cache = {}
        verts = list(v)

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (verts[a] + verts[b]) * 0.5
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf, np.int64)
    return v, f, v.copy()


def torus(nu, nv, r_major=1.0, r_minor=0.35):
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    w = np.linspace(0, 2 * np.pi, nv, endpoint=False)
    uu, ww = np.meshgrid(u, w, indexing="ij")
    x = (r_major + r_minor * np.cos(ww)) * np.cos(uu)
    z = (r_major + r_minor * np.cos(ww)) * np.sin(uu)
    y = r_minor * np.sin(ww)
    v = np.stack([x, y, z], -1).reshape(-1, 3)
    n = np.stack([np.cos(ww) * np.cos(uu), np.sin(ww), np.cos(ww) * np.sin(uu)], -1).reshape(-1, 3)
    idx = np.arange(nu * nv).reshape(nu, nv)
    a, b = idx, np.roll(idx, -1, 0)
    c, d = np.roll(idx, -1, 1), np.roll(np.roll(idx, -1, 0), -1, 1)
    f = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)])
    return v, f, n


def tess_box(n):
    """unit cube [-1, 1]^3, each face an n x n grid (flat shading: no vertex normals written)"""
    vs, fs = [], []
    g = np.linspace(-1, 1, n + 1)
    a, b = np.meshgrid(g, g, indexing="ij")
    idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
    quad = np.concatenate([np.stack([idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:]], -1).reshape(-1, 3),
                           np.stack([idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)])
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            p = np.zeros(((n + 1) ** 2, 3))
            p[:, axis] = sgn
            p[:, (axis + 1) % 3] = a.reshape(-1)
            p[:, (axis + 2) % 3] = b.reshape(-1)
            q = quad if sgn > 0 else quad[:, ::-1]
            fs.append(q + len(vs) * (n + 1) ** 2)
            vs.append(p)
    return np.concatenate(vs), np.concatenate(fs), None


def inline_mesh(name, v, f, n=None, uv=None):
    """the same mesh as an InlineMesh node (src/shapes/inline_mesh.cpp: positions / indices / normals / uvs) instead of a
    Wavefront file behind `Mesh`: the reference's own plugins can load it without assimp (tests/test_oracle_vs_ref.py)"""
    num = lambda a: ", ".join(f"{x:.7g}" for x in np.asarray(a, np.float64).reshape(-1))
    s = f"Shape {name} : InlineMesh {{\n  positions {{ {num(v)} }}\n  indices {{ {', '.join(str(int(i)) for i in np.asarray(f).reshape(-1))} }}\n"
    if n is not None:
        s += f"  normals {{ {num(n)} }}\n"
    if uv is not None:
        s += f"  uvs {{ {num(uv)} }}\n"
    return s + "}\n"


_QUAD = "0, 1, 2, 0, 2, 3"


def _quad_shape(name, pts, surface=None, light=None):
    pos = ", ".join(f"{v:g}" for p in pts for v in p)
    s = f"Shape {name} : InlineMesh {{\n  positions {{ {pos} }}\n  indices {{ {_QUAD} }}\n"
    if surface:
        s += f"  surface {{ @{surface} }}\n"
    if light:
        s += f"  light : Diffuse {{ emission : Constant {{ v {{ {light} }} }} }}\n"
    return s + "}\n"


def generate_room_scene(out_dir, target_triangles=600_000, resolution=(1024, 1024), spp=1024, depth=16, seed=19980810,
                        glass_fraction=0.07, environment=None, open_windows=False, file="render.exr",
                        sampler="Independent", name="bathroom", inline_meshes=False,
                        mesh_levels=(3, 4), torus_res=(48, 24), box_n=8, bake_transforms=False):
    """Writes <out_dir>/<name>.luisa (+ OBJ meshes) and returns its path.  inline_meshes: the fixtures' meshes as InlineMesh nodes
    instead of OBJ files (same instances, transforms and materials; the form the reference's own code can load here).
    bake_transforms: every fixture is its own InlineMesh whose vertices and normals already are in world space (scale, rotation and
    translation applied here, in double precision) and carries no transform node: the same room with object space = world space
    (tests/test_gpu_parity.py: what separates the device from the oracle on this scene is the instance transform's rounding)."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    meshes = {
        f"ico{mesh_levels[0]}": icosphere(mesh_levels[0]), f"ico{mesh_levels[1]}": icosphere(mesh_levels[1]),
        "torus": torus(*torus_res), f"box{box_n}": tess_box(box_n),
    }
    tri_counts = {k: len(m[1]) for k, m in meshes.items()}
    if not inline_meshes and not bake_transforms:
        for k, (v, f, n) in meshes.items():
            _write_obj(os.path.join(out_dir, f"{k}.obj"), v, f, n)
    out = []
    # surfaces: a palette per class
    other = 1.0 - glass_fraction
    classes = [("matte", 0.60 / 0.93 * other), ("plastic", 0.20 / 0.93 * other), ("metal", 0.08 / 0.93 * other),
               ("glass", glass_fraction), ("mirror", 0.05 / 0.93 * other)]
    surf_names = {c: [] for c, _ in classes}
    for i in range(12):
        kd = rng.uniform(0.15, 0.85, 3)
        out.append(f"Surface matte{i} : Matte {{ Kd : Constant {{ v {{ {kd[0]:.4f}, {kd[1]:.4f}, {kd[2]:.4f} }} }} }}\n")
        surf_names["matte"].append(f"matte{i}")
    for i in range(6):
        kd = rng.uniform(0.1, 0.8, 3)
        r = rng.uniform(0.05, 0.4)
        out.append(f"Surface plastic{i} : Plastic {{ Kd : Constant {{ v {{ {kd[0]:.4f}, {kd[1]:.4f}, {kd[2]:.4f} }} }} "
                   f"roughness : Constant {{ v {{ {r:.4f} }} }} eta : Constant {{ v {{ 1.5 }} }} }}\n")
        surf_names["plastic"].append(f"plastic{i}")
    for i, m in enumerate(["Al", "Cu", "Au", "Ag"]):
        r = rng.uniform(0.1, 0.4)
        out.append(f'Surface metal{i} : Metal {{ eta {{ "{m}" }} roughness : Constant {{ v {{ {r:.4f} }} }} }}\n')
        surf_names["metal"].append(f"metal{i}")
    for i, r in enumerate([0.0, 0.1]):
        rough = f"roughness : Constant {{ v {{ {r} }} }} " if r > 0 else ""
        out.append(f"Surface glass{i} : Glass {{ {rough}eta : Constant {{ v {{ 1.5 }} }} }}\n")
        surf_names["glass"].append(f"glass{i}")
    out.append("Surface mirror0 : Mirror { color : Constant { v { 0.9, 0.9, 0.9 } } }\n")
    surf_names["mirror"].append("mirror0")
    out.append("Surface wall : Matte { Kd : Constant { v { 0.75, 0.73, 0.7 } } }\n")
    out.append("Surface floor_s : Matte { Kd : Constant { v { 0.4, 0.35, 0.3 } } }\n")
    for k, (v, f, n) in meshes.items():
        if not bake_transforms:
            out.append(inline_mesh(f"mesh_{k}", v, f, n) if inline_meshes else f'Shape mesh_{k} : Mesh {{ file {{ "{k}.obj" }} }}\n')
    # room 4 x 3 x 4
    X, Y, Z = 4.0, 3.0, 4.0
    shapes = []
    out.append(_quad_shape("floor", [(0, 0, 0), (0, 0, Z), (X, 0, Z), (X, 0, 0)], "floor_s"))
    out.append(_quad_shape("ceil", [(0, Y, 0), (X, Y, 0), (X, Y, Z), (0, Y, Z)], "wall"))
    out.append(_quad_shape("wall_back", [(0, 0, 0), (X, 0, 0), (X, Y, 0), (0, Y, 0)], "wall"))
    out.append(_quad_shape("wall_front", [(0, 0, Z), (0, Y, Z), (X, Y, Z), (X, 0, Z)], "wall"))
    out.append(_quad_shape("wall_left", [(0, 0, 0), (0, Y, 0), (0, Y, Z), (0, 0, Z)], "wall"))
    shapes += ["@floor", "@ceil", "@wall_back", "@wall_front", "@wall_left"]
    if not open_windows:
        out.append(_quad_shape("wall_right", [(X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0)], "wall"))
        shapes.append("@wall_right")
    else:  # window openings: the right wall is split into strips leaving two holes
        strips = [[(X, 0, 0), (X, 0, Z), (X, 0.9, Z), (X, 0.9, 0)], [(X, 2.3, 0), (X, 2.3, Z), (X, Y, Z), (X, Y, 0)],
                  [(X, 0.9, 0), (X, 0.9, 0.6), (X, 2.3, 0.6), (X, 2.3, 0)], [(X, 0.9, 1.8), (X, 0.9, 2.2), (X, 2.3, 2.2), (X, 2.3, 1.8)],
                  [(X, 0.9, 3.4), (X, 0.9, Z), (X, 2.3, Z), (X, 2.3, 3.4)]]
        for i, q in enumerate(strips):
            out.append(_quad_shape(f"wall_right{i}", q, "wall"))
            shapes.append(f"@wall_right{i}")
    # three one-sided area lights just under the ceiling, facing down
    for i, (cx, cz) in enumerate([(1.0, 1.0), (3.0, 1.2), (2.0, 3.0)]):
        h, y = 0.35, Y - 0.02
        out.append(_quad_shape(f"lamp{i}", [(cx - h, y, cz - h), (cx + h, y, cz - h), (cx + h, y, cz + h), (cx - h, y, cz + h)],
                               "wall", light="18, 16, 13"))
        shapes.append(f"@lamp{i}")
    # fixtures
    total = 0
    names = list(meshes)
    weights = np.array([0.3, 0.2, 0.25, 0.25])
    probs = np.array([w for _, w in classes])
    probs /= probs.sum()
    i = 0
    while total < target_triangles:
        k = names[rng.choice(len(names), p=weights)]
        cls = classes[rng.choice(len(classes), p=probs)][0]
        surf = surf_names[cls][rng.integers(len(surf_names[cls]))]
        s = rng.uniform(0.05, 0.16)
        x, z = rng.uniform(0.3, X - 0.3), rng.uniform(0.3, Z - 1.2)
        y = rng.choice([s, rng.uniform(0.4, 2.2)], p=[0.6, 0.4])
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0, 360)
        if bake_transforms:
            a = axis / np.linalg.norm(axis)
            th = np.radians(ang)
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)  # Rodrigues
            v, f, n = meshes[k]
            out.append(inline_mesh(f"obj{i}", (s * v) @ R.T + np.array([x, y, z]), f, None if n is None else n @ R.T)[:-2] + f"  surface {{ @{surf} }}\n}}\n")
        else:
            out.append(f"Shape obj{i} : Instance {{ shape {{ @mesh_{k} }} surface {{ @{surf} }} transform : SRT {{ "
                       f"scale {{ {s:.4f}, {s:.4f}, {s:.4f} }} rotate {{ {axis[0]:.4f}, {axis[1]:.4f}, {axis[2]:.4f}, {ang:.2f} }} "
                       f"translate {{ {x:.4f}, {y:.4f}, {z:.4f} }} }} }}\n")
        shapes.append(f"@obj{i}")
        total += tri_counts[k]
        i += 1
    env = ""
    if environment:
        env = f"  environment : Spherical {{ emission : Constant {{ v {{ {environment} }} }} }}\n"
    out.append(f"""Camera cam : Pinhole {{
  fov {{ 55 }}  spp {{ {spp} }}  file {{ "{file}" }}
  film : Color {{ resolution {{ {resolution[0]}, {resolution[1]} }} }}
  filter : Gaussian {{ radius {{ 1 }} }}
  transform : View {{ position {{ 2.0, 1.6, 3.85 }}  front {{ 0, -0.15, -1 }}  up {{ 0, 1, 0 }} }}
}}
render {{
  cameras {{ @cam }}
  shapes {{ {", ".join(shapes)} }}
{env}  integrator : MegaPath {{ depth {{ {depth} }}  sampler : {sampler} {{ seed {{ {seed} }} }} }}
}}
""")
    path = os.path.join(out_dir, f"{name}.luisa")
    with open(path, "w") as f:
        f.write("".join(out))
    return path
