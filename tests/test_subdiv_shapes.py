"""Sphere (src/shapes/sphere.cpp) and LoopSubdiv (src/shapes/loop_subdiv.cpp, src/util/loop_subdiv.cpp) shapes: closed forms
of the subdivision (counts, watertightness, the limit surface of an icosahedron, boundary rules on an open grid) and a render."""
import numpy as np
import pytest

from luisarender_amd import Scene
from oracle.check import Oracle

SPHERE = """
Shape ball : Sphere { subdivision { LEVEL } surface : Matte { Kd : Constant { v { 0.8 } } } transform : SRT { scale { 2, 2, 2 } translate { 0, 1, 0 } } }
Camera cam : Pinhole { fov { 40 } spp { 16 } film : Color { resolution { 32, 32 } } position { 0, 1, 9 } look_at { 0, 1, 0 } }
render { cameras { @cam } shapes { @ball } environment : Spherical { emission : Constant { v { 1 } } } integrator : MegaPath { depth { 6 } } }
"""


def _mesh(sc, m=0):
    v = sc.view()
    mesh = v.meshes[m]
    verts = np.array([[v.vertices[mesh.vertex_offset + i].px, v.vertices[mesh.vertex_offset + i].py, v.vertices[mesh.vertex_offset + i].pz,
                       v.vertices[mesh.vertex_offset + i].nx, v.vertices[mesh.vertex_offset + i].ny, v.vertices[mesh.vertex_offset + i].nz,
                       v.vertices[mesh.vertex_offset + i].u, v.vertices[mesh.vertex_offset + i].v] for i in range(mesh.vertex_count)])
    tris = np.array([[v.triangles[mesh.triangle_offset + i].i0, v.triangles[mesh.triangle_offset + i].i1, v.triangles[mesh.triangle_offset + i].i2]
                     for i in range(mesh.triangle_count)])
    return verts, tris


def _edges(tris):
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    return e


@pytest.mark.parametrize("level", [0, 1, 3])
def test_sphere_is_a_watertight_unit_icosphere(level):
    sc = Scene.from_string(SPHERE.replace("LEVEL", str(level)), build_accel=False)
    verts, tris = _mesh(sc)
    assert len(tris) == 20 * 4 ** level and len(verts) == 10 * 4 ** level + 2  # Euler: V - E + F = 2
    p, n, uv = verts[:, :3], verts[:, 3:6], verts[:, 6:]
    assert np.allclose(np.linalg.norm(p, axis=1), 1, atol=1e-6) and np.allclose(n, p, atol=1e-7)  # pushed to the unit sphere
    assert np.isfinite(uv).all() and ((uv >= 0) & (uv <= 1)).all()  # fract(-tiny) rounds to 1.0f
    e = _edges(tris)
    directed = {(a, b) for a, b in e.tolist()}
    assert len(directed) == len(e) and all((b, a) in directed for a, b in directed)  # every edge twice, opposite ways
    # outward orientation: the face normal points away from the centre
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    assert (np.einsum("ij,ij->i", fn, p[tris].mean(axis=1)) > 0).all()
    # area converges to 4 pi from below
    area = 0.5 * np.linalg.norm(fn, axis=1).sum()
    assert area < 4 * np.pi and area > 4 * np.pi * {0: 0.75, 1: 0.92, 3: 0.99}[level]


def test_loop_subdivision_rules_on_an_open_grid():
    """a flat 2 x 2 grid of quads (8 triangles, boundary all around): the limit surface stays in the plane, the boundary stays on
    the boundary, corners follow the boundary rule, every limit normal is the plane normal"""
    pos = [(x, y, 0) for y in range(3) for x in range(3)]
    idx = []
    for y in range(2):
        for x in range(2):
            a, b, c, d = y * 3 + x, y * 3 + x + 1, (y + 1) * 3 + x + 1, (y + 1) * 3 + x
            idx += [a, b, c, a, c, d]
    text = f"""
Shape grid : InlineMesh {{ positions {{ {", ".join(str(c) for p in pos for c in p)} }} indices {{ {", ".join(map(str, idx))} }} }}
Shape smooth : LoopSubdiv {{ mesh {{ @grid }} level {{ 2 }} surface : Matte {{ }} }}
Camera cam : Pinhole {{ spp {{ 1 }} film : Color {{ resolution {{ 4, 4 }} }} position {{ 1, 1, 5 }} look_at {{ 1, 1, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @smooth }} environment : Spherical {{ emission : Constant {{ v {{ 1 }} }} }} integrator : MegaPath {{ }} }}
"""
    sc = Scene.from_string(text, build_accel=False)
    verts, tris = _mesh(sc)
    assert len(tris) == 8 * 16 and len(verts) == 81  # 9 x 9 grid points
    p, n = verts[:, :3], verts[:, 3:6]
    assert np.allclose(p[:, 2], 0, atol=1e-7) and np.allclose(np.abs(n), [[0, 0, 1]] * len(n), atol=1e-6)
    assert p[:, 0].min() > -1e-6 and p[:, 0].max() < 2 + 1e-6  # the boundary curve shrinks at the corners, never grows
    # the regular interior vertex keeps its place (symmetric stencil), the 4 corners moved inwards along the diagonal
    assert np.any(np.all(np.isclose(p[:, :2], [1, 1], atol=1e-6), axis=1))
    corner = p[np.argmin(p[:, 0] + p[:, 1])]
    assert 0 < corner[0] < 0.5 and corner[0] == pytest.approx(corner[1], abs=1e-6)
    v = sc.view()
    assert v.instances[0].handle.x & 0x3ff  # vertex-normal flag set: the subdivided mesh carries normals


def test_sphere_renders_like_a_ball_under_a_white_sky():
    """white furnace: a grey ball under a constant white environment has radiance <= 1 everywhere, 1 on the background, and its
    silhouette covers the disc a unit-2 sphere projects to"""
    sc = Scene.from_string(SPHERE.replace("LEVEL", "4"))
    film, _ = Oracle(sc).render(0, 16)
    img = film[..., :3] / film[..., 3:4]
    assert np.isfinite(img).all() and img.max() < 1.5 and img.mean() < 1.0  # (16 spp: single pixels scatter around their mean)
    assert img[0, 0, 0] == pytest.approx(1.0, abs=1e-5)  # background
    centre = img[14:18, 14:18].mean()
    assert 0.5 < centre < 0.98  # rho / (1 - ...) < 1: darker than the sky, lit from everywhere
    covered = (img[..., 0] < 0.999).mean()
    # the ball of radius 2 at distance 9 subtends asin(2/9); the 40 degree fov spans 32 px
    r_px = np.tan(np.arcsin(2 / 9)) / np.tan(np.radians(20)) * 16
    assert covered == pytest.approx(np.pi * r_px ** 2 / 32 ** 2, rel=0.12)


# ------------------------------------------------------------------ Mesh { subdivision }: Catmull-Clark (csrc/host/catmull_clark.cpp)
CUBE_OBJ = """
v -1 -1 -1
v  1 -1 -1
v  1  1 -1
v -1  1 -1
v -1 -1  1
v  1 -1  1
v  1  1  1
v -1  1  1
f 1 4 3 2
f 5 6 7 8
f 1 2 6 5
f 2 3 7 6
f 3 4 8 7
f 4 1 5 8
"""
MESH = """
Shape m : Mesh {{ file {{ "{file}" }} subdivision {{ {level} }} {flags} surface : Matte {{ Kd : Constant {{ v {{ 0.8 }} }} }} }}
Camera cam : Pinhole {{ fov {{ 40 }} spp {{ 16 }} film : Color {{ resolution {{ 32, 32 }} }} position {{ 0, 0, 9 }} look_at {{ 0, 0, 0 }} }}
render {{ cameras {{ @cam }} shapes {{ @m }} environment : Spherical {{ emission : Constant {{ v {{ 1 }} }} }} integrator : MegaPath {{ depth {{ 6 }} }} }}
"""


def _cc(tmp_path, obj, level, flags="drop_normal { true } drop_uv { true }"):
    path = tmp_path / "m.obj"
    path.write_text(obj)
    return Scene.from_string(MESH.format(file=str(path), level=level, flags=flags), build_accel=False)


def test_catmull_clark_first_level_of_a_cube(tmp_path):
    """closed form: 6 face points at the face centres, 12 edge points (2 end points + 2 face points) / 4 = 3/4 of the edge midpoint,
    8 vertex points (F + 2 R + (3 - 3) P) / 3 with F = 1/3 P-ish ... = 5/9 of the corner; 24 quads = 48 triangles"""
    verts, tris = _mesh(_cc(tmp_path, CUBE_OBJ, 1))
    p = verts[:, :3]
    assert len(tris) == 48 and len(verts) == 26  # V - E + F = 26 - 48 + 24 = 2, shared vertices
    kinds = {round(float(np.abs(q).sum()), 5): 0 for q in p}
    for q in p:
        kinds[round(float(np.abs(q).sum()), 5)] += 1
    # face points (+-1, 0, 0): |.|_1 = 1; edge points (+-3/4, +-3/4, 0): 1.5; vertex points 5/9 (+-1, +-1, +-1): 5/3
    assert kinds == {1.0: 6, 1.5: 12, round(5 / 3, 5): 8}, kinds
    e = _edges(tris)
    directed = {(a, b) for a, b in e.tolist()}
    assert len(directed) == len(e)  # consistently oriented
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    assert (np.einsum("ij,ij->i", fn, p[tris].mean(axis=1)) > 0).all()  # outward, like the input faces


def test_catmull_clark_converges_to_the_limit_surface_of_the_cube(tmp_path):
    areas, radii = [], []
    for level in (1, 2, 3, 4):
        verts, tris = _mesh(_cc(tmp_path, CUBE_OBJ, level))
        p = verts[:, :3]
        assert len(tris) == 12 * 4 ** level and len(verts) == 6 * 4 ** level + 2
        fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
        areas.append(0.5 * np.linalg.norm(fn, axis=1).sum())
        r = np.linalg.norm(p, axis=1)
        radii.append((r.min(), r.max()))
    # the control mesh shrinks onto a rounded, nearly spherical limit surface: areas decrease and settle, the cubic symmetry stays
    assert all(a > b for a, b in zip(areas, areas[1:])) and areas[-2] - areas[-1] < 0.02 * areas[-1]
    assert radii[-1][1] / radii[-1][0] < 1.05 and 0.8 < radii[-1][0] < radii[-1][1] < 0.9  # a slightly cubic ball of radius ~0.85
    verts, _ = _mesh(_cc(tmp_path, CUBE_OBJ, 3))
    q = np.abs(verts[:, :3])
    assert np.allclose(np.sort(q, axis=1)[np.lexsort(np.sort(q, axis=1).T)], np.sort(q[:, [1, 2, 0]], axis=1)[np.lexsort(np.sort(q[:, [1, 2, 0]], axis=1).T)], atol=1e-6)


def test_catmull_clark_open_grid_keeps_its_plane_boundary_and_uvs(tmp_path):
    """a flat 2 x 2 grid of quads with uvs = xy / 2: stays flat, boundary vertices stay, uv stays the same affine function of the
    position (every attribute goes through the same weights), normals stay the plane normal"""
    lines = [f"v {x} {y} 0" for y in range(3) for x in range(3)] + [f"vt {x / 2} {y / 2}" for y in range(3) for x in range(3)]
    for y in range(2):
        for x in range(2):
            a, b, c, d = y * 3 + x + 1, y * 3 + x + 2, (y + 1) * 3 + x + 2, (y + 1) * 3 + x + 1
            lines.append(f"f {a}/{a} {b}/{b} {c}/{c} {d}/{d}")
    verts, tris = _mesh(_cc(tmp_path, "\n".join(lines), 2, flags="flip_uv { true }"))
    p, n, uv = verts[:, :3], verts[:, 3:6], verts[:, 6:]
    assert len(tris) == 4 * 16 * 2
    assert np.allclose(p[:, 2], 0) and np.allclose(n, [0, 0, 1], atol=1e-6)
    assert p[:, :2].min() == 0 and p[:, :2].max() == 2  # the four corners did not move
    assert np.allclose(uv, p[:, :2] / 2, atol=1e-6)
    area = 0.5 * np.linalg.norm(np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]]), axis=1).sum()
    assert area <= 4.0 + 1e-5  # inside the original square


def test_catmull_clark_handles_triangles_and_ngons_and_renders(tmp_path):
    """a tetrahedron (triangles -> 3 quads each) and a render through the oracle: the subdivided solid is lit and closed"""
    tet = "v 1 1 1\nv -1 -1 1\nv -1 1 -1\nv 1 -1 -1\nf 1 2 3\nf 1 4 2\nf 1 3 4\nf 2 4 3\n"
    sc = _cc(tmp_path, tet, 2, flags="")
    verts, tris = _mesh(sc)
    assert len(tris) == 4 * 3 * 4 * 2 and np.allclose(np.linalg.norm(verts[:, 3:6], axis=1), 1, atol=1e-5)
    e = _edges(tris)
    directed = {(a, b) for a, b in e.tolist()}
    assert all((b, a) in directed for a, b in directed)  # watertight
    path = tmp_path / "m.obj"
    sc = Scene.from_string(MESH.format(file=str(path), level=2, flags=""))
    o = Oracle(sc)
    film, _ = o.render(0, 16)
    img = o.convert(film)[..., :3]
    assert img[12:20, 12:20].mean() < 0.95 * img[0:4, 0:4].mean() and np.isfinite(img).all()  # the solid shades, the sky is 1
    with pytest.raises(Exception, match="out of range"):
        _cc(tmp_path, tet, 9)


def test_catmull_clark_joins_topology_by_position_across_uv_seams(tmp_path):
    """a cube whose six faces carry their own uvs (24 distinct corners, 8 positions): the subdivider's adjacency goes by position, so
    the surface stays closed and lands on the same 26 points as the cube without uvs.  Every attribute goes through the same
    weights over that adjacency, so uvs are blended ACROSS the seams (the restated behaviour of assimp's subdivider, unpinned):
    corners that share a position come out with one uv and are joined again"""
    lines = [l for l in CUBE_OBJ.strip().splitlines() if l.startswith("v ")] + ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1"]
    for l in [l for l in CUBE_OBJ.strip().splitlines() if l.startswith("f ")]:
        idx = l.split()[1:]
        lines.append("f " + " ".join(f"{v}/{k + 1}" for k, v in enumerate(idx)))
    verts, tris = _mesh(_cc(tmp_path, "\n".join(lines), 1, flags="drop_normal { true } flip_uv { true }"))
    plain, _ = _mesh(_cc(tmp_path, CUBE_OBJ, 1))
    pos = {tuple(np.round(p, 5)) for p in verts[:, :3]}
    assert pos == {tuple(np.round(p, 5)) for p in plain[:, :3]} and len(pos) == 26
    assert len(tris) == 48
    by_pos = {}
    for a, b in _edges(tris).tolist():  # watertight BY POSITION: every edge has its opposite
        key = (tuple(np.round(verts[a, :3], 5)), tuple(np.round(verts[b, :3], 5)))
        by_pos[key] = by_pos.get(key, 0) + 1
    assert all(by_pos.get((b, a), 0) == n for (a, b), n in by_pos.items())
    uv = verts[:, 6:]
    assert uv.min() >= -1e-6 and uv.max() <= 1 + 1e-6 and np.isfinite(verts).all()
