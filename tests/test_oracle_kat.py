"""Pins for the CPU oracle: bit-exact known-answer tests of the hash/RNG functions against
INDEPENDENT implementations, alias-table and filter statistics, handle packing round trips.

The reference ships no golden vectors for this path, so these are the pins:
  * the reference's xxhash32 variants (src/util/rng.cpp:12-69) are XXH32 of the leading words with
    seed = last word - 4*len(leading words) (the 1-word form hashes a zero word)  -> checked against the
    python `xxhash` package;
  * PCG32 (rng.cpp:142-176) -> checked against the published pcg32 demo vector (seed 42, seq 54);
  * the LCG (rng.cpp:132-140) -> checked against the Numerical Recipes constants in plain python.
"""
import ctypes as C
import struct

import numpy as np
import pytest
import xxhash

from luisarender_amd import Scene, _ffi
from oracle.check import oracle_lib
from luisarender_amd.scenes import cornell_box

O = oracle_lib()


def _xxh32_words(words, seed):
    return xxhash.xxh32(struct.pack(f"<{len(words)}I", *words), seed=seed & 0xffffffff).intdigest()


def test_xxhash32_variants_against_python_xxhash():
    rng = np.random.default_rng(1)
    vals = [0, 1, 0xffffffff, 19980810, 0x80000000] + [int(v) for v in rng.integers(0, 2 ** 32, 200, dtype=np.uint64)]
    for i in range(0, len(vals) - 4):
        x, y, z, w = vals[i:i + 4]
        assert O.oracle_xxhash32_1(x) == _xxh32_words([0], x - 4)
        assert O.oracle_xxhash32_2(x, y) == _xxh32_words([x], y - 4)
        assert O.oracle_xxhash32_3(x, y, z) == _xxh32_words([x, y], z - 8)
        assert O.oracle_xxhash32_4(x, y, z, w) == _xxh32_words([x, y, z], w - 12)


def test_lcg_stream():
    state = C.c_uint32(12345)
    s = 12345
    for _ in range(100):
        s = (1664525 * s + 1013904223) & 0xffffffff
        u = O.oracle_lcg(C.byref(state))
        assert state.value == s
        expect = min(np.float32(s) * np.float32(2.0 ** -32), np.nextafter(np.float32(1), np.float32(0)))
        assert np.float32(u) == np.float32(expect) and 0.0 <= u < 1.0


def test_pcg32_published_vector():
    # pcg32_srandom_r(&rng, 42u, 54u) from the PCG reference demo -> first six outputs
    expected = [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]
    state, inc = C.c_uint64(0), C.c_uint64((54 << 1) | 1)
    O.oracle_pcg32_next(C.byref(state), C.byref(inc))
    state.value = (state.value + 42) & 0xffffffffffffffff
    O.oracle_pcg32_next(C.byref(state), C.byref(inc))
    assert [O.oracle_pcg32_next(C.byref(state), C.byref(inc)) for _ in range(6)] == expected
    # set_sequence(seq) == srandom(default_state, seq) (rng.cpp:150-156)
    s2, i2 = C.c_uint64(), C.c_uint64()
    O.oracle_pcg32_seed(54, C.byref(s2), C.byref(i2))
    state, inc = C.c_uint64(0), C.c_uint64((54 << 1) | 1)
    O.oracle_pcg32_next(C.byref(state), C.byref(inc))
    state.value = (state.value + 0x853c49e6748fea9b) & 0xffffffffffffffff
    O.oracle_pcg32_next(C.byref(state), C.byref(inc))
    assert (s2.value, i2.value) == (state.value, inc.value)


def _alias(values):
    values = np.asarray(values, np.float32)
    table = (_ffi.AliasEntry * len(values))()
    pdf = np.zeros(len(values), np.float32)
    O.oracle_create_alias_table(values.ctypes.data, len(values), table, pdf.ctypes.data)
    return table, pdf


def test_alias_table_reproduces_distribution():
    # mirrors the reference's src/tests/test_alias_method.cpp (which only prints): 128 random weights
    rng = np.random.default_rng(7)
    values = rng.random(128).astype(np.float32) ** 3
    table, pdf = _alias(values)
    assert abs(pdf.sum() - 1.0) < 1e-5
    # exact check: P(i) = (prob_i + sum_{j: alias_j = i} (1 - prob_j)) / n
    n = len(values)
    p = np.zeros(n)
    for j in range(n):
        p[j] += min(table[j].prob, 1.0) / n
        if table[j].prob < 1.0:
            p[table[j].alias] += (1.0 - table[j].prob) / n
    assert np.abs(p - pdf).max() < 1e-6
    # sampling: empirical frequencies and remapped-u uniformity
    us = (np.arange(200000) + 0.5) / 200000
    idx, ur = C.c_uint32(), C.c_float()
    counts, usum = np.zeros(n), 0.0
    for u in us[::20]:
        O.oracle_sample_alias_table(table, n, float(u), C.byref(idx), C.byref(ur))
        counts[idx.value] += 1
        usum += ur.value
        assert 0.0 <= ur.value <= 1.0
    assert np.abs(counts / counts.sum() - pdf).max() < 2e-3
    assert abs(usum / counts.sum() - 0.5) < 0.01


def test_alias_table_degenerate_inputs():
    table, pdf = _alias([0.0, 0.0, 0.0, 0.0])  # all zero -> uniform pdf (sampling.cpp:44-47)
    assert np.allclose(pdf, 0.25)
    table, pdf = _alias([5.0])
    assert table[0].prob == 1.0 and table[0].alias == 0 and pdf[0] == 1.0


def test_host_tables_equal_oracle_restatement():
    """The host builder (csrc/host/scene.cpp) and the oracle restate create_alias_table independently."""
    sc = Scene.from_string(cornell_box(32, 1))
    v = sc.view()
    for m in range(v.mesh_count):
        mesh = v.meshes[m]
        areas = []
        for t in range(mesh.triangle_count):
            tri = v.triangles[mesh.triangle_offset + t]
            p = [np.array([v.vertices[mesh.vertex_offset + i].px, v.vertices[mesh.vertex_offset + i].py,
                           v.vertices[mesh.vertex_offset + i].pz], np.float32) for i in (tri.i0, tri.i1, tri.i2)]
            areas.append(np.float32(np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32))))
        table, pdf = _alias(areas)
        for t in range(mesh.triangle_count):
            host = v.tri_alias[mesh.triangle_offset + t]
            assert abs(host.prob - table[t].prob) < 1e-6 and host.alias == table[t].alias
            assert abs(v.tri_pdf[mesh.triangle_offset + t] - pdf[t]) < 1e-7
    f = v.filter  # filter LUT alias table (filter.cpp:24-47): box filter -> uniform bins
    assert np.allclose(np.array(f.pdf[:]), 1.0 / 63.0, atol=1e-7)
    assert np.allclose(np.array(f.lut[:]), 1.0 / 63.0, atol=1e-7)


def test_instance_handle_round_trip():
    out = (C.c_uint32 * 4)()
    for base, flags, st, lt, mt, tris, sterm, off in [(0, 0, 0, 0, 0, 1, 0.0, 0.0), (4194303, 63, 4095, 4095, 255, 2 ** 32 - 1, 1.0, 1.0),
                                                       (1234, 0b101101, 77, 3, 9, 5120, 0.25, 0.5)]:
        O.oracle_encode_handle(base, flags, st, lt, mt, tris, sterm, off, out)
        assert out[0] >> 10 == base and out[0] & 1023 == flags
        assert out[1] & 4095 == lt and (out[1] >> 12) & 4095 == st and out[1] >> 24 == mt and out[2] == tris
        assert out[3] >> 16 == min(round(sterm * 65536), 65535) and out[3] & 0xffff == min(round(off * 65536), 65535)
    sc = Scene.from_string(cornell_box(32, 1))  # host encoder agrees with the oracle's restatement
    h = sc.view().instances[5].handle
    O.oracle_encode_handle(h.x >> 10, h.x & 1023, (h.y >> 12) & 4095, h.y & 4095, h.y >> 24, h.z, 0.0, 0.0, out)
    assert list(out) == [h.x, h.y, h.z, h.w]


@pytest.mark.parametrize("impl,radius", [("Box", 0.5), ("Gaussian", 1.0), ("Triangle", 1.5), ("Mitchell", 2.0), ("LanczosSinc", 1.0)])
def test_filter_importance_sampling(impl, radius):
    sc = Scene.from_string(cornell_box(32, 1, filter_impl=impl, filter_radius=radius))
    f = sc.view().filter
    out = np.zeros(3, np.float32)
    rng = np.random.default_rng(3)
    w = []
    for ux, uy in rng.random((4000, 2)):
        O.oracle_filter_sample(C.byref(f), float(ux), float(uy), out.ctypes.data)
        assert abs(out[0]) <= radius * 1.0001 and abs(out[1]) <= radius * 1.0001
        w.append(out[2])
    w = np.array(w)
    if impl in ("Box", "Gaussian", "Triangle"):
        assert (w > 0).all() and w.std() / w.mean() < 0.3  # positive filters: importance sampling flattens the weight
    assert abs(w.mean() - 1.0) < 0.12  # weights average to the normalised integral


def test_offset_ray_origin_moves_along_normal():
    out = np.zeros(3, np.float32)
    for p, n in [((10.0, -3.0, 0.5), (0.0, 1.0, 0.0)), ((0.001, 0.002, -0.003), (0.0, 0.0, -1.0)), ((552.8, 0.0, 559.2), (-1.0, 0.0, 0.0))]:
        pa, na = np.array(p, np.float32), np.array(n, np.float32)
        O.oracle_offset_ray_origin(pa.ctypes.data, na.ctypes.data, out.ctypes.data)
        d = out - pa
        assert np.dot(d, na) > 0 and np.linalg.norm(d) < 1e-3 * max(1.0, np.abs(pa).max())
