#!/usr/bin/env python3
"""Generate luisarender_amd/data/sobol_tables.bin from first principles.

The reference's Sobol / PaddedSobol samplers (src/samplers/sobol.cpp, padded_sobol.cpp) read three tables
(src/util/sobolmatrices.cpp: SobolMatrices32[1024][52], VdCSobolMatrices[25][52], VdCSobolMatricesInv[26][52])
that pbrt took from L. Gruenschloss' generator.  They are NOT copied here; they are re-derived:

  * SobolMatrices32: Joe & Kuo's "new-joe-kuo-6.21201" primitive polynomials + initial direction numbers (the
    copy that ships with scipy: scipy/stats/_sobol_direction_numbers.npz), standard Sobol recurrence carried
    to 52 columns in 52-bit fixed point, top 32 bits kept;
  * VdC tables for a 2^m x 2^m pixel grid: index bit k moves the packed pixel (x << m | y) by
    c_k = (top m bits of dim-0 column k) << m | (top m bits of dim-1 column k).  VdCSobolMatrices[m-1][j] = c_{2m+j};
    VdCSobolMatricesInv[m-1] = columns of the GF(2) inverse of [c_0 .. c_{2m-1}].

When /root/reference is present (build container) the result is checked word for word against the reference
tables before anything is written.     python tools/gen_sobol_tables.py
"""
import os
import re
import struct
import sys

import numpy as np

NDIM, NCOL = 1024, 52
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "luisarender_amd", "data", "sobol_tables.bin")
REF = "/root/reference/src/util/sobolmatrices.cpp"


def sobol_matrices():
    import scipy.stats
    z = np.load(os.path.join(os.path.dirname(scipy.stats.__file__), "_sobol_direction_numbers.npz"))
    poly, vinit = z["poly"], z["vinit"]
    out = np.zeros((NDIM, NCOL), np.uint64)
    bits = NCOL
    # dimension 0: van der Corput (identity)
    for j in range(NCOL):
        out[0, j] = 1 << (bits - 1 - j)
    for d in range(1, NDIM):
        p = int(poly[d])
        s = p.bit_length() - 1                      # degree
        a = [(p >> (s - k)) & 1 for k in range(1, s)]  # inner coefficients a_1 .. a_{s-1}
        m = [int(v) for v in vinit[d][:s]]
        v = [0] * NCOL
        for j in range(min(s, NCOL)):
            v[j] = m[j] << (bits - 1 - j)
        for j in range(s, NCOL):
            x = v[j - s] ^ (v[j - s] >> s)
            for k in range(1, s):
                if a[k - 1]:
                    x ^= v[j - k]
            v[j] = x
        out[d] = v
    return (out >> np.uint64(bits - 32)).astype(np.uint32)


def vdc_tables(mats):
    vdc = np.zeros((25, NCOL), np.uint64)
    inv = np.zeros((26, NCOL), np.uint64)
    for m in range(1, 27):
        def col(k):
            x = int(mats[0, k]) >> (32 - m) if k < NCOL else 0
            y = int(mats[1, k]) >> (32 - m) if k < NCOL else 0
            return (x << m) | y
        n = 2 * m
        if m <= 25:
            for j in range(NCOL):
                k = n + j
                vdc[m - 1, j] = col(k) if k < NCOL else 0
        # GF(2) inverse of the n x n matrix with columns c_0 .. c_{n-1} (Gauss-Jordan on bit rows)
        if n <= NCOL:
            cols = [col(k) for k in range(n)]
            rows = [sum(((cols[k] >> r) & 1) << k for k in range(n)) | (1 << (n + r)) for r in range(n)]
            for c in range(n):
                piv = next(r for r in range(c, n) if (rows[r] >> c) & 1)
                rows[c], rows[piv] = rows[piv], rows[c]
                for r in range(n):
                    if r != c and (rows[r] >> c) & 1:
                        rows[r] ^= rows[c]
            invrows = [rows[r] >> n for r in range(n)]          # row r of A^-1
            for d in range(n):                                   # column d of A^-1
                inv[m - 1, d] = sum(((invrows[r] >> d) & 1) << r for r in range(n))
    return vdc, inv


def parse_reference():
    text = open(REF).read()
    def block(name):
        body = text[text.index(name):]
        body = body[body.index("{"):]
        depth, end = 0, 0
        for i, ch in enumerate(body):
            depth += ch == "{"
            depth -= ch == "}"
            if depth == 0:
                end = i
                break
        return body[:end + 1]
    m32 = np.array([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", block("SobolMatrices32[NSobolDimensions"))], np.uint32).reshape(NDIM, NCOL)
    def rows(name, count):
        body = block(name)
        out = np.zeros((count, NCOL), np.uint64)
        for i, row in enumerate(re.findall(r"\{[^{}]*\}", body)):
            vals = [int(x[:-3], 16) for x in re.findall(r"0x[0-9a-fA-F]+ULL", row)]
            out[i, :len(vals)] = vals
        return out
    return m32, rows("VdCSobolMatrices[VdCSobolMatrixSize]", 25), rows("VdCSobolMatricesInv[VdCSobolMatrixInvSize]", 26)


def main():
    mats = sobol_matrices()
    vdc, inv = vdc_tables(mats)
    if os.path.exists(REF):
        r32, rvdc, rinv = parse_reference()
        assert np.array_equal(mats, r32), f"SobolMatrices32 mismatch in {np.argwhere(mats != r32)[:5]}"
        assert np.array_equal(vdc, rvdc), f"VdC mismatch at {np.argwhere(vdc != rvdc)[:5]}"
        assert np.array_equal(inv, rinv), f"VdCInv mismatch at {np.argwhere(inv != rinv)[:5]}"
        print("derived tables are identical to the reference tables (1024x52 u32, 25x52 + 26x52 u64)")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "wb") as f:
        f.write(struct.pack("<4sIIII", b"LRSB", NDIM, NCOL, 25, 26))
        f.write(mats.astype("<u4").tobytes())
        f.write(vdc.astype("<u8").tobytes())
        f.write(inv.astype("<u8").tobytes())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
