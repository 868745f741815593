#!/usr/bin/env python3
"""Resident 256-thread blocks per CU of precompiled kernel variants, as the HIP runtime reports them (GPU box):
    python tools/occupancy.py <lib: base | variant name> <mask>...      (heavy kernels: h<mask>)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
path = os.path.join(ROOT, "luisarender_amd", "lib", "liblrhip.so" if name == "base" else f"variants/liblrhip_{name}.so")
lib = C.CDLL(path)
for m in sys.argv[2:]:
    sym = f"lrhip_heavy_occupancy_{m[1:]}" if m.startswith("h") else f"lrhip_variant_occupancy_{m}"
    try:
        fn = getattr(lib, sym)
    except AttributeError:
        print(name, m, "not in this library")
        continue
    n = C.c_int(0)
    rc = fn(C.byref(n))
    print(f"{name} <{m}>: {n.value} blocks per CU (rc {rc})")
