"""C ABI surface: struct layouts match between C and the ctypes mirrors, and the shared libraries
export every symbol their header declares (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

from luisarender_amd import _ffi
from oracle.check import oracle_lib


def _declared(header, prefix):
    text = open(os.path.join(_ffi.REPO_ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(rf"\b({prefix}_[a-z0-9_]+)\s*\(", text)))


def test_struct_layouts_match():
    lib = _ffi.host_lib()
    for name, st in _ffi.STRUCTS.items():
        assert C.sizeof(st) == lib.lrhost_sizeof(name.encode()), name


def test_host_library_exports_header_symbols():
    lib = _ffi.host_lib()
    names = _declared("lrhost.h", "lrhost")
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_hip_library_exports_header_symbols():
    path = os.path.join(_ffi.LIB_DIR, "liblrhip.so")
    assert os.path.exists(path), "liblrhip.so missing: run __graft_entry__.build()"
    lib = C.CDLL(path)  # loads without a GPU; nothing is called
    names = _declared("lrhip.h", "lrhip")
    assert {"lrhip_create", "lrhip_upload_scene", "lrhip_render", "lrhip_film_download", "lrhip_destroy"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n


def test_oracle_exports():
    lib = oracle_lib()
    for n in _declared("../oracle/oracle.h", "oracle"):
        assert hasattr(lib, n), n


def test_plugin_exports_reference_plugin_abi():
    path = os.path.join(_ffi.REPO_ROOT, "luisarender_amd", "bin", "libluisa-render-integrator-megapath.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", _ffi.REPO_ROOT, "cli"], stdout=subprocess.DEVNULL)
    _ffi.host_lib()
    lib = C.CDLL(path)
    assert hasattr(lib, "create") and hasattr(lib, "destroy")  # scene_node.h:58-67
