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


def test_work_items_partition_the_sample_range():
    """lrhip_work_items (no device needed): the tapered work items of lrhip_render -- big chunks first, small ones at the end of the
    launch -- must partition [0, spp) exactly, within the 64 partial planes, for every frame size / spp / shard count."""
    import ctypes as C
    lib = C.CDLL(os.path.join(_ffi.REPO_ROOT, "luisarender_amd", "lib", "liblrhip.so"))
    lib.lrhip_work_items.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32 * 4)]
    out = (C.c_uint32 * 4)()
    tapered = 0
    for width, height in ((16, 16), (96, 64), (512, 512), (1024, 1024), (1280, 720), (3840, 2160)):
        for shards in (1, 2, 8, 64):
            for spp in list(range(1, 40)) + [63, 64, 65, 256, 1000, 1024, 4096, 65536]:
                assert lib.lrhip_work_items(width, height, spp, shards, C.byref(out)) == 0
                count, big_count, big, small = out
                assert 1 <= count <= 64 and big_count <= count and big >= 1 and small >= 1, (width, height, spp, shards, list(out))
                covered = 0
                for k in range(count):
                    b = k * big if k < big_count else big_count * big + (k - big_count) * small
                    e = min(b + (big if k < big_count else small), spp)
                    assert min(b, spp) == covered or b >= spp, (width, height, spp, shards, list(out), k)
                    covered = max(covered, e)
                assert covered == spp, (width, height, spp, shards, list(out))
                tapered += big_count < count
    assert tapered > 100  # the taper is what is normally used
    assert lib.lrhip_work_items(0, 16, 1, 1, C.byref(out)) != 0


def test_scheduler_rule_is_a_function_of_the_scene():
    """lrhip_pool_auto_triangles (no device needed): from how many BVH triangles on the automatic scheduler takes the pool kernels -- the
    rule of tools/sched_sweep.py's sweep (profiles/r05j_scheduler_sweep.txt): ~100 thousand triangles, twice that for shallow paths, half of
    it for scenes of few samples per pixel; a function of the SCENE (depth, its own spp), never of a call's sample range."""
    lib = C.CDLL(os.path.join(_ffi.LIB_DIR, "liblrhip.so"))
    lib.lrhip_pool_auto_triangles.argtypes = [C.c_uint32, C.c_uint32]
    lib.lrhip_pool_auto_triangles.restype = C.c_uint32
    base = lib.lrhip_pool_auto_triangles(16, 1024)
    assert base == 98304
    assert lib.lrhip_pool_auto_triangles(4, 1024) == 2 * base and lib.lrhip_pool_auto_triangles(6, 1024) == 2 * base and lib.lrhip_pool_auto_triangles(7, 1024) == base
    assert lib.lrhip_pool_auto_triangles(16, 16) == base // 2 and lib.lrhip_pool_auto_triangles(16, 64) == base and lib.lrhip_pool_auto_triangles(16, 0) == base
    assert lib.lrhip_pool_auto_triangles(4, 16) == base
    # the sweep's break-even points lie on the right side of it: (triangles, depth, spp) -> pool is the faster family
    for tris, depth, spp, pool_faster in ((30_000, 16, 256, False), (60_000, 16, 16, True), (60_000, 16, 256, False), (100_000, 16, 256, True),
                                          (100_000, 4, 256, False), (400_000, 4, 256, True), (400_000, 16, 16, True), (5_000, 4, 16, False)):
        assert (tris >= lib.lrhip_pool_auto_triangles(depth, spp)) == pool_faster, (tris, depth, spp)
