import sys, os, tempfile, numpy as np
sys.path.insert(0,'.')
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_room_scene
from oracle.check import Oracle
import luisarender_amd._ffi as ffi
def rel(a,b): return float(np.abs(a[...,:3]-b[...,:3]).sum()/np.abs(b[...,:3]).sum())
with tempfile.TemporaryDirectory() as d:
    for name, opt, kw in (("baked100k", dict(bake_transforms=True), dict(target_triangles=100_000, resolution=(256,256), spp=8)),
                          ("instanced100k", dict(inline_meshes=True), dict(target_triangles=100_000, resolution=(256,256), spp=8)),
                          ("instanced600k_1024", dict(), dict(target_triangles=600_000, resolution=(1024,1024), spp=8))):
        sc = Scene.load(generate_room_scene(d, name=name, **opt, **kw))
        w = kw["resolution"][0]
        rect = (0,0,w,w) if w==256 else (384,384,640,640)
        c,_ = Oracle(sc).render(0, 8, rect=rect)
        for lib in (None, os.path.join(ffi.LIB_DIR, "variants", "liblrhip_ieee.so")):
            r = MegaPathRenderer(0, lib_path=lib); r.upload(sc); r.render(0, 8, sync=True); g = r.download(False); r.close()
            a, b = g[rect[1]:rect[3], rect[0]:rect[2]], c[rect[1]:rect[3], rect[0]:rect[2]]
            off = float((np.abs(a[...,:3]-b[...,:3]).max(axis=-1) > 1e-4*np.abs(b[...,:3]).max(axis=-1)+1e-7).mean())
            print(name, 'ieee' if lib else 'shipped', 'rel-L1 %.3e' % rel(a,b), 'pixels off %.3e' % off, flush=True)
