import sys, tempfile, numpy as np
sys.path.insert(0, '.')
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import generate_room_scene
with tempfile.TemporaryDirectory() as tmp:
    sc = Scene.load(generate_room_scene(tmp, resolution=(1024, 1024), spp=8))
    r = MegaPathRenderer(0); r.upload(sc); r.render(0, 8, sync=True)
    film = r.download(False)
    sub, _ = Oracle(sc).render(0, 8, rect=(384, 384, 640, 640))
    a, b = film[384:640, 384:640, :3], sub[384:640, 384:640, :3]
    d = np.abs(a - b).max(axis=-1)
    print('mean diff', abs(a.mean() - b.mean()) / b.mean(), 'relL1', np.abs(a - b).sum() / np.abs(b).sum(), 'pixels differing >1%', (d > 0.01 * (np.abs(b).max(axis=-1) + 1e-3)).mean(), 'max abs', d.max())
