import sys, time, numpy as np
sys.path.insert(0, '.')
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from oracle.check import Oracle
from luisarender_amd.scenes import cornell_box
sc = Scene.from_string(cornell_box(resolution=128, spp=16))
r = MegaPathRenderer(0)
r.upload(sc)
t = time.time(); r.render(0, 16, counters=True, sync=True); print('gpu time', time.time() - t, r.last_render_ms())
g = r.download(converted=False)
print('gpu counters', r.counters())
o = Oracle(sc)
f, c = o.render(0, 16)
print('oracle counters', c)
d = np.abs(g - f)
print('max abs diff', d.max(), 'mean', d.mean(), 'film mean', f[..., :3].mean(), g[..., :3].mean())
rel = np.abs(g[..., :3] - f[..., :3]).sum() / np.abs(f[..., :3]).sum()
print('rel L1', rel, 'exact pixels', (np.abs(g - f).max(axis=-1) == 0).mean())
np.save('gpurun_out/gpu.npy', r.download(True)); np.save('gpurun_out/cpu.npy', o.convert(f))
r.clear(); t = time.time(); r.render(0, 64, sync=True); print('64spp', r.last_render_ms(), 'ms')
