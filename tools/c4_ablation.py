"""Where does the camera-class (C4) frame spend its time?  Renders the stand-in with its image textures replaced by constants and / or its
Disney surfaces by Plastic (same geometry), and prints Msamples/s + the shading block's share (counting twin).
    python tools/c4_ablation.py [spp=64] [case ...]      (LRHIP_LIB selects an experimental build)"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes.configs import generate_camera_scene

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
RES = (1920, 1080)


def no_textures(text):
    text = re.sub(r"(\w+) \{ @albedo\d \}", r"\1 : Constant { v { 0.6, 0.5, 0.4 } }", text)
    return re.sub(r"(\w+) \{ @rough\d \}", r"\1 : Constant { v { 0.4 } }", text)


def no_disney(text):
    return re.sub(r"Surface (disney\d) : Disney \{ color (\{ @albedo\d \}|: Constant \{[^}]*\} \}) roughness (\{ @rough\d \}|: Constant \{[^}]*\} \})[^\n]*\n",
                  r"Surface \1 : Plastic { Kd \2 roughness \3 eta : Constant { v { 1.5 } } }\n", text)


with tempfile.TemporaryDirectory() as tmp:
    path = generate_camera_scene(tmp, resolution=RES, spp=spp, texture_size=int(os.environ.get('TEXTURE_SIZE', '2048')))
    base = open(path).read()
    cases = {"full": base, "no_textures": no_textures(base), "no_disney": no_disney(base), "no_textures_no_disney": no_disney(no_textures(base))}
    if len(sys.argv) > 2:
        cases = {k: v for k, v in cases.items() if k in sys.argv[2:]}
    r = MegaPathRenderer(0)
    r.set_texture_storage(int(os.environ.get('TEXTURE_STORAGE', '1')))  # 0 float texels, 1 automatic, 2 8-bit texels wherever an image qualifies
    for name, text in cases.items():
        p = os.path.join(tmp, name + ".luisa")
        open(p, "w").write(text)
        sc = Scene.load(p)
        r.upload(sc)
        r.render(0, spp, counters=False, sync=True)
        ms, v = r.last_render_ms(), r.last_variant()
        r.upload(sc)
        r.render(0, spp, counters=True, sync=True)
        c = r.counters()
        print(f"{name:24s} variant {v:5d}: {RES[0] * RES[1] * spp / ms / 1e3:7.1f} Msamples/s ({ms:7.1f} ms)   shade share {c['shade_cycles'] / c['wave_cycles']:.2f} "
              f"trace share {c['trace_cycles'] / c['wave_cycles']:.2f}  cycles/shade call {c['shade_cycles'] / (c['shade_calls'] / 64):8.0f}  closure section {c['shade_closure_cycles'] / c['wave_cycles']:.2f}  "
              f"shade lanes {c['shade_busy'] / max(c['shade_calls'], 1):.2f}  path length {c['path_length_sum'] / c['paths']:.2f}", flush=True)
