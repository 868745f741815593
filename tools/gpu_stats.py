"""SIMD-occupancy diagnostics of the megakernel on the bench workload (GPU box).  LRHIP_SCHEDULER=legacy: the round 1-3 kernels."""
import os, sys, tempfile
sys.path.insert(0, '.')
from luisarender_amd import Scene
from luisarender_amd.render import MegaPathRenderer
from luisarender_amd.scenes import generate_room_scene
from luisarender_amd.scenes.configs import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"   # c1 | c2 | c3 | c4 | c5 (the bench's stand-ins, at 1024 x 1024 / 1280 x 720)
with tempfile.TemporaryDirectory() as tmp:
    gen = {"c1": lambda: None, "c2": lambda: generate_room_scene(tmp, resolution=(1024, 1024), spp=spp),
           "c3": lambda: generate_bedroom_scene(tmp, resolution=(1280, 720), spp=spp),
           "c4": lambda: generate_camera_scene(tmp, resolution=(1280, 720), spp=spp),
           "c5": lambda: generate_kitchen_scene(tmp, resolution=(1280, 720), spp=spp)}[workload]
    if workload == 'c1':
        from luisarender_amd.scenes import cornell_box
        sc = Scene.from_string(cornell_box(resolution=512, spp=spp, depth=8))
    else:
        sc = Scene.load(gen())
    r = MegaPathRenderer(0)
    if os.environ.get('LRHIP_SCHEDULER') in ('legacy', 'pool'):  # (the override lives HERE, not in the renderer: ADVICE r04)
        r.set_scheduler(os.environ['LRHIP_SCHEDULER'] == 'pool')
    r.upload(sc)
    r.render(0, spp, counters=True, sync=True)
    c = r.counters()
    rays = c['closest_rays'] + c['shadow_rays']
    print(c)
    print('variant', r.last_variant())
    print('ms', r.last_render_ms(), 'rays/sample', rays / c['paths'], 'nodes/ray', c['nodes_visited'] / rays, 'tris/ray', c['tris_tested'] / rays)
    print('trace lane utilisation', c['trace_steps_busy'] / max(c['trace_steps'], 1), 'steps per ray-lane', c['trace_steps_busy'] / rays)
    print('trace lanes starved (no sample left for the pixel)', c['trace_steps_starved'] / max(c['trace_steps'], 1))
    print('node visits without a hit child', c['nodes_empty'] / c['nodes_visited'])
    print('shade lane utilisation', c['shade_busy'] / max(c['shade_calls'], 1))
    print('wave cycles: shade %.3f  trace %.3f  of wave lifetime; cycles per wave-level trace step %.0f, per wave-level shade call %.0f' % (
        c['shade_cycles'] / c['wave_cycles'], c['trace_cycles'] / c['wave_cycles'], c['trace_cycles'] / (c['trace_steps'] / 64), c['shade_cycles'] / (c['shade_calls'] / 64)))
    print('shading block split (of wave lifetime): hit + emission + light sample + parking %.3f  closure evaluate / sample / RR %.3f  path regeneration %.3f  launch + rest %.3f' % (
        c['shade_light_cycles'] / c['wave_cycles'], c['shade_closure_cycles'] / c['wave_cycles'], c['shade_regen_cycles'] / c['wave_cycles'],
        (c['shade_cycles'] - c['shade_light_cycles'] - c['shade_closure_cycles'] - c['shade_regen_cycles']) / c['wave_cycles']))
