"""Cornell box in the reference's text scene grammar (BASELINE config C1).

Classic Cornell-box data (Cornell University Program of Computer Graphics): floor, ceiling, back
wall, red/green walls, short and tall box, ceiling light quad — 36 triangles with closed boxes...
here 5-face boxes (32 triangles) exactly as the classic data set lists them.
"""

_QUAD_INDICES = "0, 1, 2, 0, 2, 3"


def _quad(name, pts, surface, light=None):
    pos = ", ".join(f"{v:g}" for p in pts for v in p)
    extra = f"\n  light : Diffuse {{ emission : Constant {{ v {{ {light} }} }} }}" if light else ""
    return (f"Shape {name} : InlineMesh {{\n  positions {{ {pos} }}\n  indices {{ {_QUAD_INDICES} }}\n"
            f"  surface {{ @{surface} }}{extra}\n}}\n")


def cornell_box(resolution=512, spp=64, depth=8, sampler="Independent", filter_impl="Box", filter_radius=0.5,
                file="cornell.exr", rr_depth=0, short_box_surface="white", tall_box_surface="white",
                extra_surfaces="", seed=19980810):
    s = []
    s.append("Surface white : Matte { Kd : Constant { v { 0.725, 0.71, 0.68 } } }\n")
    s.append("Surface red : Matte { Kd : Constant { v { 0.63, 0.065, 0.05 } } }\n")
    s.append("Surface green : Matte { Kd : Constant { v { 0.14, 0.45, 0.091 } } }\n")
    s.append(extra_surfaces)
    s.append(_quad("floor", [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)], "white"))
    s.append(_quad("ceiling", [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)], "white"))
    s.append(_quad("back", [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)], "white"))
    s.append(_quad("right", [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)], "green"))
    s.append(_quad("left", [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)], "red"))
    s.append(_quad("lamp", [(343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227)], "white",
                   light="17, 12, 4"))
    short = [
        [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
        [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
        [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
        [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
        [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)],
    ]
    tall = [
        [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
        [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
        [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
        [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
        [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)],
    ]
    names = []
    for i, q in enumerate(short):
        s.append(_quad(f"short{i}", q, short_box_surface))
        names.append(f"@short{i}")
    for i, q in enumerate(tall):
        s.append(_quad(f"tall{i}", q, tall_box_surface))
        names.append(f"@tall{i}")
    res = f"{resolution[0]}, {resolution[1]}" if isinstance(resolution, (tuple, list)) else f"{resolution}, {resolution}"
    s.append(f"""Camera cam : Pinhole {{
  fov {{ 39.3 }}  spp {{ {spp} }}  file {{ "{file}" }}
  film : Color {{ resolution {{ {res} }} }}
  filter : {filter_impl} {{ radius {{ {filter_radius} }} }}
  transform : View {{ position {{ 278, 273, -800 }}  front {{ 0, 0, 1 }}  up {{ 0, 1, 0 }} }}
}}
render {{
  cameras {{ @cam }}
  shapes {{ @floor, @ceiling, @back, @right, @left, @lamp, {", ".join(names)} }}
  integrator : MegaPath {{ depth {{ {depth} }}  rr_depth {{ {rr_depth} }}
    sampler : {sampler} {{ seed {{ {seed} }} }} }}
}}
""")
    return "".join(s)
