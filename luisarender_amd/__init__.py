"""luisarender_amd — MI355X-native megakernel path tracer behind LuisaRender's MegaPath boundary.

The product lives in csrc/ (C++ host library + hand-written HIP for gfx950) behind the C ABIs in
include/; this package is the thin Python plumbing used by tests, bench.py and __graft_entry__.py.
"""
from .scene import Scene, HostError, save_image  # noqa: F401
