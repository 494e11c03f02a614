#!/usr/bin/env python3
"""development helper (GPU box): kernel time of cfg 2 / cfg 3 / media / fog (64 spp, production kernels) with the library given in
CORONA_MI_LIB -- which may be the ROUND-2 build (ABI 2: the descriptor's header is patched for it) -- for same-box A/B against it."""
import ctypes as C
import os
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *

pkg = load_pkg()
lib = os.environ.get("CORONA_MI_LIB", "")
old = "r02" in lib or os.environ.get("AB_ABI2") == "1"          # a library of ABI 2 (round 2, or this round before nodes_t1)
if old:
    class Dummy:                                   # entry points the round-2 library does not have yet
        argtypes = restype = None
        def __call__(self, *a): return 0
    class Lenient(C.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if name.startswith("mi_"):
                    d = Dummy(); setattr(self, name, d); return d
                raise
    _cdll, C.CDLL = C.CDLL, Lenient
    pkg.mi_lib()
    C.CDLL = _cdll
out = [(os.path.basename(lib) if lib else "this library") + (" exact" if "--exact" in sys.argv else "") + ":"]
only = os.environ.get("AB_ONLY")
for name, path, sampler in (("pt", SCENE_0010, 0), ("ptdl", SCENE_0010, 1), ("media pt", SCENE_MEDIA, 0), ("media ptdl", SCENE_MEDIA, 1), ("fog pt", SCENE_FOG, 0), ("cam_mb pt", SCENE_CAM_MB, 0), ("fog ptdl", SCENE_FOG, 1), ("cam_mb ptdl", SCENE_CAM_MB, 1)):
    if only and name != only:
        continue
    scene = make_scene(path, width=1280, height=720, max_verts=8, sampler=sampler)
    if old:
        d = scene.desc_ptr.contents
        d.struct_size = C.sizeof(pkg.MiSceneDesc) - 8
        d.abi_version = 2
    be = pkg.Backend(scene, counters=False)
    if "r02" not in lib and "--exact" in sys.argv:
        be.set_traversal("exact")
    per = 64 * scene.width * scene.height
    be.render(0, per // 8); be.sync()
    ms = []
    for k in range(3):
        be.render((k + 1) * per, per); be.sync(); ms.append(be.last_kernel_ms())
    out.append(f"{name} {min(ms):.2f} ms")
    be.close()
print(" | ".join(out), flush=True)
