"""development helper (GPU box): kernel time of a scene at 64 spp with the exact and the FAST rounds (python3 tests/dev/time_modes.py [fine|rough|metal|0010] [pt|ptdl])"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
which = sys.argv[1] if len(sys.argv) > 1 else "fine"
sampler = pkg.MI_SAMPLER_PTDL if len(sys.argv) > 2 and sys.argv[2] == "ptdl" else pkg.MI_SAMPLER_PT
path = {"0010": SCENE_0010, "fine": SCENE_FINE, "rough": SCENE_ROUGH, "metal": SCENE_METAL}[which]
scene = make_scene(path, width=1280, height=720, max_verts=32 if which == "rough" else 8, sampler=sampler)
out = []
for mode in ("exact", "fast"):
    be = pkg.Backend(scene, counters=False, traversal=mode)
    per = 64 * scene.width * scene.height
    be.render(0, per // 8); be.sync()
    ms = []
    for k in range(3):
        be.render((k + 1) * per, per); be.sync(); ms.append(be.last_kernel_ms())
    out.append(f"{mode} {min(ms):.2f} ms")
    be.close()
print(which, "ptdl" if sampler == pkg.MI_SAMPLER_PTDL else "pt", " | ".join(out))
