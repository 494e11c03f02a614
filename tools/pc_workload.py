"""development helper (GPU box): a few frames of cfg 2 (or cfg 3 with `ptdl`) for a profiler to sample.
   rocprofv3 ... -- python3 tools/pc_workload.py [ptdl] [frames]"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
sampler = 1 if "ptdl" in sys.argv else 0
frames = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 4
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
be = pkg.Backend(scene, counters=False)
per = scene.width * scene.height
for f in range(frames):
    be.render(f * 16 * per, 16 * per)
be.sync()
print("kernel ms", be.last_kernel_ms())
be.close()
