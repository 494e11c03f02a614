import sys, time, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
be = pkg.Backend(scene)
per = scene.width * scene.height
c0 = be.counters()
be.render(0, 16 * per); be.sync()
c = [b - a for a, b in zip(c0, be.counters())]
rays = c[0]
# cnt[4..6] are polluted by wf_logic's paths/splats/verts; logic adds paths (c4), splats (c5), verts (c6): subtract known values
paths = 16 * per
print("rays", rays, "nodes/ray", c[1] / rays, "prims/ray", c[3] / rays)
print("wave-level inner iterations per ray x64:", (c[4] - paths) * 64 / rays)
print("wave-level leaf slots per ray x64:", (c[5]) * 64 / rays, "(minus splats ~0)")
print("wave-level analytic passes per ray x64:", (c[6] - (c0[6] * 0)) * 64 / rays, "(polluted by verts)")
