"""development helper (GPU box): wave-level iteration counts of the traversal loops, from a -DMI_PROFILE_LOOPS build
(CORONA_MI_LIB=.../libcorona_mi_loops.so python3 tools/loop_probe.py)"""
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
print("rays/path %.3f  node visits/ray %.2f  prim tests/ray %.2f" % (rays / c[4], c[1] / rays, c[3] / rays))
print("lane slots per ray (wave-level iterations x 64 / rays):  inner %.2f (useful %.2f)   leaf %.2f (useful %.2f)   analytic %.2f" %
      (c[2] * 64 / rays, c[1] / rays, c[5] * 64 / rays, c[3] / rays, c[6] * 64 / rays))
