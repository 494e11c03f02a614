"""development helper (GPU box): share of wave time per megakernel phase, from a -DMI_PROFILE_PHASES build
(CORONA_MI_LIB=.../libcorona_mi_phase.so python3 tools/phase_probe.py)"""
import sys, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
for name, mv, sampler in (("pt mv8", 8, 0), ("ptdl mv8", 8, 1)):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=mv, sampler=sampler)
    be = pkg.Backend(scene)
    per = scene.width * scene.height
    be.render(0, per); be.sync()
    c0 = be.counters(); be.render(per, 8 * per); be.sync(); c1 = be.counters()
    d = [b - a for a, b in zip(c0, c1)]
    tot = d[1] + d[2] + d[3] + d[5]
    print(name, "kernel ms %.2f" % be.last_kernel_ms(), " refill/generate %.1f%%  traversal %.1f%%  shading %.1f%%  splat %.1f%%" %
          (100 * d[1] / tot, 100 * d[2] / tot, 100 * d[3] / tot, 100 * d[5] / tot), " rays/path %.2f" % (d[0] / d[4]))
    be.close()
