"""development helper (GPU box): share of wave time per megakernel phase, from a -DMI_PROFILE_PHASES build
(CORONA_MI_LIB=.../libcorona_mi_phase.so python3 tools/phase_probe.py)"""
import sys, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
which = sys.argv[1] if len(sys.argv) > 1 else "0010"        # python3 tools/phase_probe.py [0010|fog|media|mb] [hero]
hero = len(sys.argv) > 2 and sys.argv[2] == "hero"
path = {"0010": SCENE_0010, "fog": SCENE_FOG, "media": SCENE_MEDIA, "mb": SCENE_MB}[which]
for name, mv, sampler in (("pt mv8", 8, 0), ("ptdl mv8", 8, 1)):
    scene = make_scene(path, width=1280, height=720, max_verts=mv, sampler=sampler)
    be = pkg.Backend(scene)
    if hero:
        be.set_wavelengths(4)
    per = scene.width * scene.height
    be.render(0, per); be.sync()
    c0 = be.counters(); be.render(per, 8 * per); be.sync(); c1 = be.counters()
    d = [b - a for a, b in zip(c0, c1)]
    names = ["refill/generate", "traversal", "surface_setup", "prepare+media", "emit/RR/NEE", "sample tail", "splat", "bsdf sample"]
    ticks = [x & ((1 << 36) - 1) for x in d]; occ = [x >> 36 for x in d]
    iters = occ[1]
    # wave time of a phase ~ average ticks per occurrence (lane 0) x wave iterations
    est = [t / max(o, 1) * iters for t, o in zip(ticks, occ)]
    tot = sum(est)
    print(name, "kernel ms %.2f" % be.last_kernel_ms(), " | ".join("%s %.1f%% (%.0f ticks)" % (n, 100 * e / tot, t / max(o, 1)) for n, e, t, o in zip(names, est, ticks, occ)))
    be.close()
