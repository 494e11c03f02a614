"""development helper (GPU box): where a wave's time goes, from a -DMI_PROFILE_TRAV build (lane 0's clock ticks per part of the
wave iteration, production kernels): CORONA_MI_LIB=.../libcorona_mi_trav.so python3 tools/trav_probe.py"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
names = ["node loop", "job set-up", "job passes", "owner epilogue", "exchange", "refill+shade", "splat"]
for name, sampler in (("pt", 0), ("ptdl", 1)):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
    for mode in ("exact", "fast"):
        be = pkg.Backend(scene, counters=False, traversal=mode)
        per = scene.width * scene.height
        be.render(0, per); be.sync()
        c0 = be.counters(); be.render(per, 16 * per); be.sync(); c1 = be.counters()
        d8 = [b - a for a, b in zip(c0, c1)]
        d, iters = d8[:7], d8[7]
        tot = float(sum(d))
        print(f"{name} {mode}: kernel {be.last_kernel_ms():.2f} ms for 16 spp | " + " | ".join(f"{n} {100 * x / tot:.1f}%" for n, x in zip(names, d)) + f" | wave iterations {iters} = {16 * per / iters:.1f} paths each")
        be.close()
