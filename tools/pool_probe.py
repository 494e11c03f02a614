"""development helper (GPU box): what the exchange between waves (mi_regroup.h) does, from a -DMI_PROFILE_POOL build
(CORONA_MI_LIB=.../libcorona_mi_pool.so python3 tools/pool_probe.py): per wave iteration."""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
for name, sampler in (("pt", 0), ("ptdl", 1)):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
    be = pkg.Backend(scene, counters=False)
    per = scene.width * scene.height
    be.render(0, per); be.sync()
    c0 = be.counters(); be.render(per, 16 * per); be.sync(); c1 = be.counters()
    d = [b - a for a, b in zip(c0, c1)]
    lanes = [x & ((1 << 36) - 1) for x in d]; execs = [x >> 36 for x in d]
    it = max(execs[0] + 1, 1)   # slot 7 is a maximum over the workgroups; every slot below is a sum
    wg_it = max(execs[7], 1)
    print(f"{name} [{be.traversal()}]: wave iterations per workgroup {wg_it}; exchanges {execs[0]} ({execs[0] / (wg_it * 256.0):.2f} per wave iteration); posted {lanes[0] / max(execs[0], 1):.1f} lanes per exchange, "
          f"pulled {lanes[1] / max(execs[1], 1):.1f} lanes in {execs[1] / max(execs[0], 1):.3f} of them; turns to a class the wave's own lanes are not mostly in: {execs[2] / max(execs[0], 1):.3f} "
          f"of the exchanges, {lanes[2] / max(execs[2], 1):.1f} lanes shaded then; chosen class {lanes[5] / max(execs[5], 1):.1f} lanes per exchange; "
          f"lock: {16 * lanes[3] / max(execs[3], 1):.0f} ticks waiting, {16 * lanes[6] / max(execs[6], 1):.0f} ticks holding per first critical section; vertices left in place (pool full) {lanes[4] / max(execs[0], 1):.2f} per exchange in {execs[4] / max(execs[0], 1):.3f} of them")
    be.close()
