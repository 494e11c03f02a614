import sys, time, os
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
mv = int(os.environ.get("PROBE_MV", "8"))
sampler = int(os.environ.get("PROBE_SAMPLER", "0"))
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=mv, sampler=sampler)
be = pkg.Backend(scene)
per = scene.width * scene.height
be.render(0, per); be.sync()
t0 = time.perf_counter(); be.render(per, 16 * per); be.sync(); print("wall ms", (time.perf_counter() - t0) * 1e3, "launches", be.last_kernel_launches())
