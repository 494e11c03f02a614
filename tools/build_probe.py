"""development helper (GPU box): device-built tree vs host-built tree: hits, counters, speed"""
import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from helpers import *
pkg = load_pkg()
for name, path in (("0010_pt", SCENE_0010), ("0054_fine", SCENE_FINE)):
    scene = make_scene(path, width=1280, height=720, max_verts=8)
    t0 = time.perf_counter(); host = pkg.Backend(scene); t1 = time.perf_counter()
    devb = pkg.Backend(scene, device_build=True); t2 = time.perf_counter()
    rng = np.random.default_rng(1)
    n = 200000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a = host.intersect(pos, d); b = devb.intersect(pos, d)
    same = (a["primid"] == b["primid"])
    print(name, "create host-tree %.1f ms, device-build %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3),
          "same primid %.6f" % same.mean(), "dist bit-equal where same %.6f" % (a["dist"][same].view(np.uint32) == b["dist"][same].view(np.uint32)).mean(),
          "hit rate", (a["primid"] != 0xffffffffffffffff).mean())
    per = scene.width * scene.height
    for be, nm in ((host, "host tree"), (devb, "device tree")):
        be.render(0, per); be.sync()
        c0 = be.counters(); t = time.perf_counter(); be.render(per, 16 * per); be.sync(); ms = (time.perf_counter() - t) * 1e3; c1 = be.counters()
        dc = [y - x for x, y in zip(c0, c1)]
        print("   %-12s %8.1f Msamples/s  nodes/ray %.2f prims/ray %.2f max stack %d" % (nm, 16 * per / ms / 1e3, dc[1] / dc[0], dc[3] / dc[0], c1[7]))
    host.close(); devb.close()
