#!/usr/bin/env python3
"""development helper (GPU box): traversal work and kernel time of the device-built tree against the host-built (reference) tree,
0010_pt (cfg 2) and 0059_mb; one line per scene and tree.  CORONA_MI_LIB=... python3 tests/dev/devtree_probe.py [sah passes ...]
(with arguments: the device-built tree once per number of rotation passes, CORONA_MI_BUILD_SAH)"""
import json
import os
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
gold = json.loads((GOLDEN / "counters.json").read_text())
for name, path, key in (("0010_pt", SCENE_0010, "pt_mv8"), ("0059_mb", SCENE_MB, "mb_pt_mv8")):
    scene = make_scene(path, width=1280, height=720, max_verts=8)
    n = scene.width * scene.height
    for dev, sah in [(False, None)] + [(True, p) for p in (sys.argv[1:] or [None])]:
        if sah is not None:
            os.environ["CORONA_MI_BUILD_SAH"] = sah
        be = pkg.Backend(scene, device_build=dev, traversal="exact")
        c0 = be.counters(); be.render(0, n); be.sync(); c = [b - a for a, b in zip(c0, be.counters())]
        be.set_counters(False)
        ms = []
        for k in range(3):
            be.render((k + 1) * 64 * n, 64 * n); be.sync(); ms.append(be.last_kernel_ms())
        g = gold.get(key, {})
        print(f"{name} {('device' + ('' if sah is None else ' sah=' + sah)) if dev else 'host  '} tree: nodes {be.stats()['nodes']:6d}  node visits {c[1]} ({c[1] / g['node_visits']:.3f} x reference)  prim tests {c[3]} ({c[3] / g['prim_tests']:.3f} x)  64 spp {min(ms):.2f} ms", flush=True)
        be.close()
