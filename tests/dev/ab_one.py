#!/usr/bin/env python3
"""development helper (GPU box): one kernel-variant library (CORONA_MI_LIB) -- quick parity against the oracle (paths of cfg 2 and
cfg 3 through the RECORD kernels, ray-level hits bit for bit) in both traversal modes, then kernel time of cfg 2 / cfg 3 at 64 spp
with the production kernels, exact and fast. One line per variant, for A/B tables.  --loops: a -DMI_PROFILE_LOOPS build (slot counts
instead of parity)."""
import os
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *

pkg = load_pkg()
tag = os.path.basename(os.environ.get("CORONA_MI_LIB", "default")).replace("libcorona_mi_", "").replace(".so", "")
out = [f"{tag:14s}"]
loops = "--loops" in sys.argv
modes = [m for m in ("exact", "fast") if f"--no-{m}" not in sys.argv]
for name, sampler, n in (("pt", pkg.MI_SAMPLER_PT, 40000), ("ptdl", pkg.MI_SAMPLER_PTDL, 30000)):
    if f"--no-{name}" in sys.argv:
        continue
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
    be = pkg.Backend(scene)
    if not loops:
        ora = oracle_records(scene, 0, n)
        rng = np.random.default_rng(7)
        nr = 100000
        pos = rng.uniform(-4, 4, size=(nr, 3)).astype(np.float32) + np.float32([0, 0, 2])
        d = rng.normal(size=(nr, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        ohit, _ = oracle_intersect(scene, pos, d)
        for mode in modes:
            be.set_traversal(mode)
            gpu = be.trace_paths(0, n)
            same = (gpu["length"] == ora["length"]) & (gpu["num_splats"] == ora["num_splats"])
            for k in range(1, 8):
                m = ora["length"] > k
                same &= ~m | (gpu["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
            sv = np.abs(gpu["splat"]["value"] - ora["splat"]["value"]) / np.maximum(1e-20, np.abs(ora["splat"]["value"]))
            h = be.intersect(pos, d)
            badhit = int((h["primid"] != ohit["prim"]).sum() + (h["dist"].view(np.uint32) != ohit["dist"].view(np.uint32)).sum())
            out.append(f"{name} {mode}: bad paths {int((~same).sum())}/{n} splat p99.99 {np.quantile(sv[same], .9999):.1e} bad hits {badhit}")
    per = 64 * scene.width * scene.height
    for mode in modes:
        be.set_traversal(mode)
        res = {}
        for counting in (False, True):
            be.set_counters(counting)
            be.render(0, per // 8); be.sync()
            c0 = be.counters()
            ms = []
            for k in range(3 if not counting else 1):
                be.render((k + 1) * per, per); be.sync(); ms.append(be.last_kernel_ms())
            c = [b - a for a, b in zip(c0, be.counters())]
            res[counting] = (min(ms), c)
        ms0, _ = res[False]
        ms1, c = res[True]
        out.append(f"{name} {mode}: {ms0:7.3f} ms {per / ms0 / 1e3:7.1f} Ms/s (counting {ms1:7.3f}) nodes/ray {c[1] / max(c[0], 1):.3f} prims/ray {c[3] / max(c[0], 1):.3f}")
        if loops:
            rays = c[0]
            out.append("slots/ray inner %.2f leaf %.2f analytic %.2f" % (c[2] * 64 / rays, c[5] * 64 / rays, c[6] * 64 / rays))
    be.close()
print(" | ".join(out), flush=True)
