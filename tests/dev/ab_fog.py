#!/usr/bin/env python3
"""development helper (GPU box): one kernel-variant library (CORONA_MI_LIB, a -DMI_DEV_FAST=3 build: plain + extended exact kernels) on the
media scenes -- paths of the RECORD kernels against the oracle, then kernel ms of one 64-spp frame (production kernels), one line.
--trav: a -DMI_PROFILE_TRAV build (lane 0's ticks per part of the wave iteration instead of parity)."""
import os
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *

pkg = load_pkg()
tag = os.path.basename(os.environ.get("CORONA_MI_LIB", "default")).replace("libcorona_mi_", "").replace(".so", "")
trav = "--trav" in sys.argv
names = ["node loop", "job set-up", "job passes", "owner epilogue", "exchange", "refill+shade", "splat"]
out = []
for name, path, sampler, n in (("fog", SCENE_FOG, pkg.MI_SAMPLER_PT, 20000), ("fog_ptdl", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 20000),
                               ("media_ptdl", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 20000), ("cfg3", SCENE_0010, pkg.MI_SAMPLER_PTDL, 20000)):
    if any(a.startswith("--only=") for a in sys.argv) and name not in [a[7:] for a in sys.argv if a.startswith("--only=")][0].split(","):
        continue
    scene = make_scene(path, width=1280, height=720, max_verts=8, sampler=sampler)
    be = pkg.Backend(scene, counters=False)
    s = f"{name}:"
    if not trav and "--no-parity" not in sys.argv:
        ora = oracle_records(scene, 0, n)
        gpu = be.trace_paths(0, n)
        same = (gpu["length"] == ora["length"]) & (gpu["num_splats"] == ora["num_splats"])
        for k in range(1, 8):
            m = ora["length"] > k
            same &= ~m | (gpu["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
        s += f" bad paths {int((~same).sum())}/{n}"
    per = scene.width * scene.height
    be.render(0, 8 * per); be.sync()
    if trav:
        c0 = be.counters(); be.render(8 * per, 16 * per); be.sync(); c1 = be.counters()
        d8 = [b - a for a, b in zip(c0, c1)]
        d, iters = d8[:7], d8[7]
        tot = float(sum(d))
        s += f" kernel {be.last_kernel_ms():.2f} ms for 16 spp | " + " | ".join(f"{n_} {100 * x / tot:.1f}%" for n_, x in zip(names, d)) + f" | ticks/iteration {tot / max(iters, 1):.0f} | paths per wave iteration {16 * per / max(iters, 1):.2f}"
    else:
        ms = []
        for k in range(3):
            be.render((k + 1) * 64 * per, 64 * per); be.sync(); ms.append(be.last_kernel_ms())
        fb = be.fb_read().astype(np.float64)
        s += f" {min(ms):8.3f} ms (image mean {fb.mean() / 200:.7f} rms {np.sqrt((fb * fb).mean()) / 200:.7f})"
    out.append(s)
    be.close()
print(f"{tag:12s} " + " | ".join(out))
