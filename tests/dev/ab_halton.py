#!/usr/bin/env python3
"""development helper (GPU box): the Halton kernels of one library variant (CORONA_MI_LIB) -- parity against the oracle, kernel time"""
import os, sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
tag = os.path.basename(os.environ.get("CORONA_MI_LIB", "default")).replace("libcorona_mi_", "").replace(".so", "")
out = [f"{tag:14s} halton"]
for name, sampler, n in (("pt", pkg.MI_SAMPLER_PT, 30000), ("ptdl", pkg.MI_SAMPLER_PTDL, 20000)):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler, pointsampler=pkg.MI_POINTS_HALTON)
    be = pkg.Backend(scene, counters=False)
    gpu = be.trace_paths(0, n); ora = oracle_records(scene, 0, n)
    same = (gpu["length"] == ora["length"]) & (gpu["num_splats"] == ora["num_splats"]) & (gpu["pixel_i"] == ora["pixel_i"]) & (gpu["lambda"] == ora["lambda"])
    per = 64 * scene.width * scene.height
    be.render(0, per // 8); be.sync()
    ms = []
    for k in range(3):
        be.render((k + 1) * per, per); be.sync(); ms.append(be.last_kernel_ms())
    out.append(f"{name}: bad paths {int((~same).sum())}/{n} {min(ms):7.3f} ms {per / min(ms) / 1e3:7.1f} Ms/s")
    be.close()
print(" | ".join(out), flush=True)
