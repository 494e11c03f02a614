#!/usr/bin/env python3
"""development helper (GPU box): the film of a 1-spp render against the oracle's, float by float: how many floats / pixels differ by more than a relative bound"""
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
for sampler in (pkg.MI_SAMPLER_PT, pkg.MI_SAMPLER_PTDL):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
    n = scene.width * scene.height
    be = pkg.Backend(scene, counters=False)
    be.render(0, n)
    fb = be.fb_read().astype(np.float64)
    be.close()
    ofb = oracle_render(scene, 0, n, threads=8)[0].astype(np.float64)
    rel = np.abs(fb - ofb) / np.maximum(np.maximum(np.abs(fb), np.abs(ofb)), 1e-30)
    print("sampler", sampler, "floats != 0: %.4f" % (fb != 0).mean(), "zero pattern equal: %.6f" % ((fb == 0) == (ofb == 0)).mean(),
          "pixels with a float off by more than", {t: int((rel > t).any(axis=2).sum()) for t in (1e-7, 1e-6, 4e-6, 1e-5, 1e-4, 1e-3, 1e-2)},
          "bitwise equal floats: %.4f" % (fb.astype(np.float32).view(np.uint32) == ofb.astype(np.float32).view(np.uint32)).mean())
