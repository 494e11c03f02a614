#!/usr/bin/env python3
"""development helper (GPU box): the 2048-spp reference render (tests/golden/tilemeans_pt_mv8.npz, 32x32 tile means) against the GPU at
the same sample count, rendered as Q independent parts whose scatter gives the per-tile noise."""
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
name = sys.argv[1] if len(sys.argv) > 1 else "pt_mv8"
g = np.load(GOLDEN / f"tilemeans_{name}.npz")
sampler = pkg.MI_SAMPLER_PTDL if "ptdl" in name else pkg.MI_SAMPLER_PT
scene = make_scene(SCENE_METAL if "metal" in name else SCENE_ROUGH if "rough" in name else SCENE_0010, width=int(g["width"]), height=720 if int(g["height"]) == 736 else int(g["height"]), max_verts=int(g["max_verts"]), sampler=sampler)
be = pkg.Backend(scene, counters=False)
per = scene.width * scene.height
spp = int(g["spp"])
Q = 8
parts = []
for q in range(Q):
    be.fb_clear()
    be.render((5000 + q * spp // Q) * per, spp // Q * per)
    parts.append((be.fb_read() * scene.gain(spp // Q)).reshape(scene.height // 32, 32, scene.width // 32, 32, 3).mean(axis=(1, 3)))
parts = np.array(parts)
gpu = parts.mean(axis=0)
var = parts.var(axis=0, ddof=1) / Q                 # variance of the Q-part mean, per tile and channel
ref = g["tiles"]
d = gpu - ref
z = d / np.sqrt(2 * var)
print("tiles", ref.shape, "spp", spp, "mean gpu", gpu.mean(axis=(0, 1)), "ref", ref.mean(axis=(0, 1)), "ratio", gpu.mean(axis=(0, 1)) / ref.mean(axis=(0, 1)))
print("mean d^2 %.3e  mean 2 var %.3e  ratio %.2f" % ((d ** 2).mean(), (2 * var).mean(), (d ** 2).mean() / (2 * var).mean()))
print("mean z^2 %.2f  robust sigma(z) %.2f  median z %.3f" % ((z ** 2).mean(), 1.4826 * np.median(np.abs(z - np.median(z))), np.median(z)))
print("fraction |z| > 3: %.4f  > 4: %.4f; corr of the lit rows (8..): %.5f" % ((np.abs(z) > 3).mean(), (np.abs(z) > 4).mean(), np.corrcoef(gpu[8:, :, 1].ravel(), ref[8:, :, 1].ravel())[0, 1]))
order = np.argsort(-np.abs(z[..., 1]).ravel())[:24]
print("most significant luminance tiles (row, col): z, gpu / ref")
print("  ".join("(%d,%d) %+.1f %.4f" % (i // z.shape[1], i % z.shape[1], z[..., 1].ravel()[i], (gpu[..., 1] / ref[..., 1]).ravel()[i]) for i in order))
sig = np.abs(z[..., 1]) > 3
if sig.any():
    print("tiles with |z| > 3: %d, their summed luminance gpu / ref = %.4f; all other tiles: %.4f" % (sig.sum(), gpu[..., 1][sig].sum() / ref[..., 1][sig].sum(), gpu[..., 1][~sig].sum() / ref[..., 1][~sig].sum()))
be.close()
