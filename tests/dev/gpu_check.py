#!/usr/bin/env python3
"""Quick on-GPU check used during development: HIP path vs oracle records + image + timing."""
import sys, time, json
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *

pkg = load_pkg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=mv)
be = pkg.Backend(scene)
t0 = time.time(); gpu = be.trace_paths(0, N); t1 = time.time()
ora = oracle_records(scene, 0, N)
print("trace_paths", N, "in", t1 - t0, "s")
lm = gpu["length"] == ora["length"]
print("length match", lm.mean(), "splat count match", (gpu["num_splats"] == ora["num_splats"]).mean())
for f in ["pixel_i", "pixel_j", "lambda", "time", "scramble"]:
    print(f, np.abs(gpu[f] - ora[f]).max())
both = lm
def rel(a, b): return np.abs(a - b) / np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))
print("throughput rel max", rel(gpu["throughput"][both], ora["throughput"][both]).max())
for k in range(1, 8):
    m = both & (ora["length"] > k)
    if not m.sum(): continue
    pm = gpu["v"]["prim"][m, k] == ora["v"]["prim"][m, k]
    dx = np.abs(gpu["v"]["x"][m, k] - ora["v"]["x"][m, k]).max(axis=1)
    dn = np.abs(gpu["v"]["n"][m, k] - ora["v"]["n"][m, k]).max(axis=1)
    mm = gpu["v"]["mode"][m, k] == ora["v"]["mode"][m, k]
    tv = rel(gpu["v"]["throughput"][m, k], ora["v"]["throughput"][m, k])
    pv = rel(gpu["v"]["pdf"][m, k], ora["v"]["pdf"][m, k])
    print(f"v{k}: n={m.sum()} prim {pm.mean():.6f} mode {mm.mean():.6f} dx max {dx.max():.3g} dn max {dn.max():.3g} thr rel max {tv.max():.3g} p99.9 {np.quantile(tv, .999):.3g} pdf rel p99.9 {np.quantile(pv, .999):.3g}")
bad = np.where(~lm)[0][:5]
for b in bad:
    print("mismatch", b, "gpu len", gpu["length"][b], "ora len", ora["length"][b])
    for k in range(min(8, max(gpu["length"][b], ora["length"][b]))):
        print("   g", hex(gpu["v"]["prim"][b, k]), gpu["v"]["mode"][b, k], gpu["v"]["throughput"][b, k], gpu["v"]["x"][b, k], "| o", hex(ora["v"]["prim"][b, k]), ora["v"]["mode"][b, k], ora["v"]["throughput"][b, k], ora["v"]["x"][b, k])
# image + timing
W, H = scene.width, scene.height
per = W * H
be.fb_clear()
be.render(0, per); be.sync()   # warmup
be.fb_clear(); be.sync()
c0 = be.counters()
t0 = time.time()
for s in range(spp):
    be.render(s * per, per)
be.sync()
t1 = time.time()
c1 = be.counters()
fb = be.fb_read()
dc = [b - a for a, b in zip(c0, c1)]
print("render", spp, "spp:", t1 - t0, "s ->", spp * per / (t1 - t0) / 1e6, "Msamples/s; last kernel ms", be.last_kernel_ms())
print("counters per sample: rays %.4f nodes %.4f boxhits %.4f prims %.4f splats %.5f verts %.4f" % tuple(dc[i] / dc[4] for i in (0, 1, 2, 3, 5, 6)))
gain = scene.gain(spp)
print("mean image (gain-scaled XYZ)", (fb * gain).mean(axis=(0, 1)))
ofb, ocnt, secs = oracle_render(scene, 0, per, threads=8)
print("oracle 1 spp: %.2f s -> %.3f Msamples/s (8 threads); counters per sample rays %.4f nodes %.4f prims %.4f" % (secs, per / secs / 1e6, ocnt[0] / ocnt[4], ocnt[1] / ocnt[4], ocnt[3] / ocnt[4]))
be.fb_clear(); be.render(0, per); g1 = be.fb_read()
d = g1 - ofb
print("1spp image: gpu sum", g1.sum(axis=(0, 1)), "oracle sum", ofb.sum(axis=(0, 1)), "max abs diff", np.abs(d).max(), "rmse", np.sqrt((d ** 2).sum() / per) * scene.gain(1))
