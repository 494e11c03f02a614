import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
import numpy as np
pkg = load_pkg()
scene = make_scene(SCENE_CAM_MB, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
be = pkg.Backend(scene)
gpu = be.trace_paths(12345, 20000); ora = oracle_records(scene, 12345, 20000)
same = gpu["length"] == ora["length"]
print("same", same.mean(), "splats", (gpu["num_splats"] == ora["num_splats"]).mean())
m = same & (gpu["num_splats"] == ora["num_splats"]) & (ora["num_splats"] > 0)
a, b = gpu["splat"]["value"][m, 0], ora["splat"]["value"][m, 0]
d = np.abs(a-b)/np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))
print("quantiles", [float(np.quantile(d, q)) for q in (0.5, 0.9, 0.99, 0.999)], "n", m.sum())
w = np.argsort(d)[-5:]
idx = np.nonzero(m)[0][w]
for k in idx:
    print(k, gpu["splat"]["value"][k][:3], ora["splat"]["value"][k][:3], gpu["splat"]["length"][k][:3], "len", gpu["length"][k])
    for v in range(min(4, gpu["length"][k])):
        print("    ", v, gpu["v"]["x"][k, v], ora["v"]["x"][k, v], gpu["v"]["throughput"][k, v], ora["v"]["throughput"][k, v])
for k in range(0, 3):
    mm = same & (ora["length"] > k)
    print("v", k, "x dev max", np.abs(gpu["v"]["x"][mm, k]-ora["v"]["x"][mm, k]).max(), "n dev", np.abs(gpu["v"]["n"][mm, k]-ora["v"]["n"][mm, k]).max())
def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))
for k in range(1, 8):
    mm = same & (ora["length"] > k)
    if not mm.sum(): continue
    dx = np.abs(gpu["v"]["x"][mm, k] - ora["v"]["x"][mm, k]).max(axis=1)
    print("k", k, "n", mm.sum(), "prim", (gpu["v"]["prim"][mm, k] == ora["v"]["prim"][mm, k]).mean(), "dx p99.9", np.quantile(dx, .999),
          "thr p99.9", np.quantile(rel(gpu["v"]["throughput"][mm, k], ora["v"]["throughput"][mm, k]), .999),
          "pdf p99.9", np.quantile(rel(gpu["v"]["pdf"][mm, k], ora["v"]["pdf"][mm, k]), .999),
          "flags", (gpu["v"]["flags"][mm, k] == ora["v"]["flags"][mm, k]).mean(), "mode", (gpu["v"]["mode"][mm, k] == ora["v"]["mode"][mm, k]).mean())
