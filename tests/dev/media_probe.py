import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
import numpy as np
pkg = load_pkg()
for sampler, mv, n in ((pkg.MI_SAMPLER_PT, 8, 60000), (pkg.MI_SAMPLER_PTDL, 8, 40000), (pkg.MI_SAMPLER_PT, 32, 20000), (pkg.MI_SAMPLER_PTDL, 32, 20000)):
    scene = make_scene(SCENE_MEDIA, width=1280, height=720, max_verts=mv, sampler=sampler)
    be = pkg.Backend(scene)
    gpu = be.trace_paths(12345, n)
    ora = oracle_records(scene, 12345, n)
    same = gpu["length"] == ora["length"]
    print("sampler", sampler, "mv", mv, "same length", same.mean(), "splats", (gpu["num_splats"] == ora["num_splats"]).mean())
    bad = np.nonzero(~same)[0]
    print("  mismatches", bad[:8])
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if m.sum():
            t = np.abs(gpu["v"]["throughput"][m, k] - ora["v"]["throughput"][m, k]) / np.maximum(1e-20, np.abs(ora["v"]["throughput"][m, k]))
            pd = np.abs(gpu["v"]["pdf"][m, k] - ora["v"]["pdf"][m, k]) / np.maximum(1e-20, np.abs(ora["v"]["pdf"][m, k]))
            print("  v", k, "prim", (gpu["v"]["prim"][m, k] == ora["v"]["prim"][m, k]).mean(), "thr p99.9", np.quantile(t, .999), "pdf p99.9", np.quantile(pd, .999),
                  "mode", (gpu["v"]["mode"][m, k] == ora["v"]["mode"][m, k]).mean(), "shader", (gpu["v"]["shader"][m, k] == ora["v"]["shader"][m, k]).mean())
    m = same & (gpu["num_splats"] == ora["num_splats"]) & (ora["num_splats"] > 0)
    if m.sum():
        d = np.abs(gpu["splat"]["value"][m, 0] - ora["splat"]["value"][m, 0]) / np.maximum(1e-20, np.abs(ora["splat"]["value"][m, 0]))
        print("  splat p99", np.quantile(d, .99), "n", m.sum())
    if len(bad):
        k = bad[0]
        print("  path", k, gpu["length"][k], ora["length"][k])
        for v in range(min(8, max(gpu["length"][k], ora["length"][k]))):
            a, b = gpu["v"][k, v], ora["v"][k, v]
            print("   ", v, hex(int(a["prim"])), hex(int(b["prim"])), a["dist"], b["dist"], hex(int(a["mode"])), hex(int(b["mode"])), a["throughput"], b["throughput"], a["pdf"], b["pdf"])
    be.close()
    if True:
        mm = same & (ora["length"] > 1)
        dm = np.nonzero(mm & (gpu["v"]["mode"][:, 1] != ora["v"]["mode"][:, 1]))[0]
        for k in dm[:2]:
            print("  mode mismatch path", k, "len", gpu["length"][k], [hex(int(x)) for x in gpu["v"]["mode"][k][:4]], [hex(int(x)) for x in ora["v"]["mode"][k][:4]],
                  "thr", gpu["v"]["throughput"][k][:3], ora["v"]["throughput"][k][:3], "shader", ora["v"]["shader"][k][:3])
        nn = np.nonzero(np.isnan(gpu["splat"]["value"]).any(axis=1) | np.isnan(ora["splat"]["value"]).any(axis=1))[0]
        print("  nan splats: paths", nn[:5], "gpu", [gpu["splat"]["value"][k][:4] for k in nn[:2]], "ora", [ora["splat"]["value"][k][:4] for k in nn[:2]])
