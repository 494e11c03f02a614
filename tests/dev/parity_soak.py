"""development helper (GPU box): GPU path records against the oracle on millions of paths, in chunks
(python3 tests/dev/parity_soak.py [paths per configuration] [base|ext] [auto|exact|fast|hero]); hero: four wavelengths per path
(mi_scene_set_wavelengths) against oracle_hero_trace, the throughput deviation over all four components"""
import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import numpy as np
from helpers import *
pkg = load_pkg()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
chunk = 250000
CASES = {"base": (("cfg2 pt mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8), ("cfg3 ptdl mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8),
                  ("cfg4 rough mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 32), ("metal ptdl mv8", SCENE_METAL, pkg.MI_SAMPLER_PTDL, 8)),
         # the extended kernels (python3 tests/dev/parity_soak.py N ext): media, fog, nested media, moving camera, moving geometry, Halton
         "ext": (("media ptdl mv32", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 32), ("fog ptdl mv8", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 8),
                 ("nested pt mv32", SCENE_NESTED, pkg.MI_SAMPLER_PT, 32), ("cam mb ptdl mv8", SCENE_CAM_MB, pkg.MI_SAMPLER_PTDL, 8),
                 ("mb ptdl mv8", SCENE_MB, pkg.MI_SAMPLER_PTDL, 8), ("halton ptdl mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8),
                 ("mb light ptdl mv8", SCENE_MB_LIGHT, pkg.MI_SAMPLER_PTDL, 8), ("mb round ptdl mv8", SCENE_MB_ROUND, pkg.MI_SAMPLER_PTDL, 8),
                 ("mb round light ptdl mv8", SCENE_MB_ROUND_LIGHT, pkg.MI_SAMPLER_PTDL, 8),
                 ("halton all ptdl mv32", SCENE_ALL, pkg.MI_SAMPLER_PTDL, 32))}
for name, path, sampler, mv in CASES[sys.argv[2] if len(sys.argv) > 2 else "base"]:
    scene = make_scene(path, width=1280, height=720, max_verts=mv, sampler=sampler,
                       pointsampler=pkg.MI_POINTS_HALTON if name.startswith("halton") else pkg.MI_POINTS_RAND)
    mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
    hero = mode == "hero"
    be = pkg.Backend(scene) if mode in ("auto", "hero") else pkg.Backend(scene, traversal=mode)
    if hero:
        be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    name = f"{name} [{'hero' if hero else be.traversal()}]"
    n = same_len = same_prims = same_splats = 0
    worst_thr = 0.0
    t0 = time.time()
    for first in range(777, 777 + total, chunk):
        if hero:
            g, ge = be.trace_paths_hero(first, chunk)
            o, oe = oracle_hero_records(scene, first, chunk)
        else:
            g = be.trace_paths(first, chunk)
            o = oracle_records(scene, first, chunk)
        sl = g["length"] == o["length"]
        k = np.arange(8)[None, :]
        valid = k < np.minimum(o["length"], 8)[:, None]
        sp = sl & ((g["v"]["prim"] == o["v"]["prim"]) | ~valid).all(axis=1)
        ss = sp & (g["num_splats"] == o["num_splats"])
        n += chunk; same_len += int(sl.sum()); same_prims += int(sp.sum()); same_splats += int(ss.sum())
        m = sp[:, None] & valid & (k >= 1)
        thr_g, thr_o = (ge["throughput"][m], oe["throughput"][m]) if hero else (g["v"]["throughput"][m], o["v"]["throughput"][m])
        rel = np.abs(thr_g - thr_o) / np.maximum(1e-20, np.maximum(np.abs(thr_g), np.abs(thr_o)))
        worst_thr = max(worst_thr, float(np.quantile(rel, 0.9999)) if len(rel) else 0.0)
    be.close()
    print("%-24s %d paths: same length %d (%.5f %%), same primitive sequence %d (%.5f %%), and same splat count %d; 99.99th pct throughput deviation %.2e; %.0f s" %
          (name, n, same_len, 100.0 * same_len / n, same_prims, 100.0 * same_prims / n, same_splats, worst_thr, time.time() - t0), flush=True)
