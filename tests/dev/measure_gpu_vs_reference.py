#!/usr/bin/env python3
"""(GPU box) what the HIP kernels achieve DIRECTLY against the reference's own path dumps (tests/golden/paths_*.npz, written by the real reference
binary): agreeing path lengths, splat counts, primitives per vertex, splat energy. Writes tests/golden/gpu_vs_reference_measured.json, from which
tests/test_gpu_parity.py::test_paths_match_reference_golden takes its bounds (measurement plus a margin, as tests/test_oracle_golden.py does with
oracle_vs_reference_measured.json) -- instead of round numbers that say nothing about what the build reaches."""
import json
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from helpers import *

pkg = load_pkg()
CASES = [("pt_mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8, pkg.MI_POINTS_RAND), ("ptdl_mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, pkg.MI_POINTS_RAND),
         ("rough_mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 32, pkg.MI_POINTS_RAND), ("halton_pt_mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8, pkg.MI_POINTS_HALTON),
         ("halton_ptdl_mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, pkg.MI_POINTS_HALTON)]


def measure(ref, gpu):
    same = gpu["length"] == ref["length"]
    m = {"paths": int(len(ref)), "same_length": float(same.mean()), "same_splats": float((gpu["num_splats"] == ref["num_splats"]).mean())}
    worst = 1.0
    for k in range(1, 8):
        sel = same & (ref["length"] > k)
        if sel.sum():
            worst = min(worst, float((gpu["v"]["prim"][sel, k] == ref["v"]["prim"][sel, k]).mean()))
    m["worst_same_prim"] = worst
    e_ref, e_gpu = np.nan_to_num(ref["splat"]["col"]).sum(axis=(0, 1)), np.nan_to_num(gpu["splat"]["col"]).sum(axis=(0, 1))
    m["energy_dev"] = float(np.abs(e_ref - e_gpu).max() / np.abs(e_ref).max())
    return m


def main():
    out = {}
    for name, path, sampler, mv, points in CASES:
        fn = GOLDEN / f"paths_{name}.npz"
        if not fn.exists():
            continue
        g = np.load(fn)
        ref = g["records"]
        scene = make_scene(path, width=int(g["width"]), height=int(g["height"]), max_verts=mv, sampler=sampler, pointsampler=points)
        for traversal in ("exact", "fast"):
            be = pkg.Backend(scene, traversal=traversal)
            out[f"{name}@{traversal}"] = measure(ref, be.trace_paths(0, len(ref)))
            be.close()
            print(name, traversal, out[f"{name}@{traversal}"], flush=True)
    (GOLDEN / "gpu_vs_reference_measured.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
