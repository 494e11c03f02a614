"""development helper (GPU box): path records of the FAST rounds against the exact rounds of the same library, path by path
(python3 tests/dev/fast_vs_exact.py [paths] [pt|ptdl] [0010|media|fog|nested|cam_mb|rough|metal|fine] [rand|halton]) -- they must be identical; prints the paths that are not"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import numpy as np
from helpers import *
pkg = load_pkg()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
sampler = pkg.MI_SAMPLER_PTDL if len(sys.argv) > 2 and sys.argv[2] == "ptdl" else pkg.MI_SAMPLER_PT
which = sys.argv[3] if len(sys.argv) > 3 else "0010"
path = {"0010": SCENE_0010, "media": SCENE_MEDIA, "fog": SCENE_FOG, "nested": SCENE_NESTED, "cam_mb": SCENE_CAM_MB, "rough": SCENE_ROUGH, "metal": SCENE_METAL, "fine": SCENE_FINE}[which]
scene = make_scene(path, width=1280, height=720, max_verts=32 if which in ("rough", "nested") else 8, sampler=sampler,
                   pointsampler=pkg.MI_POINTS_HALTON if len(sys.argv) > 4 and sys.argv[4] == "halton" else pkg.MI_POINTS_RAND)
ex = pkg.Backend(scene, traversal="exact")
fa = pkg.Backend(scene, traversal="fast")
chunk, bad = 250000, 0
for first in range(777, 777 + total, chunk):
    a = ex.trace_paths(first, chunk)
    b = fa.trace_paths(first, chunk)
    same = a.tobytes() == b.tobytes()
    if same:
        continue
    diff = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
    for i in diff:
        bad += 1
        print("path", first + int(i), "length", a["length"][i], b["length"][i])
        for k in range(min(int(max(a["length"][i], b["length"][i])), 8)):
            va, vb = a["v"][i][k], b["v"][i][k]
            if va.tobytes() != vb.tobytes():
                print("  first differing vertex", k, "exact prim", va["prim"], "dist", va["dist"], "fast prim", vb["prim"], "dist", vb["dist"])
                print("   x", va["x"], vb["x"], "omega", va["omega"], vb["omega"])
                break
print(which, "ptdl" if sampler == pkg.MI_SAMPLER_PTDL else "pt", "paths compared", total, "differing", bad)
