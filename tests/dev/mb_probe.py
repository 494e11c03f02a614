"""development helper (GPU box): moving geometry (scenes/0059_mb) HIP vs oracle, the first paths that part ways"""
import sys
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from helpers import *
pkg = load_pkg()
sampler = pkg.MI_SAMPLER_PTDL if "ptdl" in sys.argv else pkg.MI_SAMPLER_PT
scene = make_scene(SCENE_MB, width=1280, height=720, max_verts=8, sampler=sampler)
be = pkg.Backend(scene)
n = 40000
gpu = be.trace_paths(12345, n)
ora = oracle_records(scene, 12345, n)
same = gpu["length"] == ora["length"]
for k in range(1, 8):
    m = ora["length"] > k
    same &= ~m | (gpu["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
bad = np.nonzero(~same)[0]
print(be.kernel_name(), "bad paths", len(bad), "of", n)
for i in bad[:6]:
    L = max(int(gpu["length"][i]), int(ora["length"][i]))
    print(" path", int(i), "len gpu/ora", int(gpu["length"][i]), int(ora["length"][i]), "time", float(ora["time"][i]))
    for k in range(1, min(L, 8)):
        print("   v%d prim %x / %x dist %.7g / %.7g" % (k, int(gpu["v"]["prim"][i, k]), int(ora["v"]["prim"][i, k]), float(gpu["v"]["dist"][i, k]), float(ora["v"]["dist"][i, k])))
be.close()
