"""development helper (GPU box): the rays of test_leaf_phase_corner_cases whose hit differs from the oracle's"""
import sys, pathlib, tempfile, numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import test_gpu_parity as T
from helpers import *
def cmp(scene, be, pos, d, ignore=None, max_dist=None):
    primid = np.ctypeslib.as_array(scene.desc.primid, shape=(scene.desc.num_prims,))
    ign_id = None if ignore is None else np.where(ignore == 0xffffffff, np.uint64(0xffffffffffffffff), primid[np.minimum(ignore, len(primid) - 1)])
    c0 = be.counters(); gpu = be.intersect(pos, d, ignore=ignore, max_dist=max_dist); c1 = be.counters()
    ora, cnt = oracle_intersect(scene, pos, d, ignore_primid=ign_id, max_dist=max_dist)
    bad = np.nonzero((gpu["primid"] != ora["prim"]) | (gpu["dist"].view(np.uint32) != ora["dist"].view(np.uint32)))[0]
    hit = gpu["primid"] != 0xffffffffffffffff
    tq = hit & ((gpu["primid"] >> np.uint64(61)) >= 3) & (gpu["primid"] == ora["prim"])
    baduv = np.nonzero(tq & ((gpu["u"].view(np.uint32) != ora["u"].view(np.uint32)) | (gpu["v"].view(np.uint32) != ora["v"].view(np.uint32))))[0]
    print("nodes", scene.desc.num_nodes, "rays", len(pos), "ignore", ignore is not None, "bad", len(bad), "bad uv", len(baduv), "counters", [c1[k] - c0[k] - cnt[k] for k in range(4)])
    for i in bad[:6]:
        print("   ray", i, "gpu %x %.9g | oracle %x %.9g" % (gpu["primid"][i], gpu["dist"][i], ora["prim"][i], ora["dist"][i]), "lane", i % 64)
    return gpu
T._compare_hits = cmp
T.test_leaf_phase_corner_cases(pathlib.Path(tempfile.mkdtemp()))
