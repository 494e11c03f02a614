"""development helper (GPU box): device BVH build against the host (reference-style SAH) build at growing primitive counts.
Backdrops of 4096 * k^2 quads are generated with tools/make_geo.py into a scratch directory. MI_SCALE_SAH="0 2": the device build once per
number of rotation passes (CORONA_MI_BUILD_SAH); default: the library's default only."""
import os, shutil, subprocess, sys, tempfile, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests"))
import numpy as np
from helpers import load_pkg
pkg = load_pkg()
work = Path(tempfile.mkdtemp(prefix="mi_scale_"))
shutil.copytree(REPO / "scenes", work / "scenes")
ks = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
for k in ks:
    geo = work / "scenes" / "geo" / f"plane_k{k}.geo"
    if k == 1:
        shutil.copy(REPO / "scenes" / "geo" / "plane.geo", geo)
    else:
        subprocess.check_call([sys.executable, str(REPO / "tools" / "make_geo.py"), "subdivide", str(REPO / "scenes" / "geo" / "plane.geo"), str(geo), str(k)])
    sdir = work / "scenes" / f"k{k}"
    sdir.mkdir()
    shutil.copy(REPO / "scenes" / "0010_pt" / "test01.cam", sdir)
    (sdir / "test.nra2").write_text((REPO / "scenes" / "0010_pt" / "test.nra2").read_text().replace("2 ../geo/plane\n", f"2 ../geo/plane_k{k}\n"))
    t0 = time.perf_counter()
    scene = pkg.Scene(sdir / "test.nra2", width=1280, height=720, max_verts=8)
    t1 = time.perf_counter()
    host = pkg.Backend(scene); host.sync()
    t2 = time.perf_counter()
    rng = np.random.default_rng(1)
    n = 100000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a = host.intersect(pos, d)
    per = scene.width * scene.height

    def measure(be):
        be.render(0, per); be.sync()
        c0 = be.counters(); t = time.perf_counter(); be.render(per, 8 * per); be.sync(); ms = (time.perf_counter() - t) * 1e3; c1 = be.counters()
        dc = [y - x for x, y in zip(c0, c1)]
        return (8 * per / ms / 1e3, dc[1] / dc[0], dc[3] / dc[0])
    res_host = measure(host)
    print("%8d prims: host load+SAH build %8.1f ms | backend with host tree %7.1f ms | render host tree %7.1f Msamples/s (%.2f nodes, %.2f prims per ray)" %
          (scene.desc.num_prims, (t1 - t0) * 1e3, (t2 - t1) * 1e3, *res_host), flush=True)
    host.close()
    for sah in (os.environ.get("MI_SCALE_SAH", "").split() or [None]):
        if sah is not None:
            os.environ["CORONA_MI_BUILD_SAH"] = sah
        t2 = time.perf_counter()
        devb = pkg.Backend(scene, device_build=True); devb.sync()
        t3 = time.perf_counter()
        b = devb.intersect(pos, d)
        same = a["primid"] == b["primid"]
        print("%8s        device build%s: backend %7.1f ms, %d nodes | same hits %.6f | render %7.1f Msamples/s (%.2f nodes, %.2f prims per ray)" %
              ("", "" if sah is None else " sah=" + sah, (t3 - t2) * 1e3, devb.stats()["nodes"], same.mean(), *measure(devb)), flush=True)
        devb.close()
shutil.rmtree(work, ignore_errors=True)
