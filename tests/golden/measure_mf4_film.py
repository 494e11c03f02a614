#!/usr/bin/env python3
"""(build container) The converged image means bench.py holds its FOUR-WAVELENGTH renders against: the reference built with
-DMF_COUNT=4 (`make -C oracle mf4`) renders the bench film -- regression/0010_pt, 1280 x 720 (padded to 736), max depth 8 -- with the pt sampler
(cfg 2) and the ptdl sampler (cfg 3) at 512 spp, sfmt generator; the means (XYZ of the PFM it writes, the sidecar's "average image intensity") go to
tests/golden/mf4_film_means.json next to the scalar build's converged means (tests/golden/tilemeans_{pt,ptdl}_mv8.npz).
    make -C oracle mf4 && python3 tests/golden/measure_mf4_film.py [spp]"""
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import GOLD, REF, REPO, read_pfm, run_ref      # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    subprocess.check_call(["make", "-C", str(REPO / "oracle"), "mf4"], stdout=subprocess.DEVNULL)
    out = {"what": f"regression/0010_pt, 1280x720 (film 1280x736), max depth 8, sfmt, {spp} spp: image mean (XYZ of the PFM) of the reference built with MF_COUNT = 4",
           "recipe": "make -C oracle mf4 && python3 tests/golden/measure_mf4_film.py", "spp": spp}
    for cfg, binary, scalar in (("cfg2", "mf4/corona_pt_sfmt_mv8", "tilemeans_pt_mv8.npz"), ("cfg3", "mf4/corona_ptdl_sfmt_mv8", "tilemeans_ptdl_mv8.npz")):
        work, _ = run_ref(binary, 8, "0010_pt", ["-s", str(spp), "--batch", "16", "-w", "1280", "-h", "720", "-t", "8", "-x", "_mf4"],
                          env={"LD_LIBRARY_PATH": str(REF / "mf4" / "shaders")})
        img = read_pfm(work / "scenes" / "0010_pt" / "test_mf4_fb00.pfm")
        side = (work / "scenes" / "0010_pt" / "test_mf4_fb00.pfm.txt").read_text()
        m = re.search(r"elapsed wallclock prog ([\d.]+)s", side)
        sc = np.load(GOLD / scalar)
        out[cfg] = {"binary": "oracle/_ref/" + binary, "film": [int(img.shape[1]), int(img.shape[0])], "mean_xyz": [float(x) for x in img.astype(np.float64).mean(axis=(0, 1))],      # (in doubles: a float32 sum over 942 080 pixels is 0.15 % off)
                    "seconds": float(m.group(1)) if m else None,
                    "scalar_build_mean_xyz": [float(x) for x in sc["tiles"].astype(np.float64).mean(axis=(0, 1))], "scalar_build_spp": int(sc["spp"])}
        out[cfg]["mf4_over_scalar"] = [a / b for a, b in zip(out[cfg]["mean_xyz"], out[cfg]["scalar_build_mean_xyz"])]
        print(cfg, out[cfg], flush=True)
        shutil.rmtree(work, ignore_errors=True)
    (GOLD / "mf4_film_means.json").write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
