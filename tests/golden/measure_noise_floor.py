#!/usr/bin/env python3
"""(build container) The regression thresholds (scenes/*/maxerror) from the reference's own noise, not from our renders.

For every scenes/NNNN_*/ with an `args` file: two independent renders a, b of the REAL reference binary (oracle/_ref, sfmt generator, frames 1 and 2)
with exactly the test's arguments. tests/regression_report.py compares a corona-mi render g at those arguments with a reference render r at four
times the samples; for an unbiased g

    E |g - r|^2 = sigma^2 (1 + 1/4)        and        E |a - b|^2 = 2 sigma^2        (sigma^2: per-pixel variance of one render, summed over channels)

so the expected rmse of the test is rmse(a, b) * sqrt(1.25 / 2) = 0.79 rmse(a, b); maxerror = 1.15 x that (the margin of
test_per_pixel_against_reference_render). Writes tests/golden/regression_noise_floor.json and the maxerror files.
The images are compared as the reference's regression scripts compare them (tools/img/pfmdiff.c:75-86)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import GOLD, REF, REPO, read_pfm, run_ref      # noqa: E402


def sampler_of(scene_dir):
    for line in (scene_dir / "config.mk").read_text().splitlines() if (scene_dir / "config.mk").exists() else []:
        if line.startswith("MOD_sampler="):
            return line.split("=", 1)[1].strip()
    return "pt"


def main():
    only = sys.argv[1:]
    fn = GOLD / "regression_noise_floor.json"
    out = json.loads(fn.read_text()) if fn.exists() else {}
    out["_rule"] = "maxerror = 1.15 * sqrt(1.25 / 2) * rmse(a, b); a, b: two independent renders of the reference binary at the test's own arguments"
    for d in sorted(p for p in (REPO / "scenes").iterdir() if p.is_dir() and (p / "args").exists() and (not only or p.name in only)):
        args = (d / "args").read_text().split()
        imgs = []
        for frame in (1, 2):
            work, _ = run_ref(f"corona_{sampler_of(d)}_sfmt_mv8", 8, d.name, args + ["--batch", "16", "--frame", str(frame), "-x", "_nf"])
            imgs.append(read_pfm(next((work / "scenes" / d.name).glob("*_nf_fb00.pfm"))).astype(np.float64))
            subprocess.run(["rm", "-rf", str(work)])
        a, b = imgs
        rmse_ab = float(np.sqrt(((a - b) ** 2).sum() / (a.shape[0] * a.shape[1])))
        expected = rmse_ab * float(np.sqrt(1.25 / 2.0))
        out[d.name] = {"args": " ".join(args), "rmse_ab": rmse_ab, "expected_rmse_of_the_test": expected, "maxerror": round(1.15 * expected, 3),
                       "mean_a": [float(x) for x in a.mean(axis=(0, 1))], "mean_b": [float(x) for x in b.mean(axis=(0, 1))]}
        (d / "maxerror").write_text("%.3f\n" % out[d.name]["maxerror"])
        print(d.name, out[d.name], flush=True)
        fn.write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
