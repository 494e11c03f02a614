#!/usr/bin/env python3
"""(build container) Hero wavelengths, SURVEY 8(f) row 2: is the reference's MF_COUNT = 4 mode something a restatement could be pinned to?
`make -C oracle mf4` builds the reference with -DMF_COUNT=4 next to the usual MF_COUNT = 1 binaries (oracle/_ref/mf4/, the gcc errors of the two
BSDF plugins in compile_gcc.log); this script renders regression/0010_pt with both (sfmt generator, 256 x 256, 256 spp, max depth 8, three independent
frames each) and writes tests/golden/mf4_vs_mf1_measured.json: the compile record, the image means, their scatter and the ratio.
A hero-wavelength estimator has the same expectation as the single-wavelength one; the two binaries' means must agree to their noise."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import GOLD, REF, REPO, read_pfm, run_ref      # noqa: E402


def render(binary, shader_dir, frame, spp=256, size=256):
    work, _ = run_ref(binary, 8, "0010_pt", ["-s", str(spp), "--batch", str(spp), "-w", str(size), "-h", str(size), "--frame", str(frame), "-x", "_m"],
                      env={"LD_LIBRARY_PATH": str(shader_dir)})
    img = read_pfm(next((work / "scenes" / "0010_pt").glob("*_m_fb00.pfm")))
    return [float(x) for x in img.mean(axis=(0, 1))]


def main():
    subprocess.check_call(["make", "-C", str(REPO / "oracle"), "mf4"], stdout=subprocess.DEVNULL)
    out = {"what": "regression/0010_pt, pt, sfmt, 256x256, 256 spp, max depth 8: image mean (XYZ of the PFM) of the reference built with MF_COUNT = 1 and = 4",
           "recipe": "make -C oracle mf4 && python3 tests/golden/measure_mf4.py",
           "compile_gcc": (REF / "mf4" / "compile_gcc.log").read_text().splitlines()}
    for name, binary, shaders in (("mf1", "corona_pt_sfmt_mv8", REF / "shaders_mv8"), ("mf4", "mf4/corona_pt_sfmt_mv8", REF / "mf4" / "shaders")):
        out[name] = {"frames": {str(f): render(binary, shaders, f) for f in (1, 2, 3)}}
        m = np.array(list(out[name]["frames"].values()))
        out[name]["mean"] = [float(x) for x in m.mean(axis=0)]
        out[name]["sd_between_frames"] = [float(x) for x in m.std(axis=0, ddof=1)]
    r = np.array(out["mf4"]["mean"]) / np.array(out["mf1"]["mean"])
    out["mf4_over_mf1"] = [float(x) for x in r]
    sd = np.sqrt(np.array(out["mf1"]["sd_between_frames"]) ** 2 + np.array(out["mf4"]["sd_between_frames"]) ** 2) / np.sqrt(3)
    out["difference_in_sd_of_the_difference"] = [float(x) for x in (np.array(out["mf4"]["mean"]) - np.array(out["mf1"]["mean"])) / sd]
    (GOLD / "mf4_vs_mf1_measured.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
