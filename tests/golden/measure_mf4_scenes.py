#!/usr/bin/env python3
"""(build container) Image means of the reference built with -DMF_COUNT=4 (oracle/_ref/mf4/corona_pt_sfmt_mv8, `make -C oracle mf4`) on the extended
scenes -- media inside the glass sphere, global fog, nested media, a moving camera, moving geometry --: pt, sfmt generator, 256 x 256, 128 spp, depth 8,
three independent frames each -> tests/golden/mf4_scene_means.json. tests/test_gpu_hero.py holds the device's hero renders of the same scenes against
them: the end-to-end check of the HERO x MEDIA x MB kernels against that build's own images (the per-path pins are tests/test_oracle_hero.py)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import GOLD, REF, REPO, read_pfm, run_ref      # noqa: E402

SCENES = ["0055_media", "0056_fog", "0057_nested", "0058_cam_mb", "0059_mb", "0066_smooth"]
SPP, SIZE = 128, 256


def render(scene, frame):
    work, _ = run_ref("mf4/corona_pt_sfmt_mv8", 8, scene, ["-s", str(SPP), "--batch", str(SPP), "-w", str(SIZE), "-h", str(SIZE), "--frame", str(frame), "-x", "_m"],
                      env={"LD_LIBRARY_PATH": str(REF / "mf4" / "shaders")})
    img = read_pfm(next((work / "scenes" / scene).glob("*_m_fb00.pfm")))
    subprocess.run(["rm", "-rf", str(work)])
    return [float(x) for x in img.mean(axis=(0, 1))]


def main():
    subprocess.check_call(["make", "-C", str(REPO / "oracle"), "mf4"], stdout=subprocess.DEVNULL)
    out = {"what": f"image mean (XYZ of the PFM) of the reference built with MF_COUNT = 4: pt, sfmt, {SIZE}x{SIZE}, {SPP} spp, max depth 8, frames 1-3",
           "recipe": "make -C oracle mf4 && python3 tests/golden/measure_mf4_scenes.py", "spp": SPP, "size": SIZE, "scenes": {}}
    for scene in SCENES:
        frames = np.array([render(scene, f) for f in (1, 2, 3)])
        out["scenes"][scene] = {"frames": frames.tolist(), "mean": frames.mean(axis=0).tolist(), "sd_between_frames": frames.std(axis=0, ddof=1).tolist()}
        print(scene, out["scenes"][scene]["mean"], out["scenes"][scene]["sd_between_frames"], flush=True)
    (GOLD / "mf4_scene_means.json").write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
