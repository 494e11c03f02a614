#!/usr/bin/env python3
"""kernel time of a hero (four wavelengths per path) render next to the scalar render of the same scene: tools/hero_time.py [--extended] [spp]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from helpers import SCENE_0010, SCENE_CAM_MB, SCENE_FOG, SCENE_MB, SCENE_MEDIA, SCENE_ROUGH, load_pkg, make_scene
pkg = load_pkg()
CASES = [("cfg2 pt", SCENE_0010, pkg.MI_SAMPLER_PT, 8), ("cfg3 ptdl", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8), ("cfg4 rough pt mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 32)]
if "--extended" in sys.argv:
    sys.argv.remove("--extended")
    CASES += [("media pt", SCENE_MEDIA, pkg.MI_SAMPLER_PT, 8), ("media ptdl", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 8), ("fog pt", SCENE_FOG, pkg.MI_SAMPLER_PT, 8),
              ("fog ptdl", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 8), ("moving camera pt", SCENE_CAM_MB, pkg.MI_SAMPLER_PT, 8), ("moving geometry pt", SCENE_MB, pkg.MI_SAMPLER_PT, 8)]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, path, sampler, mv in CASES:
    scene = make_scene(path, width=1280, height=720, max_verts=mv, sampler=sampler)
    be = pkg.Backend(scene, counters=False)
    n = scene.width * scene.height * spp
    out = []
    for wl in (1, 4):
        be.set_wavelengths(wl)
        ms = []
        for it in range(4):
            be.render(it * n, n); be.sync(); ms.append(be.last_kernel_ms())
        out.append(min(ms[1:]))
    print("%-20s scalar %.2f ms (%.0f Msamples/s)   hero %.2f ms (%.0f Mpaths/s, %.0f M wavelength samples/s)   x%.2f   %s" %
          (name, out[0], n / out[0] / 1e3, out[1], n / out[1] / 1e3, 4 * n / out[1] / 1e3, out[1] / out[0], be.kernel_name()))
    be.close()
