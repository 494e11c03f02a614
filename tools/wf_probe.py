"""development helper (GPU box): what the waves of the wavefront kernel (csrc/mi_wavefront.h) do, from -DMI_PROFILE_WF=1|2|3 builds:
CORONA_MI_WAVEFRONT=1 CORONA_MI_LIB=.../libcorona_mi_wfp<k>.so python3 tools/wf_probe.py <k>"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
be = pkg.Backend(scene, counters=False)
per = scene.width * scene.height
be.render(0, per); be.sync()
c0 = be.counters(); be.render(per, 16 * per); be.sync(); c1 = be.counters()
c = [b - a for a, b in zip(c0, c1)]
paths = 16 * per
print(f"kernel {be.kernel_name()}: {be.last_kernel_ms():.2f} ms for 16 spp, paths counted {c[4]} of {paths}")
if mode == 1:
    print("shading passes %d with %.1f lanes | generating passes %d with %.1f lanes | queue turns of tracing waves %d taking %.1f rays each" %
          (c[0], c[1] / max(c[0], 1), c[2], c[3] / max(c[2], 1), c[5], c[6] / max(c[5], 1)))
elif mode == 2:
    names = ["shading", "generating", "queue turns", "traversal rounds", None, "waiting", "deciding"]
    tot = float(sum(c[k] for k in range(7) if k != 4))
    print(" | ".join(f"{n} {100 * c[k] / tot:.1f}%" for k, n in enumerate(names) if n))
else:
    print("traversal rounds %d with %.1f busy lanes | waits %d | partial shading passes %d | tracing episodes %d | lanes valid summed over queue turns %d" %
          (c[0], c[1] / max(c[0], 1), c[2], c[3], c[5], c[6]))
be.close()
