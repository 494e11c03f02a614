"""development helper (GPU box): lane 0's ticks inside the traversal rounds of the wavefront kernel / the megakernel, -DMI_PROFILE_TRAV builds"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
be = pkg.Backend(scene, counters=False)
per = scene.width * scene.height
be.render(0, per); be.sync()
c0 = be.counters(); be.render(per, 16 * per); be.sync(); c1 = be.counters()
c = [b - a for a, b in zip(c0, c1)]
print(f"{be.kernel_name()[:16]} {be.last_kernel_ms():.2f} ms | ticks (M): node loop {c[0]/1e6:.0f} job set-up {c[1]/1e6:.0f} job passes {c[2]/1e6:.0f} owner epilogue {c[3]/1e6:.0f} | other {c[5]/1e6:.0f} {c[6]/1e6:.0f} | rounds/iterations {c[7]}")
