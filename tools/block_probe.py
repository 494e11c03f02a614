"""development helper (GPU box): how full the wave is in the blocks of the shading code, from a -DMI_PROFILE_BLOCKS build
(CORONA_MI_LIB=.../libcorona_mi_blocks.so python3 tools/block_probe.py): lanes per execution of a block, executions per wave iteration"""
import sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
names = ["path_generate", "path_shade (any)", "surface vertex", "emitter hit + splat", "next event estimation", "sample diffuse", "sample dielectric", "sample metal"]
for name, sampler in (("pt", 0), ("ptdl", 1)):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
    be = pkg.Backend(scene, counters=False)
    per = scene.width * scene.height
    be.render(0, per); be.sync()
    c0 = be.counters(); be.render(per, 16 * per); be.sync(); c1 = be.counters()
    d = [b - a for a, b in zip(c0, c1)]
    lanes = [x & ((1 << 36) - 1) for x in d]; execs = [x >> 36 for x in d]
    print(f"{name} [{be.traversal()}]: " + " | ".join(f"{n}: {l / max(e, 1):.1f} lanes x {e / max(execs[1], 1):.2f} per shading pass" for n, l, e in zip(names, lanes, execs)))
    be.close()
