#!/usr/bin/env python3
"""development helper (GPU box): time a few configurations of the kernel"""
import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tests"))
from helpers import *
pkg = load_pkg()
def run(name, sampler, mv, spp=16, scene_path=SCENE_0010):
    scene = make_scene(scene_path, width=1280, height=720, max_verts=mv, sampler=sampler)
    be = pkg.Backend(scene)
    per = scene.width * scene.height
    be.render(0, per); be.sync()
    c0 = be.counters()
    t0 = time.perf_counter()
    be.render(per, spp * per); be.sync()
    ms = (time.perf_counter() - t0) * 1e3
    c1 = be.counters()
    dc = [b - a for a, b in zip(c0, c1)]
    print(f"{name:28s} {spp*per/ms/1e3:9.1f} Msamples/s  {dc[0]/ms/1e3:9.1f} Mrays/s  rays/sample {dc[0]/dc[4]:.3f} nodes/ray {dc[1]/dc[0]:.2f} prims/ray {dc[3]/dc[0]:.2f}  wall {ms:.2f} ms  trace-kernel avg {be.last_kernel_ms():.3f} ms x {be.last_kernel_launches()}")
    be.close()
run("pt mv2 (camera+1 hit)", pkg.MI_SAMPLER_PT, 2)
run("pt mv3", pkg.MI_SAMPLER_PT, 3)
run("pt mv8 (cfg2)", pkg.MI_SAMPLER_PT, 8)
run("ptdl mv8 (cfg3)", pkg.MI_SAMPLER_PTDL, 8)
run("pt mv32 rough (cfg4)", pkg.MI_SAMPLER_PT, 32, scene_path=SCENE_ROUGH)
