"""-m gpu: the wavefront kernel (corona-13_amd/csrc/mi_wavefront.h, CORONA_MI_WAVEFRONT=1) against the megakernel and the oracle.

Round 6's structural experiment: paths as entries of a per-workgroup table, rays and vertices as work items that the workgroup's waves
take in full batches (profiles/r06_wavefront.txt holds the measurements and why it is not the default). It runs the megakernel's own
device functions, so a path must come out the same whichever kernel renders it:
  * path records byte for byte the megakernel's (and through them the oracle's), launches of any size
  * the reference's traversal work counters in the counting instantiation
  * frames equal up to the order of the float atomics, every path counted
"""
import json

import numpy as np
import pytest

from helpers import GOLDEN, SCENE_0010, SCENE_FINE, SCENE_ROUGH, load_pkg, make_scene, oracle_records, oracle_render

pkg = load_pkg()
pytestmark = pytest.mark.gpu


def backends(monkeypatch, scene, **kw):
    monkeypatch.setenv("CORONA_MI_WAVEFRONT", "0")
    mega = pkg.Backend(scene, **kw)
    monkeypatch.setenv("CORONA_MI_WAVEFRONT", "1")
    wave = pkg.Backend(scene, **kw)
    monkeypatch.delenv("CORONA_MI_WAVEFRONT")
    assert mega.kernel_name().startswith("mi_path_kernel<") and wave.kernel_name().startswith("mi_wave_kernel<"), (mega.kernel_name(), wave.kernel_name())
    return mega, wave


@pytest.mark.parametrize("name,path,points,mv,n", [
    ("cfg2", SCENE_0010, "rand", 8, 400000),
    ("cfg4 rough dielectric, depth 32", SCENE_ROUGH, "rand", 32, 200000),
    ("cfg2 halton", SCENE_0010, "halton", 8, 200000),
    ("fine backdrop (tree read from L2)", SCENE_FINE, "rand", 8, 200000),
])
def test_records_are_the_megakernels(monkeypatch, name, path, points, mv, n):
    scene = make_scene(path, width=1280, height=720, max_verts=mv, pointsampler=pkg.MI_POINTS_HALTON if points == "halton" else pkg.MI_POINTS_RAND)
    mega, wave = backends(monkeypatch, scene)
    # launches smaller than a wave, than a workgroup, than the tables; indices beyond 2^32
    for first, count in ((5, n), (123456789, 1), (77, 63), (1000, 1000), (2 ** 33 + 9, 70000)):
        a, b = mega.trace_paths(first, count), wave.trace_paths(first, count)
        assert a.tobytes() == b.tobytes(), (name, first, count)
    if path == SCENE_0010 and points == "rand":
        ora = oracle_records(scene, 5, 50000)
        g = wave.trace_paths(5, 50000)
        same = (g["length"] == ora["length"]) & (g["num_splats"] == ora["num_splats"])
        for k in range(1, 8):
            sel = ora["length"] > k
            same &= ~sel | (g["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
        assert (~same).sum() <= 1, int((~same).sum())
    mega.close(); wave.close()


@pytest.mark.parametrize("counters", [True, False], ids=["counting", "production"])
def test_frames_and_work_counters(monkeypatch, counters):
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    mega, wave = backends(monkeypatch, scene, counters=counters)
    per = scene.width * scene.height
    frames = []
    for be in (mega, wave):
        c0 = be.counters()
        be.render(3 * per, 4 * per)
        frames.append(be.fb_read())
        c = [b - a for a, b in zip(c0, be.counters())]
        assert c[4] == 4 * per
        if counters:
            frames.append(c)
        else:
            assert c[:4] == [0, 0, 0, 0]
    if counters:
        fa, ca, fb, cb = frames
        assert ca[:7] == cb[:7], (ca, cb)          # rays, node visits, box hits, primitive tests, paths, splats, vertices: the same work
    else:
        fa, fb = frames
    assert np.abs(fa - fb).max() <= 2e-4 * np.abs(fa).max()
    mega.close(); wave.close()


def test_image_and_reference_counters_1spp(monkeypatch):
    """the wavefront kernel's own frame against the oracle's image of the same path indices, its work against the reference's -DACCEL_DEBUG totals"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    monkeypatch.setenv("CORONA_MI_WAVEFRONT", "1")
    be = pkg.Backend(scene, counters=True)
    n = scene.width * scene.height
    be.render(0, n)
    fb = be.fb_read()
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / n)
    assert rmse < 0.05, rmse
    cnt = be.counters()
    gold = json.loads((GOLDEN / "counters.json").read_text())["pt_mv8"]
    for k, key in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
        assert abs(cnt[k] - gold[key]) <= 3e-3 * gold[key], (key, cnt[k], gold[key])
        assert abs(cnt[k] - ocnt[k]) <= 1e-3 * ocnt[k], (key, cnt[k], ocnt[k])
    be.close()


def test_small_tables_and_other_samplers(monkeypatch):
    """256 entries per workgroup (every wave starves for entries: partial batches, waits) give the same records; a ptdl scene keeps the megakernel"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    monkeypatch.setenv("CORONA_MI_WAVEFRONT_ENTRIES", "256")
    mega, wave = backends(monkeypatch, scene)
    a, b = mega.trace_paths(11, 150000), wave.trace_paths(11, 150000)
    assert a.tobytes() == b.tobytes()
    mega.close(); wave.close()
    monkeypatch.delenv("CORONA_MI_WAVEFRONT_ENTRIES")
    monkeypatch.setenv("CORONA_MI_WAVEFRONT", "1")
    ptdl = pkg.Backend(make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL))
    assert ptdl.kernel_name().startswith("mi_path_kernel<")
    ptdl.close()
