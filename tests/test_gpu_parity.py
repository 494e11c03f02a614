"""-m gpu: the HIP path (through the C ABI) against the oracle on the same path indices, plus
size-independent properties at BASELINE.json's full size.

Tolerances (float32, fp contraction off on both sides, libm differences in sin/cos/atan2/acos/pow only):
  * camera sample exact; identical vertex count and primitive sequence for >= 99.9 % of paths
  * per-vertex throughput relative deviation: 99.9th percentile < 1e-3
  * 1-spp image vs the oracle image: per-pixel L2 (pfmdiff RMSE on gain-scaled XYZ) < 0.05
    (noise floor between two independent 64-spp renders of the reference itself: 4.34)
"""
import ctypes as C
import json

import numpy as np
import pytest

from helpers import oracle_lib, GOLDEN, SCENE_0010, SCENE_ALL, SCENE_CAM_MB, SCENE_FINE, SCENE_LARGE, SCENE_FOG, SCENE_MB, SCENE_MB_LIGHT, SCENE_MB_ROUND, SCENE_MB_ROUND_LIGHT, SCENE_MEDIA, SCENE_NESTED, SCENE_METAL, SCENE_ROUGH, load_pkg, make_scene, oracle_intersect, oracle_records, oracle_render, oracle_render_tiles, oracle_pixels

pkg = load_pkg()
pytestmark = pytest.mark.gpu


@pytest.fixture(params=["exact", "fast"])
def traversal(request):
    """both traversal modes of the backend (corona_mi.h: MI_TRAVERSAL_EXACT keeps the reference's order of operations and with it
    its work counters; MI_TRAVERSAL_FAST, the library's default and what bench.py times, puts leaves aside while a lane descends
    on): hits, paths and images must be the same, only the work counters may differ"""
    return request.param


@pytest.fixture(params=[True, False], ids=["counting", "production"])
def counters(request):
    """the counting (COUNT = true) and the production (COUNT = false, what bench.py and the CLI launch) instantiations"""
    return request.param


def uses_moving_primitives(name):
    return name.startswith("moving geometry") or "everything at once" in name


def check_work(cnt, ocnt, traversal, tol=1e-3, keys=(0, 1, 2, 3)):
    """traversal work against the oracle's (= the reference's -DACCEL_DEBUG semantics): equal in exact mode; in fast mode the same
    rays, and never less work than the reference's traversal needs -- but not much more either (speculative visits)"""
    for k in keys:
        if traversal == "exact" or k == 0:
            assert abs(cnt[k] - ocnt[k]) <= tol * ocnt[k], (k, cnt[k], ocnt[k])
        else:
            assert (1 - tol) * ocnt[k] <= cnt[k] <= 1.2 * ocnt[k], (k, cnt[k], ocnt[k])      # measured: +7 % node visits, +12 % primitive tests


def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))


CASES = [
    ("cfg2 pt 1280x720 mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 1280, 720, 8, 100000),
    ("cfg1 pt 256x256 mv4", SCENE_0010, pkg.MI_SAMPLER_PT, 256, 256, 4, 8000),
    ("cfg3 ptdl 1280x720 mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 60000),
    ("cfg4 rough dielectric mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 1280, 720, 32, 30000),
    ("fine backdrop (tree in HBM) pt mv8", SCENE_FINE, pkg.MI_SAMPLER_PT, 1280, 720, 8, 30000),
    ("fine backdrop (tree in HBM) ptdl mv8", SCENE_FINE, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 10000),
    ("metal pt mv8", SCENE_METAL, pkg.MI_SAMPLER_PT, 1280, 720, 8, 8000),
    ("metal ptdl mv8", SCENE_METAL, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 8000),
    # row a19 in the reference BUILD's behaviour (mi_scene_set_metal_reference / oracle_set_reference_metal: the NaN of its compiled
    # Fresnel term ends 2-4 % of the samples at a gold vertex, src/shaders/metal.c:79-157): the same predicate on both sides, path for path
    ("metal pt mv8, reference build", SCENE_METAL, pkg.MI_SAMPLER_PT, 1280, 720, 8, 60000),
    ("metal ptdl mv8, reference build", SCENE_METAL, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 40000),
    # homogeneous medium inside the glass sphere (`interior`, `medium_rgb`): free-flight sampling, volume
    # vertices, Henyey-Greenstein, transmittance and volume pdfs in next event estimation / MIS; depth 32 = long random walks
    ("media pt mv8", SCENE_MEDIA, pkg.MI_SAMPLER_PT, 1280, 720, 8, 60000),
    ("media ptdl mv8", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 40000),
    ("media pt mv32", SCENE_MEDIA, pkg.MI_SAMPLER_PT, 1280, 720, 32, 20000),
    ("media ptdl mv32", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 1280, 720, 32, 20000),
    # thin global fog (`exterior <medium> 0`): volume vertices in the open, next event estimation from them reaches the emitters
    ("fog pt mv8", SCENE_FOG, pkg.MI_SAMPLER_PT, 1280, 720, 8, 40000),
    ("fog ptdl mv8", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 40000),
    ("fog ptdl mv32", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 1280, 720, 32, 10000),
    # fog outside + scattering medium in the sphere + absorbing medium in the cone
    ("nested media pt mv8", SCENE_NESTED, pkg.MI_SAMPLER_PT, 1280, 720, 8, 40000),
    ("nested media ptdl mv32", SCENE_NESTED, pkg.MI_SAMPLER_PTDL, 1280, 720, 32, 10000),
    # camera motion blur: the camera frame is interpolated per path (acosf / sinf differ in the last ulp between host and device)
    ("camera motion blur pt mv8", SCENE_CAM_MB, pkg.MI_SAMPLER_PT, 1280, 720, 8, 40000),
    ("camera motion blur ptdl mv8", SCENE_CAM_MB, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 20000),
    # motion-blurred geometry: 4096 backdrop quads and the cylinder cap move during the exposure
    ("moving geometry pt mv8", SCENE_MB, pkg.MI_SAMPLER_PT, 1280, 720, 8, 40000),
    ("moving geometry ptdl mv8", SCENE_MB, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 20000),
    ("moving geometry and emitter ptdl mv8", SCENE_MB_LIGHT, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 30000),
    ("moving geometry: sphere, cone, cylinder pt mv8", SCENE_MB_ROUND, pkg.MI_SAMPLER_PT, 1280, 720, 8, 40000),
    ("moving geometry: sphere, cone, cylinder ptdl mv8", SCENE_MB_ROUND, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 20000),
    ("moving sphere and cone as emitters pt mv8", SCENE_MB_ROUND_LIGHT, pkg.MI_SAMPLER_PT, 1280, 720, 8, 20000),
    ("moving sphere and cone as emitters ptdl mv8", SCENE_MB_ROUND_LIGHT, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 20000),
    # MOD_pointsampler = halton; ptdl at depth 32 reaches dimensions >= 256 (generator fall-back)
    ("halton pt mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 1280, 720, 8, 60000),
    ("halton ptdl mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 40000),
    ("halton ptdl rough dielectric mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PTDL, 1280, 720, 32, 30000),
    ("halton fine backdrop (tree in HBM) pt mv8", SCENE_FINE, pkg.MI_SAMPLER_PT, 1280, 720, 8, 10000),
    ("halton fog ptdl mv8 (free-flight dimension from the sampler)", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 20000),
    ("halton nested media pt mv32", SCENE_NESTED, pkg.MI_SAMPLER_PT, 1280, 720, 32, 10000),
    # every feature in one scene: the RECORD x PTDL x HALTON x MEDIA x MB instantiations
    ("halton everything at once ptdl mv8", SCENE_ALL, pkg.MI_SAMPLER_PTDL, 1280, 720, 8, 30000),
    ("moving geometry: everything at once pt mv32", SCENE_ALL, pkg.MI_SAMPLER_PT, 1280, 720, 32, 10000),
]


def points_of(name):
    return pkg.MI_POINTS_HALTON if name.startswith("halton") else pkg.MI_POINTS_RAND


@pytest.mark.parametrize("name,scene_path,sampler,w,h,mv,n", CASES)
def test_paths_match_oracle(name, scene_path, sampler, w, h, mv, n, traversal):
    if traversal == "fast" and uses_moving_primitives(name):
        pytest.skip("scenes with moving primitives always run the exact rounds")
    scene = make_scene(scene_path, width=w, height=h, max_verts=mv, sampler=sampler, pointsampler=points_of(name))
    be = pkg.Backend(scene, traversal=traversal)
    first = 12345
    reference_build = name.endswith("reference build")
    if reference_build:
        be.set_metal_reference(True)
        oracle_lib().oracle_set_reference_metal(1)
    try:
        gpu = be.trace_paths(first, n)
        ora = oracle_records(scene, first, n)
    finally:
        oracle_lib().oracle_set_reference_metal(0)
    if reference_build:
        # the switch does something: paths end at metal vertices that the formula as written lets go on
        be.set_metal_reference(False)
        plain = be.trace_paths(first, n)
        assert 0.005 * n < (plain["length"] != gpu["length"]).sum() < 0.1 * n
    assert np.array_equal(gpu["index"], ora["index"])
    for f in ("pixel_i", "pixel_j", "lambda", "time", "scramble"):
        assert np.abs(gpu[f] - ora[f]).max() <= 1e-5, f
    # what the build achieves (99.9995 % identical paths on 20 M-path soaks), not a loose bound: at most
    # 2 paths in 100 000 may part ways with the oracle (grazing hits decided by the last ulp of the device libm's sinf / atan2f)
    allowed = max(2, int(np.ceil(2e-5 * n)))
    if reference_build:
        # the reference build's NaN is the SIGN of a rounding error of cos theta (oracle/oracle_shade.c:638-663): where the device's cosine
        # differs from the host's in the last bit (sinf / cosf of the two libms, 1 ulp apart on a few per cent of their arguments) the verdict
        # is a coin both sides toss. Measured: 27 of 60 000 (pt), i.e. 4.5e-4; every other field of the agreeing paths is held to the usual bounds
        allowed = int(np.ceil(1e-3 * n))
    same = gpu["length"] == ora["length"]
    assert (~same).sum() <= allowed, (~same).sum()
    # a path's splat count also changes when ONE of its next-event connections flips (a shadow ray grazing an edge, a weight at the
    # float underflow limit): up to 7 connections per path at depth 8, 31 at depth 32, most of them in the media scenes
    allowed_splats = allowed if sampler == pkg.MI_SAMPLER_PT else int(np.ceil((2e-4 if mv <= 8 else 1e-3) * n))
    assert (gpu["num_splats"] != ora["num_splats"]).sum() <= allowed_splats
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if not m.sum():
            continue
        assert (gpu["v"]["prim"][m, k] != ora["v"]["prim"][m, k]).sum() <= allowed
        dx = np.abs(gpu["v"]["x"][m, k] - ora["v"]["x"][m, k]).max(axis=1)
        # positions drift with every glossy bounce (libm sin/cos/atan2 differ in the last ulp between host and device)
        # (from the fifth vertex on only a few hundred of the paths are left and the 99.9th percentile is the single most chaotic one --
        # a path bouncing around inside the glass sphere amplifies a last-ulp difference a thousandfold: the 99th is asserted there)
        # Round 4: the 99.9th percentile at every vertex; where fewer than 3000 paths are left, at most three of them may lie outside
        # (it used to be the 99th percentile for every case from the fifth vertex on: ten times as many outliers as the chaotic path needs)
        outliers = max(3, int(np.ceil(1e-3 * m.sum())))
        assert (dx >= (2e-3 if k <= 2 else 1e-2)).sum() <= outliers
        # a moving camera's frame comes out of acosf / sinf per path: the last-ulp libm difference sits on every vertex from the start
        assert (rel(gpu["v"]["throughput"][m, k], ora["v"]["throughput"][m, k]) >= (2e-2 if name.startswith(("camera motion blur", "moving geometry")) else 1e-3)).sum() <= outliers
        assert (gpu["v"]["flags"][m, k] != ora["v"]["flags"][m, k]).sum() <= allowed
        assert (gpu["v"]["mode"][m, k] != ora["v"]["mode"][m, k]).sum() <= allowed
        assert (gpu["v"]["shader"][m, k] != ora["v"]["shader"][m, k]).sum() <= allowed
        assert (rel(gpu["v"]["pdf"][m, k], ora["v"]["pdf"][m, k]) >= (2e-2 if name.startswith(("camera motion blur", "moving geometry")) else 5e-3)).sum() <= outliers
    m = same & (gpu["num_splats"] == ora["num_splats"]) & (ora["num_splats"] > 0)
    if m.sum():
        a, b = gpu["splat"]["value"][m, 0], ora["splat"]["value"][m, 0]
        fin = np.isfinite(a) & np.isfinite(b)     # deep ptdl paths: the reference's own MIS products overflow to NaN, on both sides alike
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.quantile(rel(a[fin], b[fin]), 0.99) < 1e-3
    be.close()


def test_soak_two_million_paths(traversal):
    """a 2 M-path slice of tests/dev/parity_soak.py inside the suite: cfg 2 (pt, depth 8) against the oracle, chunk by chunk --
    primitive sequence, vertex count and splat count of every path; at most 2 in 100 000 may differ (measured: 5 in a million)"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    be = pkg.Backend(scene, traversal=traversal)
    total, chunk, bad = 2000000, 250000, 0
    worst = 0.0
    for first in range(777, 777 + total, chunk):
        g = be.trace_paths(first, chunk)
        o = oracle_records(scene, first, chunk)
        k = np.arange(8)[None, :]
        valid = k < np.minimum(o["length"], 8)[:, None]
        ok = (g["length"] == o["length"]) & ((g["v"]["prim"] == o["v"]["prim"]) | ~valid).all(axis=1) & (g["num_splats"] == o["num_splats"])
        bad += int((~ok).sum())
        m = ok[:, None] & valid & (k >= 1)
        worst = max(worst, float(np.quantile(rel(g["v"]["throughput"][m], o["v"]["throughput"][m]), 0.9999)))
    be.close()
    assert bad <= 2e-5 * total, bad
    assert worst < 1e-3, worst


with open(GOLDEN / "gpu_vs_reference_measured.json") as _f:
    GPU_VS_REFERENCE = json.load(_f)          # written by tests/dev/measure_gpu_vs_reference.py on the GPU box (committed next to the goldens)


@pytest.mark.parametrize("name,scene_path,sampler,mv,points", [
    ("pt_mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8, pkg.MI_POINTS_RAND), ("ptdl_mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, pkg.MI_POINTS_RAND),
    ("rough_mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 32, pkg.MI_POINTS_RAND), ("halton_ptdl_mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, pkg.MI_POINTS_HALTON)])
def test_paths_match_reference_golden(name, scene_path, sampler, mv, points, traversal):
    """directly against the records dumped from the real reference. The bounds are what the build was MEASURED to reach against these very
    dumps (tests/golden/gpu_vs_reference_measured.json) plus a margin -- a quarter of the measured miss, at least one path in 5000 -- not round
    numbers: pt 3000 of 3000 identical, ptdl 2999 of 3000 (one grazing next-event connection), depth 32 1998 of 2000."""
    g = np.load(GOLDEN / f"paths_{name}.npz")
    ref = g["records"]
    was = GPU_VS_REFERENCE[f"{name}@{traversal}"]

    def bound(measured):
        return 1.0 - 1.25 * (1.0 - measured) - 2e-4
    scene = make_scene(scene_path, width=int(g["width"]), height=int(g["height"]), max_verts=mv, sampler=sampler, pointsampler=points)
    be = pkg.Backend(scene, traversal=traversal)
    gpu = be.trace_paths(0, len(ref))
    same = gpu["length"] == ref["length"]
    assert same.mean() >= bound(was["same_length"]), (same.mean(), was["same_length"])
    assert (gpu["num_splats"] == ref["num_splats"]).mean() >= bound(was["same_splats"])
    for k in range(1, 8):
        m = same & (ref["length"] > k)
        if m.sum():
            assert (gpu["v"]["prim"][m, k] == ref["v"]["prim"][m, k]).mean() >= bound(was["worst_same_prim"])
    e_ref, e_gpu = np.nan_to_num(ref["splat"]["col"]).sum(axis=(0, 1)), np.nan_to_num(gpu["splat"]["col"]).sum(axis=(0, 1))
    assert np.abs(e_ref - e_gpu).max() / np.abs(e_ref).max() <= 1.5 * was["energy_dev"] + 1e-5
    be.close()


def test_image_matches_oracle_1spp(traversal, counters):
    """configs[1]'s film at one sample per pixel against the oracle's image of the same path indices -- for the counting kernels and
    for the production kernels (COUNT = false: the instantiation bench.py times), in both traversal modes"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    be = pkg.Backend(scene, traversal=traversal, counters=counters)
    n = scene.width * scene.height
    be.render(0, n)
    fb = be.fb_read()
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    gain = scene.gain(1)
    rmse = np.sqrt((((fb - ofb) * gain) ** 2).sum() / n)           # tools/img/pfmdiff.c:75-86
    assert rmse < 0.05, rmse
    assert np.allclose(fb.sum(axis=(0, 1)), ofb.sum(axis=(0, 1)), rtol=1e-3)
    cnt = be.counters()
    assert cnt[4] == n
    if counters:
        # same traversal work as the oracle (and through it the reference's -DACCEL_DEBUG counters) in exact mode
        check_work(cnt, ocnt, traversal)
        gold = json.loads((GOLDEN / "counters.json").read_text())["pt_mv8"]
        if traversal == "exact":
            for k, key in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
                assert abs(cnt[k] - gold[key]) <= 3e-3 * gold[key], (key, cnt[k], gold[key])
    else:
        assert cnt[:4] == [0, 0, 0, 0]                 # the production kernels count paths only
    be.close()


@pytest.mark.parametrize("sampler", ["pt", "ptdl"])
def test_film_pixel_by_pixel(sampler):
    """the film itself, float by float: the oracle adds every one of the 4 x 4 taps of a splat as the reference does (include/filter/blackmanharris.h:63-72),
    the kernels leave out the taps whose window weight is exactly 0 (MI_SPLAT_SKIP_ZERO, csrc/mi_kernels.h) -- an addition of 0.0. At one sample per pixel the
    two films touch the SAME floats (a tap left out by mistake, or one added beyond the window, would show here) and agree to the rounding of a window weight
    next to its zero (6e-5 left of a sum of terms of 0.1-0.5: a last-bit difference of a cosine is 1e-4 of it; measured with tests/dev/film_probe.py:
    no pixel beyond 1e-3 for pt, 331 for ptdl, whose any-hit shadow rays may graze the emitter's edge the other way)."""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL if sampler == "ptdl" else pkg.MI_SAMPLER_PT)
    n = scene.width * scene.height
    be = pkg.Backend(scene, counters=False)
    be.render(0, n)
    fb = be.fb_read().astype(np.float64)
    be.close()
    ofb = oracle_render(scene, 0, n, threads=8)[0].astype(np.float64)
    rel = np.abs(fb - ofb) / np.maximum(np.maximum(np.abs(fb), np.abs(ofb)), 1e-30)
    off = int((rel > 1e-3).any(axis=2).sum())
    assert off <= (1000 if sampler == "ptdl" else 48), off
    same_floats = ((fb == 0) == (ofb == 0)).mean()
    assert same_floats >= (0.9999 if sampler == "ptdl" else 0.99999) and (fb != 0).mean() > 0.01, same_floats


def test_ptdl_image_matches_oracle_1spp(monkeypatch, traversal, counters):
    """BASELINE config 3 (0011_ptdl: next event estimation + shadow rays). Shadow rays towards the (planar quad) emitter stop at
    the first occluder by default (MI_LIGHT_ANYHIT, mi_device.h): same image and splats as the oracle's closest-hit traversal,
    fewer node visits; CORONA_MI_SHADOW=closest runs the reference's traversal, whose counters then equal the oracle's and,
    through it, the reference's own -DACCEL_DEBUG totals."""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    n = scene.width * scene.height
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    gain = scene.gain(1)
    gold = json.loads((GOLDEN / "counters.json").read_text())["ptdl_mv8"]
    counts = {}
    for mode in ("anyhit", "closest"):
        if mode == "closest":
            monkeypatch.setenv("CORONA_MI_SHADOW", "closest")
        be = pkg.Backend(scene, traversal=traversal, counters=counters)
        be.render(0, n)
        fb = be.fb_read()
        rmse = np.sqrt((((fb - ofb) * gain) ** 2).sum() / n)
        assert rmse < 0.5, rmse                       # a handful of shadow rays grazing the emitter edge may flip
        assert np.allclose(fb.sum(axis=(0, 1)), ofb.sum(axis=(0, 1)), rtol=2e-3)
        cnt = counts[mode] = be.counters()
        assert cnt[4] == n
        if counters:
            assert abs(cnt[5] - ocnt[5]) <= 2e-3 * ocnt[5]             # splats
            check_work(cnt, ocnt, traversal, tol=2e-3, keys=(0,) if mode == "anyhit" else (0, 1, 2, 3))   # rays always; the traversal work in closest-hit mode
        if mode == "closest":
            if counters and traversal == "exact":
                for k, key in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
                    assert abs(cnt[k] - gold[key]) <= 3e-3 * gold[key], (key, cnt[k], gold[key])
        else:
            # 64 spp mean vs the reference's own 64-spp ptdl value (1.0743, 1.0716, 1.0624), BASELINE.md
            be.fb_clear()
            be.render(0, 64 * n)
            mean = (be.fb_read() * scene.gain(64)).mean(axis=(0, 1))
            assert np.all(np.abs(mean - np.array([1.0743, 1.0716, 1.0624])) < 0.01), mean
        be.close()
    if counters:
        assert counts["anyhit"][0] == counts["closest"][0] and counts["anyhit"][5] == counts["closest"][5]      # same rays, same splats
        if traversal == "exact":
            assert counts["anyhit"][1] < counts["closest"][1] and counts["anyhit"][3] < counts["closest"][3]    # ... for less traversal work


def test_full_size_properties_cfg2():
    """1280x720, 64 spp, max depth 8 (BASELINE config 2) with the kernels bench.py times (production instantiation, the library's
    default traversal): properties that do not need the oracle, and the 2048-spp reference render at matched sample count."""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    be = pkg.Backend(scene, counters=False)
    per = scene.width * scene.height
    spp = 64
    be.render(0, spp * per)
    full = be.fb_read()
    cnt = be.counters()
    assert cnt[4] == spp * per
    assert np.isfinite(full).all() and (full >= 0).all()
    # additivity over index ranges: one launch == many launches (up to float atomic order)
    be.fb_clear()
    for s in range(0, spp, 16):
        be.render(s * per, 16 * per)
    parts = be.fb_read()
    assert np.allclose(full.sum(axis=(0, 1)), parts.sum(axis=(0, 1)), rtol=2e-5)
    gain = scene.gain(spp)
    assert np.sqrt((((full - parts) * gain) ** 2).sum() / per) < 1e-3
    # mean image vs the reference's own 64-spp value (1.0675, 1.0683, 1.0539 +- noise)
    mean = (full * gain).mean(axis=(0, 1))
    assert np.all(np.abs(mean - np.array([1.0675, 1.0683, 1.0539])) < 0.02), mean
    fn = GOLDEN / "tilemeans_pt_mv8.npz"
    if fn.exists():
        g = np.load(fn)
        tiles = (full * gain).reshape(scene.height // 32, 32, scene.width // 32, 32, 3).mean(axis=(1, 3))
        assert np.all(np.abs(mean - g["mean"]) < 0.015), (mean, g["mean"])
        # per-tile agreement with the 2048-spp reference render: at 64 spp a 32x32 tile of pure pt still carries
        # ~20 % noise (0.5 % of the paths find the 0.48 dm^2 emitter), so compare the luminance maps as a whole
        a, b = tiles[..., 1].ravel(), g["tiles"][..., 1].ravel()
        assert np.corrcoef(a, b)[0, 1] > 0.9
        assert abs(a.sum() / b.sum() - 1) < 0.02
        # the same 2048 spp as the reference render, as eight independent parts whose scatter measures the noise of this estimator on
        # this film tile by tile (pure pt: heavy-tailed, 0.5 % of the paths find the emitter, and the dark upper third of the film
        # carries +-50 % per tile even at 2048 spp). z = (gpu - ref) / sqrt(2 var) must then look like unit noise: centred, unit
        # robust width (the plain variance of a heavy-tailed sample is dominated by a handful of fireflies -- measured mean z^2 1.5
        # at robust sigma 1.07 --, so width and tails are asserted separately); and the image means within 0.3 %
        rspp, Q = int(g["spp"]), 8
        parts = []
        for q in range(Q):
            be.fb_clear()
            be.render((5000 + q * rspp // Q) * per, rspp // Q * per)
            parts.append((be.fb_read() * scene.gain(rspp // Q)).reshape(scene.height // 32, 32, scene.width // 32, 32, 3).mean(axis=(1, 3)))
        parts = np.array(parts)
        gpu_tiles = parts.mean(axis=0)
        var = parts.var(axis=0, ddof=1) / Q                       # variance of a 2048-spp tile mean; the reference's is the same
        z = (gpu_tiles - g["tiles"]) / np.sqrt(2 * var)
        width = 1.4826 * np.median(np.abs(z - np.median(z)))
        assert abs(np.median(z)) < 0.1, np.median(z)
        assert 0.85 < width < 1.25, width
        assert (np.abs(z) > 4).mean() < 0.01, (np.abs(z) > 4).mean()
        assert np.all(np.abs(gpu_tiles.mean(axis=(0, 1)) / g["tiles"].mean(axis=(0, 1)) - 1) < 3e-3)
        lower = slice(8, None)                                    # the lit two thirds of the film: per-tile agreement to a few per cent
        assert np.corrcoef(gpu_tiles[lower, :, 1].ravel(), g["tiles"][lower, :, 1].ravel())[0, 1] > 0.995
    be.close()


def test_edge_cases(traversal):
    scene = make_scene(SCENE_0010, width=32, height=32, max_verts=2)       # smallest film, shortest paths
    be = pkg.Backend(scene, traversal=traversal)
    be.render(0, 0)                                                         # empty range is a no-op
    assert be.fb_read().sum() == 0
    be.render(7, 1)                                                         # single path, ragged start
    assert be.counters()[4] == 1
    be.render(1 << 40, 1000)                                                # 64-bit path indices
    assert be.counters()[4] == 1001
    rec = be.trace_paths(1 << 40, 100)
    assert (rec["length"] <= 2).all() and (rec["index"] >= (1 << 40)).all()
    ora = oracle_records(scene, 1 << 40, 100)
    assert np.array_equal(rec["length"], ora["length"])
    be.close()


def test_max_depth_32_deep_paths(traversal):
    scene = make_scene(SCENE_ROUGH, width=1280, height=720, max_verts=32)
    be = pkg.Backend(scene, traversal=traversal)
    n = 200000
    be.render(0, n)
    cnt = be.counters()
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    fb = be.fb_read()
    assert abs(cnt[6] - ocnt[6]) <= 2e-3 * ocnt[6]                           # total path vertices
    assert np.allclose(fb.sum(axis=(0, 1)), ofb.sum(axis=(0, 1)), rtol=5e-3)
    be.close()


def test_errors_are_reported():
    scene = make_scene(SCENE_0010, width=64, height=64, max_verts=8)
    d = scene.desc_ptr
    m = pkg.mi_lib()
    import ctypes as C
    out = C.c_void_p()
    old = d.contents.max_verts
    d.contents.max_verts = 1
    assert m.mi_scene_create(d, C.byref(out)) < 0 and m.mi_last_error()
    d.contents.max_verts = old
    d.contents.struct_size = 3
    assert m.mi_scene_create(d, C.byref(out)) < 0
    d.contents.struct_size = C.sizeof(pkg.MiSceneDesc)
    nv = d.contents.num_vtx
    d.contents.num_vtx = 10                                  # a descriptor whose primitives point past its vertex array
    assert m.mi_scene_create(d, C.byref(out)) < 0 and b"outside the arrays" in m.mi_last_error()
    d.contents.num_vtx = nv
    d.contents.pointsampler = 7
    assert m.mi_scene_create(d, C.byref(out)) < 0
    d.contents.pointsampler = 0
    assert m.mi_render(None, 0, 1) < 0
    # a tree that is not one: a child link back to the root (cycle) and a node reached twice (shared subtree) are both found in O(N)
    nodes = d.contents.nodes
    inner = [(n, c) for n in range(d.contents.num_nodes) for c in range(4) if not (nodes[n].child[c] >> 63)]
    (n0, c0), (n1, c1) = inner[5], inner[6]
    keep = nodes[n0].child[c0]
    nodes[n0].child[c0] = 0
    assert m.mi_scene_create(d, C.byref(out)) < 0 and b"not a tree" in m.mi_last_error()
    nodes[n0].child[c0] = nodes[n1].child[c1]
    assert m.mi_scene_create(d, C.byref(out)) < 0 and b"not a tree" in m.mi_last_error()
    nodes[n0].child[c0] = keep
    # a shape whose material index lies outside the material list (checked for every shape, also those without primitives)
    shapes = d.contents.shapes
    mat = shapes[d.contents.num_shapes - 1].material
    shapes[d.contents.num_shapes - 1].material = d.contents.num_materials
    assert m.mi_scene_create(d, C.byref(out)) < 0 and b"material" in m.mi_last_error()
    shapes[d.contents.num_shapes - 1].material = -1
    assert m.mi_scene_create(d, C.byref(out)) < 0
    shapes[d.contents.num_shapes - 1].material = mat
    assert m.mi_scene_create(d, C.byref(out)) == 0            # and the untouched descriptor still loads
    m.mi_scene_destroy(out)
    assert m.mi_init(4096) < 0 and b"no such device" in m.mi_last_error()      # a device index is not wrapped around any more
    assert m.mi_init(0) == 0


def _compare_hits(scene, be, pos, direction, ignore=None, max_dist=None, traversal="exact"):
    """mi_intersect vs the oracle's accel_intersect on the same rays: primitive and distance bit-exact, u/v bit-exact on
    triangles and quads (spheres / lines get theirs at shading time); work counters equal in exact mode, in fast mode the same rays
    and at least the oracle's node visits / primitive tests"""
    primid = np.ctypeslib.as_array(scene.desc.primid, shape=(scene.desc.num_prims,))
    ign_id = None if ignore is None else np.where(ignore == 0xffffffff, np.uint64(0xffffffffffffffff), primid[np.minimum(ignore, len(primid) - 1)])
    c0 = be.counters()
    gpu = be.intersect(pos, direction, ignore=ignore, max_dist=max_dist)
    c1 = be.counters()
    ora, cnt = oracle_intersect(scene, pos, direction, ignore_primid=ign_id, max_dist=max_dist)
    assert np.array_equal(gpu["primid"], ora["prim"])
    assert np.array_equal(gpu["dist"].view(np.uint32), ora["dist"].view(np.uint32))
    hit = gpu["primid"] != 0xffffffffffffffff
    triquad = hit & ((gpu["primid"] >> np.uint64(61)) >= 3)
    assert np.array_equal(gpu["u"][triquad].view(np.uint32), ora["u"][triquad].view(np.uint32))
    assert np.array_equal(gpu["v"][triquad].view(np.uint32), ora["v"][triquad].view(np.uint32))
    for k in range(4):                    # rays, node visits, box hits, primitive tests
        if traversal == "exact" or k == 0:
            assert c1[k] - c0[k] == cnt[k], (k, c1[k] - c0[k], cnt[k])
        else:
            assert cnt[k] <= c1[k] - c0[k] <= 2 * cnt[k] + 64, (k, c1[k] - c0[k], cnt[k])
    return gpu


def test_intersect_random_rays_bit_exact(traversal):
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8)
    be = pkg.Backend(scene, traversal=traversal)
    rng = np.random.default_rng(7)
    n = 200000
    lo, hi = np.array(scene.desc.aabb[:3]), np.array(scene.desc.aabb[3:6])
    # origins around the objects in the middle of the backdrop, directions uniform on the sphere
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gpu = _compare_hits(scene, be, pos, d, traversal=traversal)
    assert (gpu["primid"] != 0xffffffffffffffff).mean() > 0.3
    kinds = (gpu["primid"][gpu["primid"] != 0xffffffffffffffff] >> np.uint64(61))
    assert set(np.unique(kinds)) >= {1, 2, 4}                     # spheres, lines and quads were all hit
    # with an ignored primitive and a finite search distance
    ignore = rng.integers(0, scene.desc.num_prims, size=n).astype(np.uint32)
    ignore[::3] = 0xffffffff
    _compare_hits(scene, be, pos, d, ignore=ignore, max_dist=rng.uniform(0.5, 30, size=n).astype(np.float32), traversal=traversal)
    assert lo[0] < hi[0]
    be.close()


def test_intersect_degenerate_rays_follow_sse_nan_semantics(traversal):
    """rays with zero direction components (1/dir = +-inf) whose origin lies exactly in box planes: 0*inf = NaN in the
    slab test, resolved by the reference's SSE min/max operand order (qbvhmp.c:1188-1246). The kernel switches to its
    literal compare/select slab test for such waves; hits, distances and counters must still equal the oracle's."""
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8)
    be = pkg.Backend(scene, traversal=traversal)
    rng = np.random.default_rng(11)
    nodes = scene.desc.nodes
    planes = [[], [], []]
    for i in range(scene.desc.num_nodes):
        for k in range(3):
            for c in range(4):
                for b in (nodes[i].aabb[k][c], nodes[i].aabb[k + 3][c]):
                    if abs(b) < 1e30:
                        planes[k].append(b)
    planes = [np.unique(np.float32(p)) for p in planes]
    n = 60000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3)).astype(np.float32)
    for i in range(n):
        zero = rng.integers(1, 7)                    # bit mask of axes with a zero direction component (not all three)
        for k in range(3):
            if zero & (1 << k):
                d[i, k] = 0.0 if rng.random() < 0.5 else -0.0
                if rng.random() < 0.7:
                    pos[i, k] = rng.choice(planes[k])  # origin exactly in a box plane of that axis
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    assert ((d == 0).sum(axis=1) >= 1).all()
    gpu = _compare_hits(scene, be, pos, d, traversal=traversal)
    assert (gpu["primid"] != 0xffffffffffffffff).mean() > 0.2
    # mixed waves: only every 64th ray degenerate
    mixed = rng.normal(size=(n, 3)).astype(np.float32)
    mixed[::64] = d[::64]
    mixed = (mixed / np.linalg.norm(mixed, axis=1, keepdims=True)).astype(np.float32)
    _compare_hits(scene, be, pos, mixed, traversal=traversal)
    be.close()


def test_intersect_rays_through_vertices_and_edges(traversal):
    """rays aimed at mesh vertices and edge midpoints: several primitives -- often in different leaves -- are hit at the SAME
    distance, and which one is reported depends on the order the reference tests them in and on which leaves it reaches at all
    (a leaf whose box the ray enters an ulp behind the tie is never tested). The FAST rounds reach more leaves than the reference;
    their entry-distance bookkeeping (trace_round_spec, MI_SPEC_EXACT) has to drop exactly the hits the reference cannot have."""
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8)
    be = pkg.Backend(scene, traversal=traversal)
    rng = np.random.default_rng(23)
    vtx = np.ctypeslib.as_array(C.cast(scene.desc.vtx, C.POINTER(C.c_float)), shape=(scene.desc.num_vtx, 4))[:, :3].copy()
    n = 400000
    a = vtx[rng.integers(0, len(vtx), size=n)]
    # half of the targets: a vertex; the other half: the midpoint to the next vertex of the array (a mesh edge for grid meshes)
    k = rng.integers(0, len(vtx) - 1, size=n)
    mid = np.float32(0.5) * (vtx[k] + vtx[k + 1])
    target = np.where((np.arange(n) % 2 == 0)[:, None], a, mid).astype(np.float32)
    pos = (rng.uniform(-4, 4, size=(n, 3)) + [0, 0, 3]).astype(np.float32)
    d = target - pos
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gpu = _compare_hits(scene, be, pos, d, traversal=traversal)
    assert (gpu["primid"] != 0xffffffffffffffff).mean() > 0.9
    be.close()


def test_fast_rounds_give_the_exact_rounds_paths():
    """the FAST rounds against the exact rounds of the same library, path record by path record (bytes): pt, ptdl and the extended
    kernels. Before the entry-distance bookkeeping 0.25 paths per million differed (ties on shared edges, tests/dev/fast_vs_exact.py)."""
    for path, sampler, n in ((SCENE_0010, pkg.MI_SAMPLER_PT, 2000000), (SCENE_0010, pkg.MI_SAMPLER_PTDL, 1000000), (SCENE_MEDIA, pkg.MI_SAMPLER_PT, 500000)):
        scene = make_scene(path, width=1280, height=720, max_verts=8, sampler=sampler)
        ex, fa = pkg.Backend(scene, traversal="exact"), pkg.Backend(scene, traversal="fast")
        assert fa.traversal() == "fast"
        for first in range(31, 31 + n, 250000):
            a, b = ex.trace_paths(first, 250000), fa.trace_paths(first, 250000)
            assert a.tobytes() == b.tobytes(), (path, sampler, first)
        ex.close(); fa.close()


def _three_class_scene(tmp_path):
    """0010 with the cylinder in gold: diffuse, dielectric and metal vertices in one scene (three classes of the exchange between waves)"""
    import shutil
    dst = tmp_path / "0010_three"
    shutil.copytree(SCENE_0010.parent, dst)
    lines = (dst / "test.nra2").read_text().splitlines()
    n = int(lines[1])
    assert lines[2 + n - 1].startswith("mult 2 11 3 0") and lines[2 + n + 1 + 4].endswith("../geo/cylinder")
    lines[1] = str(n + 2)
    lines[2 + n:2 + n] = ["metal Au # %d" % n, "mult 1 7 %d # %d gold" % (n, n + 1)]
    k = 2 + (n + 2) + 1 + 4
    lines[k] = "%d ../geo/cylinder" % (n + 1)
    text = "\n".join(lines) + "\n"
    (dst / "test.nra2").write_text(text.replace("../geo/", str(SCENE_0010.parent.parent / "geo") + "/"))       # the geometry stays where it is
    return dst / "test.nra2"


def test_exchange_between_waves_changes_no_path(monkeypatch, tmp_path):
    """The material queues (csrc/mi_regroup.h): the waves of a workgroup trade path vertices by class of the material -- which lane
    finishes a path must not matter to the path. Path records with the exchange (the default) and without it (CORONA_MI_REGROUP=0) are
    the same BYTES: plain pt / ptdl, a scene with three classes (diffuse, dielectric, metal: the cylinder of 0010 in gold) against the oracle
    too, the extended kernels (volume vertices as a class of their own), the Halton sampler (the path index travels with the vertex), a tree
    in HBM (bigger pools), launches smaller than a wave and smaller than a workgroup (the pools must run empty before the last wave leaves);
    rendered frames agree to the order of the float atomics and count every path."""
    three = _three_class_scene(tmp_path)
    cases = [(SCENE_0010, pkg.MI_SAMPLER_PT, "rand", 8, 400000), (SCENE_0010, pkg.MI_SAMPLER_PTDL, "rand", 8, 300000),
             (three, pkg.MI_SAMPLER_PT, "rand", 8, 300000), (three, pkg.MI_SAMPLER_PTDL, "rand", 8, 200000),
             (SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, "rand", 32, 200000), (SCENE_CAM_MB, pkg.MI_SAMPLER_PT, "rand", 8, 200000),
             (SCENE_0010, pkg.MI_SAMPLER_PTDL, "halton", 8, 200000), (SCENE_FINE, pkg.MI_SAMPLER_PT, "rand", 8, 200000),
             (SCENE_ROUGH, pkg.MI_SAMPLER_PT, "rand", 32, 200000)]
    for path, sampler, points, mv, n in cases:
        scene = make_scene(path, width=1280, height=720, max_verts=mv, sampler=sampler,
                           pointsampler=pkg.MI_POINTS_HALTON if points == "halton" else pkg.MI_POINTS_RAND)
        monkeypatch.setenv("CORONA_MI_REGROUP", "0")
        off = pkg.Backend(scene)
        monkeypatch.delenv("CORONA_MI_REGROUP")
        on = pkg.Backend(scene)
        for first, count in ((5, n), (123456789, 1), (77, 63), (1000, 1000), (2 ** 33 + 9, 70000)):
            a, b = off.trace_paths(first, count), on.trace_paths(first, count)
            assert a.tobytes() == b.tobytes(), (str(path), sampler, points, first, count)
        if path == three:
            ora = oracle_records(scene, 5, 50000)
            g = on.trace_paths(5, 50000)
            same = (g["length"] == ora["length"]) & (g["num_splats"] == ora["num_splats"])
            for k in range(1, 8):
                sel = ora["length"] > k
                same &= ~sel | (g["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
            assert (~same).sum() <= 2, int((~same).sum())
            assert len(np.unique(g["v"]["shader"][g["length"] > 2, 1])) >= 3          # the three materials are all hit
        off.close(); on.close()
        # production kernels: every path counted, the same frame up to the order of the float atomics
        per = 4 * scene.width * scene.height
        frames = []
        for env in ("0", None):
            if env is None:
                monkeypatch.delenv("CORONA_MI_REGROUP", raising=False)
            else:
                monkeypatch.setenv("CORONA_MI_REGROUP", env)
            be = pkg.Backend(scene, counters=False)
            c0 = be.counters()
            be.render(9 * per, per)
            frames.append(be.fb_read())
            assert be.counters()[4] - c0[4] == per
            be.close()
        monkeypatch.delenv("CORONA_MI_REGROUP", raising=False)
        assert np.abs(frames[0] - frames[1]).max() <= 2e-4 * np.abs(frames[0]).max(), (str(path), sampler)


def test_cfg5_film_3840x2160(counters):
    """BASELINE config 5's film (3840x2160, padded to 3840x2176; 100 MB framebuffer): one sample per pixel, sharded over two
    path-index ranges like two ranks would, against the oracle's image of the same indices"""
    scene = make_scene(SCENE_0010, width=3840, height=2160, max_verts=8)
    assert (scene.width, scene.height) == (3840, 2176)
    be = pkg.Backend(scene, counters=counters)          # the library's default traversal (fast)
    n = scene.width * scene.height
    for r in range(2):
        first, count = pkg.shard_range(0, n, r, 2)
        be.render(first, count)
    fb = be.fb_read()
    cnt = be.counters()
    assert cnt[4] == n
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    gain = scene.gain(1)
    rmse = np.sqrt((((fb - ofb) * gain) ** 2).sum() / n)
    assert rmse < 0.05, rmse
    assert np.allclose(fb.sum(axis=(0, 1)), ofb.sum(axis=(0, 1)), rtol=1e-3)
    if counters:
        check_work(cnt, ocnt, "fast")
    be.close()


def test_tree_larger_than_lds_is_read_from_hbm(traversal, counters):
    """the 1711-node tree of scenes/0054_fine (198 KB) does not fit next to the 96 KB of traversal stacks: the kernels
    instantiated with the tree in HBM take over; ray-level results stay bit-exact and the image matches the oracle"""
    scene = make_scene(SCENE_FINE, width=640, height=352, max_verts=8)
    assert scene.desc.num_nodes * 112 > 64 * 1024
    be = pkg.Backend(scene, traversal=traversal, counters=counters)
    assert not be.nodes_in_lds()
    rng = np.random.default_rng(3)
    n = 100000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[::50, 1] = 0.0                                       # some degenerate rays too
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    _compare_hits(scene, be, pos, d, traversal=traversal)
    npx = scene.width * scene.height
    be.render(0, npx)
    fb = be.fb_read()
    ofb, _, _ = oracle_render(scene, 0, npx, threads=8)
    rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / npx)
    assert rmse < 0.05, rmse
    be.close()


@pytest.mark.parametrize("top", [None, "0", "37"], ids=["top that fits", "nothing staged", "37 nodes staged"])
def test_large_tree_top_in_lds_rest_from_hbm(monkeypatch, traversal, counters, top):
    """scenes/0064_large: 262 156 primitives (the backdrop split 8 x 8, tools/make_geo.py, hash-checked in conftest.py), a QBVH of 27 104
    nodes = 3.5 MB of 128-byte records (qbvh_node_t of src/accel.d/qbvhmp.c:62-81 in 128 instead of 256 bytes). Only the top of the tree
    -- as many breadth-first numbered nodes as the LDS takes next to stacks and pools -- is staged; everything below is read from
    HBM / L2, one record per visit. Ray-level results bit-exact against the oracle (same counters in exact mode), the image against the
    oracle's; the same with no node staged and with a top that ends in the middle of a level (CORONA_MI_NODES_TOP)."""
    if top is not None:
        monkeypatch.setenv("CORONA_MI_NODES_TOP", top)
    scene = make_scene(SCENE_LARGE, width=640, height=352, max_verts=8)
    assert scene.desc.num_prims == 262156 and scene.desc.num_nodes * 128 > 3 << 20
    be = pkg.Backend(scene, traversal=traversal, counters=counters)
    staged = be.lds_nodes()
    assert not be.nodes_in_lds()
    if top is None:
        assert 128 <= staged < scene.desc.num_nodes, staged           # at least the first four levels of the 4-wide tree
    else:
        assert staged == int(top)
    assert "mi_path_kernel<false, false, false," in be.kernel_name()
    rng = np.random.default_rng(11)
    n = 60000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3))
    d[::50, 1] = 0.0                                       # some rays with an infinite 1/dir: the literal slab test
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    _compare_hits(scene, be, pos, d, traversal=traversal)
    if top is None or (traversal == "exact" and counters):
        npx = scene.width * scene.height
        be.render(0, npx)
        fb = be.fb_read()
        ofb, _, _ = oracle_render(scene, 0, npx, threads=8)
        rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / npx)
        assert rmse < 0.05, rmse
        gpu = be.trace_paths(0, 4000)
        ora = oracle_records(scene, 0, 4000)
        assert (gpu["length"] == ora["length"]).mean() >= 0.999
    be.close()


def test_per_pixel_against_reference_render():
    """per-pixel L2 against the REAL reference: 64x64 film, 65 536 spp (tests/golden/image_pt_mv8_64.npz holds two
    independent reference renders a, b of that job). Pure pt finds the small emitter with 0.5 % of its paths, so even at
    this sample count a pixel carries a few per cent of noise; the tolerance is the reference's own noise floor:
    E|g - (a+b)/2|^2 = 0.75 E|a - b|^2 for an unbiased g, asserted with a 15 % margin, plus the image means."""
    g = np.load(GOLDEN / "image_pt_mv8_64.npz")
    a, b, spp = g["a"], g["b"], int(g["spp"])
    scene = make_scene(SCENE_0010, width=64, height=64, max_verts=int(g["max_verts"]))
    assert a.shape == (scene.height, scene.width, 3)
    be = pkg.Backend(scene)
    be.render(0, spp * scene.width * scene.height)
    img = be.fb_read() * scene.gain(spp)
    be.close()
    ref = 0.5 * (a + b)
    floor = np.sqrt(((a - b) ** 2).mean())
    rmse = np.sqrt(((img - ref) ** 2).mean())
    assert rmse <= floor, (rmse, floor)                       # expected 0.87 * floor for Gaussian noise; measured 0.68 (heavy-tailed)
    assert rmse >= 0.5 * floor                                # and not suspiciously smooth either
    assert np.all(np.abs(img.mean(axis=(0, 1)) / ref.mean(axis=(0, 1)) - 1) < 5e-3)


def test_command_line_renderer(tmp_path):
    """the stand-alone C host (corona-13_amd/host/corona-mi: scene loaders, progression loop with --batch, PFM writer,
    sidecar) drives the same ABI: its image equals the library render of the same path indices (float atomic order
    aside), compared with host/pfmdiff-mi like the reference's regression scripts compare renders"""
    import re
    import shutil
    import subprocess
    from helpers import REPO
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    cli = REPO / "corona-13_amd" / "host" / "corona-mi"
    scene_file = tmp_path / "scenes" / "0010_pt" / "test.nra2"
    out = subprocess.run([str(cli), str(scene_file), "-s", "24", "--batch", "10", "-w", "256", "-h", "256", "--max-verts", "8", "-x", "_cli"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    pfm = tmp_path / "scenes" / "0010_pt" / "test_cli_fb00.pfm"
    side = (tmp_path / "scenes" / "0010_pt" / "test_cli_fb00.pfm.txt").read_text()
    assert re.search(r"samples per pixel: 24 ", side) and re.search(r"elapsed wallclock prog [\d.]+s", side) and "res 256x256" in side
    # the same job through the Python view of the ABI, with the host's own colour fit (no injected reference coefficients)
    scene = make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=8)
    be = pkg.Backend(scene)
    be.render(0, 24 * scene.width * scene.height)
    img = be.fb_read() * scene.gain(24)
    be.close()
    with open(tmp_path / "lib.pfm", "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.float32).tobytes())     # rows in framebuffer order, like view_write_images
    d = subprocess.run([str(REPO / "corona-13_amd" / "host" / "pfmdiff-mi"), str(pfm), str(tmp_path / "lib.pfm")], capture_output=True, text=True)
    assert d.returncode == 0, d.stdout + d.stderr
    assert float(d.stdout.split("rmse:")[1]) < 1e-3, d.stdout
    # --device-build: same image from the tree built on the GPU
    out = subprocess.run([str(cli), str(scene_file), "-s", "24", "--batch", "24", "-w", "256", "-h", "256", "--max-verts", "8", "-x", "_dev", "--device-build"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    d = subprocess.run([str(REPO / "corona-13_amd" / "host" / "pfmdiff-mi"), str(tmp_path / "scenes" / "0010_pt" / "test_dev_fb00.pfm"), str(tmp_path / "lib.pfm")],
                       capture_output=True, text=True)
    assert d.returncode == 0 and float(d.stdout.split("rmse:")[1]) < 1e-3, d.stdout + d.stderr


def test_other_frame_seed_and_large_indices():
    """the per-path generator is keyed by (path index, frame) (src/points.d/xorshift128p.c:53-59): another frame number and
    path indices beyond 2^32 (a long progressive render) still give the oracle's paths"""
    scene = make_scene(SCENE_0010, width=640, height=352, max_verts=8, frame=7)
    be = pkg.Backend(scene)
    for first in (0, (1 << 33) + 12345):
        gpu = be.trace_paths(first, 4000)
        ora = oracle_records(scene, first, 4000)
        assert np.array_equal(gpu["index"], ora["index"])
        assert np.abs(gpu["pixel_i"] - ora["pixel_i"]).max() <= 1e-5 and np.abs(gpu["lambda"] - ora["lambda"]).max() <= 1e-5
        assert (gpu["length"] == ora["length"]).mean() >= 0.999
        m = (gpu["length"] == ora["length"]) & (ora["length"] > 2)
        assert (gpu["v"]["prim"][m, 2] == ora["v"]["prim"][m, 2]).mean() >= 0.999
    # and the frame does change the samples
    other = pkg.Backend(make_scene(SCENE_0010, width=640, height=352, max_verts=8, frame=8))
    assert np.abs(other.trace_paths(0, 64)["pixel_i"] - be.trace_paths(0, 64)["pixel_i"]).max() > 1.0
    other.close()
    be.close()


@pytest.mark.parametrize("scene_path", [SCENE_0010, SCENE_FINE])
def test_device_built_tree_gives_the_same_hits(scene_path, traversal):
    """with no tree handed over (mi_scene_desc.nodes = NULL) the backend builds its own 4-wide BVH on the
    GPU (LBVH + collapse, csrc/mi_build.h). Closest hits do not depend on the tree: same primitive and bit-identical distance
    as the oracle (which walks the host-built reference tree) for every ray, exact ties aside."""
    scene = make_scene(scene_path, width=640, height=352, max_verts=8)
    be = pkg.Backend(scene, device_build=True, traversal=traversal)
    rng = np.random.default_rng(21)
    n = 200000
    pos = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.float32([0, 0, 2])
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[::40, 2] = 0.0                                       # degenerate rays too
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    gpu = be.intersect(pos, d)
    ora, _ = oracle_intersect(scene, pos, d)
    same = gpu["primid"] == ora["prim"]
    assert same.mean() >= 0.9999, same.mean()
    assert np.array_equal(gpu["dist"][same].view(np.uint32), ora["dist"][same].view(np.uint32))
    assert (gpu["primid"] != 0xffffffffffffffff).mean() > 0.3
    be.close()


@pytest.mark.parametrize("sampler", [pkg.MI_SAMPLER_PT, pkg.MI_SAMPLER_PTDL])
def test_device_built_tree_renders_the_same_paths(sampler, traversal):
    """paths and image with the device-built tree against the oracle (emitter indices are re-mapped to the new primitive
    order, which ptdl's next event estimation depends on)"""
    scene = make_scene(SCENE_0010, width=640, height=352, max_verts=8, sampler=sampler)
    be = pkg.Backend(scene, device_build=True, traversal=traversal)
    gpu = be.trace_paths(500, 20000)
    ora = oracle_records(scene, 500, 20000)
    same = gpu["length"] == ora["length"]
    assert same.mean() >= 0.999
    assert (gpu["num_splats"] == ora["num_splats"]).mean() >= 0.999
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if m.sum():
            assert (gpu["v"]["prim"][m, k] == ora["v"]["prim"][m, k]).mean() >= 0.999
    npx = scene.width * scene.height
    c0 = be.counters()
    be.render(0, npx)
    fb = be.fb_read()
    ofb, ocnt, _ = oracle_render(scene, 0, npx, threads=8)
    rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / npx)
    assert rmse < 0.05, rmse
    cnt = [b - a for a, b in zip(c0, be.counters())]
    assert abs(cnt[0] - ocnt[0]) <= 1e-4 * ocnt[0] and cnt[4] == npx     # same rays; node / primitive counts belong to the other tree
    be.close()


def test_device_build_tiny_scene(tmp_path):
    """three quads (the emitter alone): the smallest trees the device build produces (two internal binary nodes)"""
    import shutil
    from helpers import REPO
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    nra = tmp_path / "scenes" / "0010_pt" / "test.nra2"
    lines = nra.read_text().splitlines()
    k = lines.index("6")                                   # the shape count line
    nra.write_text("\n".join(lines[:k] + ["1", "5 ../geo/emitter"]) + "\n")
    scene = make_scene(nra, inject=False, width=64, height=64, max_verts=4)
    assert scene.desc.num_prims == 3
    host, devb = pkg.Backend(scene), pkg.Backend(scene, device_build=True)
    rng = np.random.default_rng(2)
    lo, hi = np.array(scene.desc.aabb[:3]), np.array(scene.desc.aabb[3:6])
    n = 20000
    target = rng.uniform(lo, hi, size=(n, 3))
    pos = (target + rng.normal(size=(n, 3)) * 3).astype(np.float32)
    d = target - pos
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a, b = host.intersect(pos, d), devb.intersect(pos, d)
    assert (a["primid"] != 0xffffffffffffffff).mean() > 0.05
    assert np.array_equal(a["primid"], b["primid"]) and np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32))
    host.close(); devb.close()


@pytest.mark.parametrize("kind,count", [("uniform", 5), ("uniform", 61), ("uniform", 700), ("clusters", 3000), ("stacked", 400), ("sliver", 900)])
def test_device_build_random_soups(tmp_path, monkeypatch, kind, count):
    """the device build (Morton sort, rotations, collapse, final permutation) on generated triangle / quad soups -- few primitives, many,
    tight clusters (equal Morton codes), copies stacked on one spot (equal codes AND equal boxes), long slivers: the same closest hit
    as the host-built (reference-style) tree for every ray, exact ties aside, with and without the SAH refinement"""
    import shutil, sys
    from helpers import REPO
    sys.path.insert(0, str(REPO / "tools"))
    import make_geo
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    rng = np.random.default_rng(count)
    if kind == "uniform":
        c = rng.uniform(-3, 3, size=(count, 3))
        ext = rng.uniform(0.05, 0.6, size=(count, 1))
    elif kind == "clusters":
        c = rng.uniform(-3, 3, size=(12, 3))[rng.integers(0, 12, count)] + rng.normal(size=(count, 3)) * 1e-3
        ext = rng.uniform(0.01, 0.05, size=(count, 1))
    elif kind == "stacked":
        c = np.repeat(rng.uniform(-2, 2, size=(count // 20, 3)), 20, axis=0)
        ext = np.repeat(rng.uniform(0.1, 0.4, size=(count // 20, 1)), 20, axis=0)
    else:
        c = rng.uniform(-3, 3, size=(count, 3))
        ext = rng.uniform(0.01, 0.03, size=(count, 1))
    count = len(c)
    quad = rng.random(count) < 0.5
    e1 = rng.normal(size=(count, 3)); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(e1, rng.normal(size=(count, 3))); e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    long = 40.0 if kind == "sliver" else 1.0
    if kind == "stacked":
        e1[:] = np.repeat(e1[::20], 20, axis=0); e2[:] = np.repeat(e2[::20], 20, axis=0)
    corners = np.stack([c, c + e1 * ext * long, c + e1 * ext * long + e2 * ext, c + e2 * ext], axis=1).astype(np.float32)
    _, _, plane_vtx = make_geo.read_geo(REPO / "scenes" / "geo" / "plane.geo")
    primid, vtxidx, vtx = [], [], []
    for i in range(count):
        nv = 4 if quad[i] else 3
        primid.append((nv << 61) | (len(vtxidx) << 32))
        for k in range(nv):
            vtxidx.append((len(vtx), 0))
            vtx.append((tuple(corners[i, k]), int(plane_vtx["n"][0])))
    make_geo.write_geo(tmp_path / "scenes" / "geo" / "soup.geo", np.array(primid, dtype=np.uint64), np.array(vtxidx, dtype=make_geo.VTXIDX), np.array(vtx, dtype=make_geo.VTX))
    nra = tmp_path / "scenes" / "0010_pt" / "test.nra2"
    lines = nra.read_text().splitlines()
    k = lines.index("6")
    nra.write_text("\n".join(lines[:k] + ["2", "2 ../geo/soup", "5 ../geo/emitter"]) + "\n")
    scene = make_scene(nra, inject=False, width=64, height=64, max_verts=4)
    assert scene.desc.num_prims == count + 3
    n = 60000
    lo, hi = np.array(scene.desc.aabb[:3]), np.array(scene.desc.aabb[3:6])
    target = corners[rng.integers(0, count, n)].mean(axis=1) + rng.normal(size=(n, 3)) * 0.05
    pos = (target + rng.normal(size=(n, 3)) * 4).astype(np.float32)
    d = target - pos
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    host = pkg.Backend(scene)
    a = host.intersect(pos, d)
    host.close()
    assert (a["primid"] != 0xffffffffffffffff).mean() > (0.1 if kind == "sliver" else 0.3)
    for sah in ("0", "2", "5"):
        monkeypatch.setenv("CORONA_MI_BUILD_SAH", sah)
        for leaf in ("2", "4", "7"):
            monkeypatch.setenv("CORONA_MI_BUILD_LEAF", leaf)
            devb = pkg.Backend(scene, device_build=True)
            b = devb.intersect(pos, d)
            devb.close()
            # same distance for every ray, bit for bit; the primitive may differ where several lie at that distance (the stacked copies)
            assert np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32)), (kind, sah, leaf, int((a["dist"] != b["dist"]).sum()))
            if kind != "stacked":
                assert (a["primid"] == b["primid"]).mean() >= 0.9999, (kind, sah, leaf)


def test_scene_stats():
    """mi_scene_stats: node count, where the tree lives, stack need, who built it"""
    scene = make_scene(width=64, height=64, max_verts=4)
    be = pkg.Backend(scene)
    st = be.stats()
    import os
    forced_hbm = os.environ.get("CORONA_MI_NODES") == "global"        # developer switch: keep every tree in HBM
    assert st["nodes"] == scene.desc.num_nodes and st["nodes_in_lds"] == (not forced_hbm) and not st["device_built"] and 3 <= st["stack_need"] <= 64
    be.close()
    fine = make_scene(SCENE_FINE, width=64, height=64, max_verts=4)
    be = pkg.Backend(fine)
    st = be.stats()
    assert st["nodes"] == fine.desc.num_nodes and not st["nodes_in_lds"]
    be.close()
    be = pkg.Backend(scene, device_build=True)
    st = be.stats()
    assert st["device_built"] and scene.desc.num_prims // 8 <= st["nodes"] < scene.desc.num_prims
    be.close()


def test_halton_long_paths_against_reference_golden():
    """the reference's own 29..32-vertex ptdl paths (dimension >= 256 falls back to the per-path generator, halton.c:78-80)"""
    g = np.load(GOLDEN / "paths_halton_long_mv32.npz")
    ref = g["records"]
    scene = make_scene(SCENE_ROUGH, width=1280, height=720, max_verts=32, sampler=pkg.MI_SAMPLER_PTDL, pointsampler=pkg.MI_POINTS_HALTON)
    be = pkg.Backend(scene)
    gpu = np.concatenate([be.trace_paths(int(i), 1) for i in ref["index"]])
    ora = np.concatenate([oracle_records(scene, int(i), 1) for i in ref["index"]])
    assert (gpu["length"] == ora["length"]).mean() >= 0.98 and (gpu["length"] == ref["length"]).mean() >= 0.97
    same = gpu["length"] == ref["length"]
    for k in range(8):
        assert (gpu["v"]["prim"][same, k] == ref["v"]["prim"][same, k]).mean() >= 0.99
    m = same & (gpu["num_splats"] == ref["num_splats"]) & (ref["num_splats"] > 0)
    assert m.sum() > 100 and (gpu["splat"]["length"][m] == ref["splat"]["length"][m]).all()
    be.close()


def test_halton_golden_and_reseeding():
    """the first reference paths directly; and a range whose end passes 2^32 indices is rendered with the permutations of
    seed frame + 1 (pointsampler_prepare_frame, halton.c:122-129) while the index itself is cut to 32 bits"""
    g = np.load(GOLDEN / "paths_halton_pt_mv8.npz")
    ref = g["records"]
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, pointsampler=pkg.MI_POINTS_HALTON)
    be = pkg.Backend(scene)
    gpu = be.trace_paths(0, len(ref))
    assert np.abs(gpu["pixel_i"] - ref["pixel_i"]).max() <= 1e-4 and np.abs(gpu["pixel_j"] - ref["pixel_j"]).max() <= 1e-4
    assert np.abs(gpu["lambda"] - ref["lambda"]).max() <= 1e-4
    assert (gpu["length"] == ref["length"]).mean() >= 0.998
    first = (1 << 32) - 700
    hi = be.trace_paths(first, 1500)                       # end >> 32 == 1
    ora = oracle_records(scene, first, 1500)
    assert np.array_equal(hi["pixel_i"], ora["pixel_i"]) and np.array_equal(hi["lambda"], ora["lambda"])
    assert (hi["length"] == ora["length"]).mean() >= 0.998
    lo = be.trace_paths(first, 600)                        # end >> 32 == 0: same indices, other permutations
    assert np.array_equal(lo["pixel_i"], hi["pixel_i"][:600])          # base 2 and 3 are not permuted
    assert not np.array_equal(lo["lambda"], hi["lambda"][:600])        # base 5 is
    again = be.trace_paths(0, len(ref))                    # back to the first tables
    assert np.array_equal(again["lambda"], gpu["lambda"]) and np.array_equal(again["length"], gpu["length"])
    be.close()


def test_halton_image_matches_oracle_and_differs_from_rand(traversal, counters):
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, pointsampler=pkg.MI_POINTS_HALTON)
    npx = scene.width * scene.height
    be = pkg.Backend(scene, traversal=traversal, counters=counters)
    be.render(0, 4 * npx)
    fb = be.fb_read()
    ofb, ocnt, _ = oracle_render(scene, 0, 4 * npx, threads=8)
    cnt = be.counters()
    be.close()
    rmse = np.sqrt((((fb - ofb) * scene.gain(4)) ** 2).sum() / npx)
    assert rmse < 0.05, rmse
    assert cnt[4] == 4 * npx and (not counters or abs(cnt[0] - ocnt[0]) <= 1e-4 * ocnt[0])
    rnd = make_scene(SCENE_0010, width=256, height=256, max_verts=8)
    be = pkg.Backend(rnd)
    be.render(0, 4 * npx)
    fr = be.fb_read()
    be.close()
    assert abs(fr.sum() - fb.sum()) < 0.25 * fb.sum() and not np.allclose(fr, fb)      # same image in expectation (pt at 4 spp: noisy), other samples


@pytest.mark.parametrize("scene_path,sampler", [(SCENE_MEDIA, pkg.MI_SAMPLER_PT), (SCENE_FOG, pkg.MI_SAMPLER_PTDL)])
def test_media_image_matches_oracle(scene_path, sampler, traversal, counters):
    """1-spp film through the MEDIA kernels (splats of paths with volume vertices included) against the oracle's"""
    scene = make_scene(scene_path, width=512, height=288, max_verts=8, sampler=sampler)
    npx = scene.width * scene.height
    be = pkg.Backend(scene, traversal=traversal, counters=counters)
    be.render(0, npx)
    fb = be.fb_read()
    cnt = be.counters()
    be.close()
    ofb, ocnt, _ = oracle_render(scene, 0, npx, threads=8)
    assert np.isfinite(fb).all()
    rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / npx)
    assert rmse < 0.05, rmse
    assert cnt[4] == npx
    if counters:
        assert abs(cnt[0] - ocnt[0]) <= 1e-4 * ocnt[0] and abs(cnt[6] - ocnt[6]) <= 1e-4 * ocnt[6]   # rays, vertices
        assert cnt[5] == ocnt[5] or abs(cnt[5] - ocnt[5]) <= 1e-3 * ocnt[5]                         # splats


def test_motion_blur_image_and_restrictions(counters):
    """1-spp film of the moving-geometry scene against the oracle; what the backend cannot do with moving primitives is reported"""
    scene = make_scene(SCENE_MB, width=512, height=288, max_verts=8)
    npx = scene.width * scene.height
    be = pkg.Backend(scene, counters=counters)
    be.render(0, npx)
    fb = be.fb_read()
    cnt = be.counters()
    ofb, ocnt, _ = oracle_render(scene, 0, npx, threads=8)
    rmse = np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / npx)
    assert rmse < 0.05, rmse
    assert cnt[4] == npx and (not counters or (abs(cnt[0] - ocnt[0]) <= 1e-4 * ocnt[0] and abs(cnt[6] - ocnt[6]) <= 1e-4 * ocnt[6]))
    with pytest.raises(RuntimeError, match="no time"):
        be.intersect(np.zeros((4, 3), np.float32), np.tile(np.float32([0, 0, 1]), (4, 1)))
    host_paths = be.trace_paths(777, 8000)
    be.close()
    # the device-built tree (boxes enclosing both states of a moving primitive) finds the same paths
    bd = pkg.Backend(scene, device_build=True)
    dev_paths = bd.trace_paths(777, 8000)
    bd.close()
    assert np.array_equal(host_paths["length"], dev_paths["length"]) and np.array_equal(host_paths["v"]["prim"], dev_paths["v"]["prim"])
    assert np.array_equal(host_paths["v"]["dist"].view(np.uint32), dev_paths["v"]["dist"].view(np.uint32))
    # the same geometry frozen at shutter open gives another image: the interpolation is really applied
    still = make_scene(SCENE_0010, width=512, height=288, max_verts=8)
    bs = pkg.Backend(still)
    bs.render(0, npx)
    assert not np.allclose(bs.fb_read(), fb)
    bs.close()


def test_frame_reducer_on_device_matches_plain_render():
    """bench.py's multi-GPU step on one GPU: RCCL process group of one rank, renders into the two buffers of FrameReducer with the
    asynchronous all-reduce in flight; every reduced frame equals a plain render of the same path indices"""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8)
        per = 4 * scene.width * scene.height
        be = pkg.Backend(scene, device=0)
        be.set_stream(torch.cuda.current_stream().cuda_stream)
        shape = (scene.height, scene.width, 3)
        red = pkg.FrameReducer([torch.zeros(shape, device="cuda:0"), torch.zeros(shape, device="cuda:0")], dist)
        got = []
        for k in range(4):
            be.set_framebuffer(red.begin(k).data_ptr())
            be.render(k * per, per)
            red.end(k)
            if k:
                got.append(red.finished(k - 1).clone())
        got.append(red.finished(3).clone())
        red.drain()
        torch.cuda.synchronize()
        plain = torch.zeros(shape, device="cuda:0")
        be.set_framebuffer(plain.data_ptr())
        for k in range(4):
            plain.zero_()
            be.render(k * per, per)
            be.sync()
            err = float((got[k] - plain).abs().max())
            assert err <= 1e-4 * float(plain.abs().max()) and float(plain.sum()) > 0, (k, err, float(plain.abs().max()), float(got[k].sum()), float(plain.sum()))
        be.close()
    finally:
        dist.destroy_process_group()


def _chunk_tree(desc, leaf_size):
    """a valid 4-wide tree over the descriptor's primitive list whose leaves hold `leaf_size` primitives each (the last one the rest):
    every child box is the scene's box, so a ray that meets the scene visits every leaf in link order"""
    P = int(desc.num_prims)
    links = [pkg.MI_NODE_LEAF | (first << 5) | min(leaf_size, P - first) for first in range(0, P, leaf_size)]
    nodes = []                                   # (children links, parent); node 0 is appended last and moved to the front below

    def group(level):                            # -> links of the level above
        out = []
        for i in range(0, len(level), 4):
            nodes.append(level[i:i + 4])
            out.append(len(nodes) - 1)
        return out
    level = links
    while True:
        level = group(level)
        if len(level) == 1:
            break
    n = len(nodes)
    remap = lambda k: n - 1 - k                  # the root was made last: reverse the numbering so that it becomes node 0
    arr = (pkg.MiNode * n)()
    for k, ch in enumerate(nodes):
        nd = arr[remap(k)]
        nd.axis0, nd.axis00, nd.axis01, nd.parent = 0, 1, 2, -1
        for c in range(4):
            if c < len(ch):
                link = ch[c]
                nd.child[c] = link if link & pkg.MI_NODE_LEAF else remap(link)
                if not link & pkg.MI_NODE_LEAF:
                    arr[remap(link)].parent = remap(k)
                for a in range(3):
                    nd.aabb[a][c], nd.aabb[a + 3][c] = desc.aabb[a], desc.aabb[a + 3]
            else:
                nd.child[c] = pkg.MI_NODE_LEAF       # empty child: no primitives, inverted box (qbvhmp.c:1095-1112)
                for a in range(3):
                    nd.aabb[a][c], nd.aabb[a + 3][c] = 3.4028234663852886e38, -3.4028234663852886e38
    arr[0].parent = -1
    return arr


def test_leaf_phase_corner_cases(tmp_path, traversal):
    """the distributed leaf phase (leaf_jobs, mi_kernels.h) hands three cases back to the per-lane loop: a folded quad crossed in
    both halves (what it reports depends on the running closest hit, src/prims.c:654-663), leaves of more than 7 primitives, and
    rounds with more jobs than the wave's list holds. All three against the oracle's accel_intersect, bit for bit, counters included:
    a stack of folded quads seen edge-on, under the builder's tree and under trees with leaves of 3 / 7 / 20 / 31 primitives."""
    import shutil, sys
    from helpers import REPO
    sys.path.insert(0, str(REPO / "tools"))
    from make_geo import read_geo, write_geo, VTXIDX, VTX
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    # 40 quads v0 v1 v2 v3 folded along the diagonal v0-v2 into a V (v1 and v3 lifted): a ray across the diagonal meets both halves
    Q = 40
    _, tvi, tv = read_geo(REPO / "scenes" / "geo" / "emitter.geo")
    primid = (np.uint64(4) << np.uint64(61)) | (np.arange(Q, dtype=np.uint64) * np.uint64(4) << np.uint64(32))
    vtxidx = np.zeros(4 * Q, dtype=VTXIDX)
    vtxidx["v"] = np.arange(4 * Q)
    vtxidx["uv"] = np.tile(tvi["uv"][:4], Q)
    vtx = np.zeros(4 * Q, dtype=VTX)
    vtx["n"] = tv["n"][0]
    for q in range(Q):
        z = np.float32(0.5 + 0.05 * q)
        vtx["p"][4 * q:4 * q + 4] = np.float32([[-1, -1, z], [1, -1, z + 0.6], [1, 1, z], [-1, 1, z + 0.6]])
    write_geo(tmp_path / "scenes" / "geo" / "folded.geo", primid, vtxidx, vtx)
    nra = tmp_path / "scenes" / "0010_pt" / "test.nra2"
    lines = nra.read_text().splitlines()
    k = lines.index("6")
    nra.write_text("\n".join(lines[:k] + ["5", "5 ../geo/emitter", "10 ../geo/folded", "10 ../geo/cone", "10 ../geo/sphere", "10 ../geo/cylinder"]) + "\n")
    scene = make_scene(nra, width=64, height=64, max_verts=4)
    P = int(scene.desc.num_prims)
    assert P == Q + 6
    rng = np.random.default_rng(5)
    n = 64 * 700
    # half of the rays cross the valleys nearly horizontally, across the fold; the rest come from anywhere
    pos = np.zeros((n, 3), dtype=np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    h = n // 2
    ang = rng.uniform(0, 2 * np.pi, size=h)
    pos[:h] = np.stack([4 * np.cos(ang), 4 * np.sin(ang), rng.uniform(0.4, 3.2, size=h)], axis=1)
    tgt = np.stack([rng.uniform(-.8, .8, size=h), rng.uniform(-.8, .8, size=h), rng.uniform(0.5, 3.0, size=h)], axis=1)
    d[:h] = tgt - pos[:h]
    pos[h:] = rng.uniform(-4, 4, size=(n - h, 3)) + [0, 0, 2]
    d[h:] = rng.normal(size=(n - h, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ignore = rng.integers(0, P, size=n).astype(np.uint32)
    ignore[::2] = 0xffffffff
    be = pkg.Backend(scene, traversal=traversal)
    gpu = _compare_hits(scene, be, pos, d, traversal=traversal)
    assert (gpu["primid"] != 0xffffffffffffffff).mean() > 0.5
    _compare_hits(scene, be, pos, d, ignore=ignore, max_dist=rng.uniform(0.5, 30, size=n).astype(np.float32), traversal=traversal)
    be.close()
    desc = scene.desc
    keep_nodes, keep_n = desc.nodes, desc.num_nodes
    try:
        for leaf_size in (3, 7, 20, 31):
            arr = _chunk_tree(desc, leaf_size)
            desc.nodes, desc.num_nodes = C.cast(arr, C.POINTER(pkg.MiNode)), len(arr)
            be = pkg.Backend(scene, traversal=traversal)
            _compare_hits(scene, be, pos[:64 * 200], d[:64 * 200], traversal=traversal)
            _compare_hits(scene, be, pos[:64 * 200], d[:64 * 200], ignore=ignore[:64 * 200], traversal=traversal)
            be.close()
    finally:
        desc.nodes, desc.num_nodes = keep_nodes, keep_n


@pytest.mark.parametrize("reduce", ["peer", "rccl"])
def test_group_behind_the_c_abi(monkeypatch, reduce):
    """several GPUs from ONE host thread through the C ABI (mi_group_*, corona_mi.h; the reference's dispatcher src/view.c:630-695):
    a group of one is the single render; a group of two members on device 0 (two copies of the scene, the frame's path indices
    split in two contiguous ranges, framebuffers added up on member 0 by peer copy + add kernel) equals the single render up to the
    order of the float atomics; with one member the RCCL code path (dlopen'ed librccl, ncclCommInitAll + ncclReduce) runs too.
    Accumulating read-back (mi_fb_read(accumulate = 1), the reference host's per-batch path) adds frame after frame."""
    monkeypatch.setenv("CORONA_MI_GROUP_REDUCE", reduce)
    scene = make_scene(SCENE_0010, width=512, height=288, max_verts=8)
    per = 4 * scene.width * scene.height
    single = pkg.Backend(scene, counters=False)
    single.render(0, per)
    ref = single.fb_read()
    one = pkg.Group(scene, [0])
    assert one.uses_rccl() == (reduce == "rccl")
    one.render(0, per)
    img = one.fb_read()
    assert np.abs(img - ref).max() <= 1e-4 * ref.max() and one.counters()[4] == per
    # progressive: two more frames accumulated on the host equal a render of the three frames' indices
    acc = img.copy()
    for k in (1, 2):
        one.fb_clear()
        one.render(k * per, per)
        one.fb_read(accumulate_into=acc)
    single.fb_clear()
    single.render(0, 3 * per)
    ref3 = single.fb_read()
    assert np.abs(acc - ref3).max() <= 2e-4 * ref3.max()
    one.close()
    if reduce == "peer":
        two = pkg.Group(scene, [0, 0])
        assert not two.uses_rccl()
        two.render(0, 3 * per)
        got = two.fb_read()
        assert np.abs(got - ref3).max() <= 2e-4 * ref3.max() and two.counters()[4] == 3 * per
        again = two.fb_read()                          # the reduce cleared member 1: reading twice adds nothing
        assert np.array_equal(again, got)
        # reduce, reduce, read without a render or a sync in between: the root's second copy of member 1's framebuffer has to wait for
        # the clear the first reduce queued behind it (round 3 ordered it only against the stale `rendered` event: values counted twice)
        for rep in range(3):
            two.fb_clear()
            two.render(0, 3 * per)
            two.reduce()
            two.reduce()
            twice = two.fb_read()
            assert np.abs(twice - ref3).max() <= 2e-4 * ref3.max(), rep
        two.close()
    else:
        with pytest.raises(RuntimeError, match="not distinct"):
            pkg.Group(scene, [0, 0])                   # RCCL needs distinct devices: asked for explicitly, that is an error, not a silent fallback
    single.close()


@pytest.mark.parametrize("sampler,points", [(pkg.MI_SAMPLER_PT, pkg.MI_POINTS_RAND), (pkg.MI_SAMPLER_PTDL, pkg.MI_POINTS_RAND), (pkg.MI_SAMPLER_PT, pkg.MI_POINTS_HALTON)],
                         ids=["pt", "ptdl", "pt halton"])
def test_pixels_from_path_indices(sampler, points, traversal):
    """MI_PIXELS_FROM_INDEX, the hook of render_sample_path's tiled branch (src/render.d/gi.c:88-95; path_set_pixel, include/pathspace.h:355-360;
    camera_sample, src/camera.d/thinlens.c:117-118): path i starts inside pixel (i mod W H), at the position its two film numbers name there, with
    its generator seeded through a hash of the index (corona_mi.h says why). Path for path against the oracle's restatement of the mode; the
    sampled mode is back afterwards."""
    scene = make_scene(SCENE_0010, width=640, height=352, max_verts=8, sampler=sampler, pointsampler=points)
    W, H = scene.width, scene.height
    be = pkg.Backend(scene, traversal=traversal)
    be.set_pixels(True)
    first, n = 3 * W * H - 7000, 40000                    # across a frame boundary
    gpu = be.trace_paths(first, n)
    with oracle_pixels():
        ora = oracle_records(scene, first, n)
    idx = (first + np.arange(n)) % (W * H)
    assert np.array_equal(np.floor(gpu["pixel_i"]), (idx % W).astype(np.float32)) and np.array_equal(np.floor(gpu["pixel_j"]), (idx // W).astype(np.float32))
    assert np.abs(gpu["pixel_i"] - ora["pixel_i"]).max() <= 1e-4 and np.abs(gpu["pixel_j"] - ora["pixel_j"]).max() <= 1e-4
    frac = gpu["pixel_i"] - np.floor(gpu["pixel_i"])
    assert 0.45 < frac.mean() < 0.55 and frac.std() > 0.25           # ... somewhere inside the pixel, not at its corner
    for f in ("lambda", "time", "scramble"):
        assert np.abs(gpu[f] - ora[f]).max() <= 1e-5, f
    same = gpu["length"] == ora["length"]
    assert (~same).sum() <= 2, (~same).sum()
    assert (gpu["num_splats"] != ora["num_splats"]).sum() <= (2 if sampler == pkg.MI_SAMPLER_PT else 8)
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if m.sum():
            assert (gpu["v"]["prim"][m, k] != ora["v"]["prim"][m, k]).sum() <= 2
            assert (rel(gpu["v"]["throughput"][m, k], ora["v"]["throughput"][m, k]) >= 1e-3).sum() <= max(3, int(1e-3 * m.sum()))
    # ... and the paths differ from the sampled mode's (other pixels, other random numbers)
    be.set_pixels(False)
    sampled = be.trace_paths(first, n)
    assert (sampled["pixel_i"] != gpu["pixel_i"]).mean() > 0.99
    assert (sampled["length"] != oracle_records(scene, first, n)["length"]).sum() <= 2
    be.close()


@pytest.mark.parametrize("sampler", [pkg.MI_SAMPLER_PT, pkg.MI_SAMPLER_PTDL], ids=["pt", "ptdl"])
def test_tile_owned_sharding(sampler, counters):
    """mi_render_tiles: member g of G renders the 32 x 32 tiles t = g (mod G) (tile scheme of include/render_tiles.h:148-170); the members'
    framebuffers add up to the render of the frames' index range in MI_PIXELS_FROM_INDEX mode -- the same paths, whoever renders them --,
    a member's own image is empty outside its tiles and their two-pixel rim (the 4 x 4 filter footprint), and both equal the oracle's."""
    scene = make_scene(SCENE_0010, width=320, height=224, max_verts=8, sampler=sampler)
    W, H = scene.width, scene.height
    frames, G = 6, 3
    be = pkg.Backend(scene, counters=counters)
    with pytest.raises(RuntimeError, match="samples its pixels"):
        be.render_tiles(0, 1)
    be.set_pixels(True)
    be.render(2 * W * H, frames * W * H)
    whole = be.fb_read()
    c0 = be.counters()
    parts = []
    for g in range(G):
        be.fb_clear()
        be.render_tiles(2, frames, g, G)
        parts.append(be.fb_read())
    c1 = be.counters()
    assert c1[4] - c0[4] == frames * W * H                   # every path of the frames exactly once
    if counters:
        assert all(abs((c1[k] - c0[k]) - c0[k]) <= 1e-6 * c0[k] for k in (0, 1, 3))      # ... with the same rays, node visits, primitive tests
    total = sum(parts)
    assert np.abs(total - whole).max() <= 2e-4 * whole.max()
    ty, tx = np.arange(H) // 32, np.arange(W) // 32
    owner = (ty[:, None] * (W // 32) + tx[None, :]) % G
    for g in range(G):
        inside = owner == g
        rim = np.zeros_like(inside)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                rim |= np.roll(np.roll(inside, dy, axis=0), dx, axis=1)
        assert parts[g][~rim].sum() == 0.0                    # nothing lands further than the filter reaches
        assert parts[g][inside].sum() > 0.8 * parts[g].sum()
    with oracle_pixels():
        ofb = sum(oracle_render_tiles(scene, 2, frames, g, G, threads=8)[0] for g in range(G))
    gain = scene.gain(frames)
    assert np.sqrt((((total - ofb) * gain) ** 2).sum() / (W * H)) < 0.05
    assert np.allclose(total.sum(axis=(0, 1)), ofb.sum(axis=(0, 1)), rtol=2e-3)
    # one member of one: all tiles; more members than tiles: the surplus members render nothing
    be.fb_clear(); be.render_tiles(2, frames, 0, 1)
    assert np.abs(be.fb_read() - whole).max() <= 2e-4 * whole.max()
    be.fb_clear(); be.render_tiles(2, frames, 100, 101)
    assert be.fb_read().sum() == 0.0
    with pytest.raises(RuntimeError):
        be.render_tiles(0, 1, 3, 3)
    be.close()


def test_tile_sharding_statistics_against_the_reference_render():
    """the tiled branch is `#if 0` in the reference's default build, so there are no reference dumps of its paths: it is pinned statistically.
    2048 spp rendered through mi_group_render_tiles (two members on this GPU, tiles 0 / 1 (mod 2), framebuffers reduced) against the real
    reference's 2048-spp render of the same film (tests/golden/tilemeans_pt_mv8.npz, sampled pixels): tile means as unit noise around the
    reference's, the same robust statistics as test_full_size_properties_cfg2 asserts for the sampled mode."""
    g = np.load(GOLDEN / "tilemeans_pt_mv8.npz")
    scene = make_scene(SCENE_0010, width=int(g["width"]), height=int(g["height"]), max_verts=int(g["max_verts"]))
    assert (scene.height // 32, scene.width // 32, 3) == g["tiles"].shape
    two = pkg.Group(scene, [0, 0])
    rspp, Q = int(g["spp"]), 8
    parts = []
    for q in range(Q):
        two.fb_clear()
        two.render_tiles(q * rspp // Q, rspp // Q)
        parts.append((two.fb_read() * scene.gain(rspp // Q)).reshape(scene.height // 32, 32, scene.width // 32, 32, 3).mean(axis=(1, 3)))
    assert two.counters()[4] == rspp * scene.width * scene.height
    parts = np.array(parts)
    tiles = parts.mean(axis=0)
    var = parts.var(axis=0, ddof=1) / Q
    z = (tiles - g["tiles"]) / np.sqrt(2 * var)
    width = 1.4826 * np.median(np.abs(z - np.median(z)))
    assert abs(np.median(z)) < 0.1, np.median(z)
    assert 0.85 < width < 1.25, width
    assert (np.abs(z) > 4).mean() < 0.012             # (sampled mode: < 0.01; one frame per pixel and step leaves the fireflies of a tile a little less averaged)
    assert np.all(np.abs(tiles.mean(axis=(0, 1)) / g["tiles"].mean(axis=(0, 1)) - 1) < 3e-3)
    assert np.corrcoef(tiles[8:, :, 1].ravel(), g["tiles"][8:, :, 1].ravel())[0, 1] > 0.995
    two.close()


def test_bsdf_battle_test_on_the_device():
    """the reference's own BSDF test (tools/battle-test.c -- regression/0052_dielectric, 0053_dielectric; goldens from the tool built
    in oracle/_ref, tests/golden/make_battle_golden.py) reproduced with the kernels' sample / eval / pdf functions: rough dielectric
    1.7 / 73 at roughness 0.4, reflected and transmitted hemisphere, and gold at roughness 0.3, four incidence angles, 525 nm.
    Every one of the reference's four integrals per angle is matched to its own pass criterion (difference^2 < 1e-5,
    regression/makebattletest.sh:13-14) -- the estimate from sample() against the reference's estimate, the evaluation against the
    reference's evaluation -- so where the reference passes its test the device does, and where the reference's own sample() and
    brdf() disagree (transmission at grazing angles: .876 vs .856, .804 vs .717; gold at normal incidence) the device shows the same
    disagreement. 16 x the reference's sample count keeps the Monte Carlo error of the estimates below 1e-4."""
    cases = json.loads((GOLDEN / "battle.json").read_text())
    scene = make_scene(SCENE_0010, width=64, height=64, max_verts=4)
    be = pkg.Backend(scene)
    for c in cases:
        ref = np.array(c["rows"])
        metal = c["bsdf"] == "metal"
        # the reference BUILD's metal sample() ends samples its formula does not (a NaN of its compiled Fresnel term, corona_mi.h:
        # mi_scene_set_metal_reference): with the switch on all four integrals are the reference's; with it off (the default) the
        # evaluation still is, and the sampler agrees with the evaluation instead -- the device passes the test the reference fails
        for reference_build in ((True, False) if metal else (False,)):
            be.set_metal_reference(reference_build)
            got = be.bsdf_test(c["bsdf"], c["param"], c["roughness"], c["reflect"], count=c["count"], lambda_=c["lambda"], size=c["size"], spp=128)
            passes = np.stack([(got[:, 1] - got[:, 0]) ** 2 < 1e-5, (got[:, 3] - got[:, 2]) ** 2 < 1e-5], axis=1)
            if metal and not reference_build:
                assert np.all((got[:, [1, 3]] - ref[:, [1, 3]]) ** 2 < 1e-5), (c["name"], got, ref)       # brdf() and pdf() integrals
                assert np.all((got[:, 1] - got[:, 0]) ** 2 < 1e-5), (c["name"], got)                          # sample() consistent with brdf()
                assert np.all(got[:, 2] >= ref[:, 2] - 1e-3)                                                 # and it loses no samples the reference keeps
                continue
            assert np.all((got - ref) ** 2 < 1e-5), (c["name"], reference_build, got, ref)
            # the verdict per angle and quantity equals the reference's, except where the reference sits within noise of its own threshold
            margin = np.stack([np.abs((ref[:, 1] - ref[:, 0]) ** 2 - 1e-5), np.abs((ref[:, 3] - ref[:, 2]) ** 2 - 1e-5)], axis=1)
            clear = margin > 3e-6
            assert np.array_equal(passes[clear], np.array(c["reference_passes_bsdf_pdf"])[clear]), (c["name"], passes, c["reference_passes_bsdf_pdf"])
    be.set_metal_reference(False)
    # 0052 as the reference runs it passes outright
    assert all(all(p) for p in cases[0]["reference_passes_bsdf_pdf"])
    be.close()


@pytest.mark.parametrize("reference_build", [False, True], ids=["formula", "reference-build"])
def test_metal_image_against_the_reference_render(reference_build):
    """scenes/0053_metal (gold, roughness 0.3, on sphere / cone / cylinder) with ptdl at the sample count of the reference render
    (tests/golden/tilemeans_metal_ptdl_mv8.npz: 1024 spp by the real reference binary): tile means as unit noise around the
    reference's, image means within 0.3 %. Run with the metal sampler as written (default) and with the reference BUILD's behaviour
    (mi_scene_set_metal_reference: 2-4 % of the samples at a gold vertex ended by a NaN of its compiled Fresnel term): both pass --
    the ended samples carry so little energy in this scene that the image means of the two modes differ by 3e-5 (measured),
    a hundred times less than the noise of the comparison. The sphere's own tiles only agree since its intersection carries the
    reference build's roundings (sphere_t, mi_kernels.h): with the plain evaluation they came out 4-7 % darker (z = -5 .. -7)."""
    g = np.load(GOLDEN / "tilemeans_metal_ptdl_mv8.npz")
    scene = make_scene(SCENE_METAL, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    be = pkg.Backend(scene, counters=False)
    be.set_metal_reference(reference_build)
    per, rspp, Q = scene.width * scene.height, int(g["spp"]), 8
    parts = []
    for q in range(Q):
        be.fb_clear()
        be.render((9000 + q * rspp // Q) * per, rspp // Q * per)
        parts.append((be.fb_read() * scene.gain(rspp // Q)).reshape(scene.height // 32, 32, scene.width // 32, 32, 3).mean(axis=(1, 3)))
    be.close()
    parts = np.array(parts)
    tiles = parts.mean(axis=0)
    var = parts.var(axis=0, ddof=1) / Q
    z = (tiles - g["tiles"]) / np.sqrt(2 * var)
    width = 1.4826 * np.median(np.abs(z - np.median(z)))
    assert abs(np.median(z)) < 0.1 and 0.85 < width < 1.3, (np.median(z), width)
    assert (np.abs(z) > 4).mean() < 0.02
    assert np.all(np.abs(tiles.mean(axis=(0, 1)) / g["tiles"].mean(axis=(0, 1)) - 1) < 3e-3)
    # the sphere fills the tiles around row 9..11, column 17..22 (first-hit statistics of the oracle): no cluster of dark tiles there
    zs = z[9:12, 17:23, 1]
    assert np.abs(np.median(zs)) < 1.0 and (zs < -4).sum() <= 1, zs


@pytest.mark.parametrize("scene_path,key", [(SCENE_MB, "mb_pt_mv8"), (SCENE_MB_ROUND, "mb_round_pt_mv8")])
def test_motion_blur_traversal_work_equals_the_reference(scene_path, key):
    """moving geometry with the reference's time-interpolated node boxes (mi_scene_desc.nodes_t1: the shutter-close boxes next to the
    shutter-open ones, interpolated per ray in the motion-blur kernels; src/accel.d/qbvhmp.c:1208-1224): one sample per pixel of
    the 1280x720 film does the reference's traversal work -- rays, node visits, box hits, primitive tests within 3e-3 of its own
    -DACCEL_DEBUG totals (tests/golden/counters.json) and equal to the oracle's; and the same image as the oracle."""
    gold = json.loads((GOLDEN / "counters.json").read_text())[key]
    scene = make_scene(scene_path, width=1280, height=720, max_verts=8)
    assert bool(scene.desc.nodes_t1)
    n = scene.width * scene.height
    be = pkg.Backend(scene, traversal="exact")
    be.render(0, n)
    fb, cnt = be.fb_read(), be.counters()
    be.close()
    for k, name in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
        assert abs(cnt[k] - gold[name]) <= 3e-3 * gold[name], (name, cnt[k], gold[name])
    ofb, ocnt, _ = oracle_render(scene, 0, n, threads=8)
    for k in range(4):
        assert abs(cnt[k] - ocnt[k]) <= 1e-3 * ocnt[k], (k, cnt[k], ocnt[k])
    assert np.sqrt((((fb - ofb) * scene.gain(1)) ** 2).sum() / n) < 0.05


@pytest.mark.parametrize("scene_path,key", [(SCENE_MB, "mb_pt_mv8"), (SCENE_MB_ROUND, "mb_round_pt_mv8")])
def test_device_built_tree_of_moving_geometry(monkeypatch, scene_path, key):
    """the device build on a scene with moving primitives (round 4): shutter-open and shutter-close boxes per node, refitted on one topology
    as the reference does (src/accel.d/qbvhmp.c:259-283) and interpolated per ray -- the same paths as the oracle on the reference's
    tree; node visits within 10 % of the reference's own -DACCEL_DEBUG count on ITS tree (measured 1.016 x with the SAH refinement by
    tree rotations, csrc/mi_build.h; 1.104 x for the plain LBVH, CORONA_MI_BUILD_SAH=0) and fewer primitive tests; with one box around
    both states (CORONA_MI_BUILD_T1=0, rounds 1-3) clearly more of both."""
    gold = json.loads((GOLDEN / "counters.json").read_text())[key]
    scene = make_scene(scene_path, width=1280, height=720, max_verts=8)
    n = scene.width * scene.height
    work = {}
    monkeypatch.setenv("CORONA_MI_BUILD_LEAF", "4")
    be = pkg.Backend(scene, traversal="exact", device_build=True)
    c0 = be.counters(); be.render(0, n); be.sync()
    leaf4 = [b - a for a, b in zip(c0, be.counters())]
    be.close()
    assert leaf4[1] <= 1.10 * gold["node_visits"] and leaf4[3] <= gold["prim_tests"], (leaf4, gold)
    monkeypatch.delenv("CORONA_MI_BUILD_LEAF")
    for t1 in ("1", "0", "lbvh"):
        monkeypatch.setenv("CORONA_MI_BUILD_T1", "0" if t1 == "0" else "1")
        monkeypatch.setenv("CORONA_MI_BUILD_SAH", "0" if t1 == "lbvh" else "2")
        be = pkg.Backend(scene, traversal="exact", device_build=True)
        assert be.stats()["device_built"]
        if t1 == "1":
            m = 20000
            gpu, ora = be.trace_paths(0, m), oracle_records(scene, 0, m)
            same = gpu["length"] == ora["length"]
            for k in range(1, 8):
                sel = ora["length"] > k
                same &= ~sel | (gpu["v"]["prim"][:, k] == ora["v"]["prim"][:, k])        # the reference's packed primid: does not depend on the tree
            assert (~same).sum() <= 2, int((~same).sum())
        c0 = be.counters()
        be.render(0, n)
        be.sync()
        work[t1] = [b - a for a, b in zip(c0, be.counters())]
        be.close()
    assert abs(work["1"][0] - work["0"][0]) <= 1e-5 * work["0"][0]            # the same rays (a tie between two primitives at one distance may fall either way: 1 ray in 2.3 M)
    assert abs(work["1"][0] - gold["rays"]) <= 3e-3 * gold["rays"]
    assert work["1"][1] <= 1.05 * gold["node_visits"], (work["1"][1], gold["node_visits"])
    assert work["1"][3] <= gold["prim_tests"], (work["1"][3], gold["prim_tests"])
    assert work["0"][1] >= 1.03 * work["1"][1] and work["0"][3] >= 1.10 * work["1"][3], (work["0"], work["1"])
    assert work["lbvh"][1] >= 1.04 * work["1"][1], (work["lbvh"], work["1"])          # what the rotations are for
    assert abs(work["lbvh"][0] - work["1"][0]) <= 1e-5 * work["1"][0]


@pytest.mark.gpu
def test_device_build_sah_refinement(monkeypatch):
    """the SAH refinement of the device build (tree rotations between refit and collapse, csrc/mi_build.h; the reference gets its tree quality
    from a binned SAH sweep, src/accel.d/qbvhmp.c:425-525): on cfg 2's scene the refined tree needs FEWER node visits than the reference's
    own tree (measured 0.957 x its -DACCEL_DEBUG count; the plain LBVH 1.018 x) and fewer primitive tests, for the same rays and -- closest
    hits do not depend on the tree -- the same paths."""
    gold = json.loads((GOLDEN / "counters.json").read_text())["pt_mv8"]
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    n = scene.width * scene.height
    work, recs = {}, {}
    for sah in ("0", "2", "6"):
        monkeypatch.setenv("CORONA_MI_BUILD_SAH", sah)
        be = pkg.Backend(scene, traversal="exact", device_build=True)
        assert be.stats()["device_built"]
        recs[sah] = be.trace_paths(0, 20000)
        c0 = be.counters(); be.render(0, n); be.sync()
        work[sah] = [b - a for a, b in zip(c0, be.counters())]
        be.close()
    ora = oracle_records(scene, 0, 20000)
    for sah in recs:
        same = recs[sah]["length"] == ora["length"]
        for k in range(1, 8):
            sel = ora["length"] > k
            same &= ~sel | (recs[sah]["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
        assert (~same).sum() <= 2, (sah, int((~same).sum()))
        assert abs(work[sah][0] - gold["rays"]) <= 3e-3 * gold["rays"]
    assert work["2"][1] <= 0.98 * gold["node_visits"] and work["2"][3] <= 0.85 * gold["prim_tests"], (work["2"], gold)
    assert work["2"][1] <= 0.97 * work["0"][1], (work["2"][1], work["0"][1])
    assert work["6"][1] <= work["2"][1] * 1.005, (work["6"][1], work["2"][1])        # more passes: converged, not worse


@pytest.mark.gpu
def test_ptdl_scene_without_emitters(tmp_path):
    """a ptdl scene whose shapes emit nothing (lights.num_prims == 0: no emitter records on the device) -- the plain ptdl kernels lay their
    LDS out with room for the records whether or not there are any, so the host has to allocate it by the same rule (round 3 took the
    bytes off when d_lights was NULL: the job lists of the last waves lay outside the allocation). Paths against the oracle, black image."""
    import shutil
    src = SCENE_0010.parent
    dst = tmp_path / "0010_dark"
    shutil.copytree(src, dst)
    nra = (dst / "test.nra2").read_text().replace("color e 3200 3200 3200 1.", "color d 0.5 0.5 0.5").replace("color e 10 10 10 1.", "color d 0.5 0.5 0.5")
    (dst / "test.nra2").write_text(nra.replace("../geo/", str(SCENE_0010.parent.parent / "geo") + "/"))       # the geometry stays where it is
    scene = make_scene(dst / "test.nra2", width=256, height=144, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    assert scene.desc.lights.num_prims == 0 and scene.desc.num_shapes == 6 and scene.desc.num_prims > 4000
    n = 20000
    ora = oracle_records(scene, 0, n)
    for mode in ("exact", "fast"):
        be = pkg.Backend(scene, traversal=mode)
        gpu = be.trace_paths(0, n)
        same = (gpu["length"] == ora["length"]) & (gpu["num_splats"] == ora["num_splats"])
        for k in range(1, 8):
            m = ora["length"] > k
            same &= ~m | (gpu["v"]["prim"][:, k] == ora["v"]["prim"][:, k])
        assert (~same).sum() <= 1, (mode, int((~same).sum()))
        be2 = pkg.Backend(scene, traversal=mode, counters=False)
        be2.render(0, 4 * scene.width * scene.height)
        assert float(np.abs(be2.fb_read()).max()) == 0.0 and be2.counters()[4] == 4 * scene.width * scene.height
        be.close(); be2.close()
