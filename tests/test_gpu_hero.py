"""Hero wavelengths on the device (mi_scene_set_wavelengths, corona-13_amd/csrc/mi_hero.h) against the oracle's restatement of the reference built
with -DMF_COUNT=4 (oracle_hero_trace; itself pinned to that build's per-path dumps: tests/test_oracle_hero.py): path for path, all four
components of every spectral quantity."""
import numpy as np
import pytest

from helpers import GOLDEN, SCENE_0010, SCENE_ALL, SCENE_CAM_MB, SCENE_FINE, SCENE_FOG, SCENE_MB, SCENE_MB_LIGHT, SCENE_MB_ROUND, SCENE_MB_ROUND_LIGHT, SCENE_MEDIA, SCENE_METAL, SCENE_NESTED, SCENE_ROUGH, SCENE_SMOOTH, load_pkg, make_scene, oracle_hero_records, oracle_lib, oracle_records

pkg = load_pkg()
pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))


HERO_GPU_CASES = [
    ("pt 1280x720 mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8, 60000),
    ("ptdl 1280x720 mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, 40000),
    ("rough dielectric pt mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PT, 32, 20000),
    ("rough dielectric ptdl mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PTDL, 32, 10000),
    ("smooth glass ptdl mv8 (one component survives a specular transmission)", SCENE_SMOOTH, pkg.MI_SAMPLER_PTDL, 8, 40000),
    ("smooth glass pt mv8", SCENE_SMOOTH, pkg.MI_SAMPLER_PT, 8, 20000),
    ("metal pt mv8", SCENE_METAL, pkg.MI_SAMPLER_PT, 8, 20000),
    ("metal ptdl mv8", SCENE_METAL, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("fine backdrop (tree in HBM) ptdl mv8", SCENE_FINE, pkg.MI_SAMPLER_PTDL, 8, 10000),
    # MOD_pointsampler = halton: one number for the four wavelength draws (components a quarter of the range apart); depth 32 leaves the tables
    ("halton ptdl mv8", SCENE_0010, pkg.MI_SAMPLER_PTDL, 8, 40000),
    ("halton pt mv8", SCENE_0010, pkg.MI_SAMPLER_PT, 8, 30000),
    ("halton rough dielectric ptdl mv32", SCENE_ROUGH, pkg.MI_SAMPLER_PTDL, 32, 10000),
    # the extended kernels: media (the hero's medium samples the free-flight distance, transmittance and pdf per component), a moving camera,
    # moving geometry and emitters -- the HERO x MEDIA x MB x NORG instantiations
    ("media ptdl mv8", SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, 8, 40000),
    ("media pt mv32", SCENE_MEDIA, pkg.MI_SAMPLER_PT, 32, 20000),
    ("fog ptdl mv8", SCENE_FOG, pkg.MI_SAMPLER_PTDL, 8, 30000),
    ("fog pt mv8", SCENE_FOG, pkg.MI_SAMPLER_PT, 8, 30000),
    ("nested media ptdl mv32", SCENE_NESTED, pkg.MI_SAMPLER_PTDL, 32, 10000),
    ("nested media pt mv8", SCENE_NESTED, pkg.MI_SAMPLER_PT, 8, 30000),
    ("camera motion blur ptdl mv8", SCENE_CAM_MB, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("moving geometry ptdl mv8", SCENE_MB, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("moving geometry pt mv8", SCENE_MB, pkg.MI_SAMPLER_PT, 8, 20000),
    ("moving geometry and emitter ptdl mv8", SCENE_MB_LIGHT, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("moving geometry: sphere, cone, cylinder ptdl mv8", SCENE_MB_ROUND, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("moving sphere and cone as emitters ptdl mv8", SCENE_MB_ROUND_LIGHT, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("halton everything at once ptdl mv8", SCENE_ALL, pkg.MI_SAMPLER_PTDL, 8, 20000),
    ("moving geometry: everything at once pt mv32", SCENE_ALL, pkg.MI_SAMPLER_PT, 32, 10000),
]


@pytest.mark.parametrize("name,scene_path,sampler,mv,n", HERO_GPU_CASES)
def test_hero_paths_match_oracle(name, scene_path, sampler, mv, n):
    halton = name.startswith("halton")
    scene = make_scene(scene_path, width=1280, height=720, max_verts=mv, sampler=sampler, pointsampler=pkg.MI_POINTS_HALTON if halton else pkg.MI_POINTS_RAND)
    be = pkg.Backend(scene)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    first = 4321
    gpu, gext = be.trace_paths_hero(first, n)
    be.close()
    ora, oext = oracle_hero_records(scene, first, n)
    assert np.array_equal(gpu["index"], ora["index"])
    for f in ("pixel_i", "pixel_j", "lambda", "time", "scramble"):
        assert np.abs(gpu[f] - ora[f]).max() <= 1e-5, f
    assert np.abs(gext["lambda"] - oext["lambda"]).max() <= 1e-4             # fmodf + the range product, four draws
    if halton:
        d = (gext["lambda"][:, 1:] - gext["lambda"][:, :1]) % 470.0
        assert np.abs(d - np.array([117.5, 235.0, 352.5], dtype=np.float32)).max() < 1e-2
    # the bounds of the scalar kernels' parity test (test_gpu_parity.py: test_paths_match_oracle)
    allowed = max(2, int(np.ceil(2e-5 * n)))
    same = gpu["length"] == ora["length"]
    assert (~same).sum() <= allowed, (~same).sum()
    allowed_splats = allowed if sampler == pkg.MI_SAMPLER_PT else int(np.ceil((2e-4 if mv <= 8 else 1e-3) * n))
    assert (gpu["num_splats"] != ora["num_splats"]).sum() <= allowed_splats
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if not m.sum():
            continue
        outliers = max(3, int(np.ceil(1e-3 * m.sum())))
        assert (gpu["v"]["prim"][m, k] != ora["v"]["prim"][m, k]).sum() <= allowed
        assert (gpu["v"]["mode"][m, k] != ora["v"]["mode"][m, k]).sum() <= allowed
        assert (gpu["v"]["flags"][m, k] != ora["v"]["flags"][m, k]).sum() <= allowed
        dx = np.abs(gpu["v"]["x"][m, k] - ora["v"]["x"][m, k]).max(axis=1)
        assert (dx >= (2e-3 if k <= 2 else 1e-2)).sum() <= outliers
        # the record holds component 0 ...
        assert np.array_equal(gpu["v"]["throughput"][m, k], gext["throughput"][m, k, 0], equal_nan=True) and np.array_equal(gpu["v"]["pdf"][m, k], gext["pdf"][m, k, 0], equal_nan=True)
        # ... and the extension all four: throughput, pdf, shading, index of refraction, each against the oracle's
        # (a moving camera's frame / a moving primitive's vertices come out of acosf / sinf and products per path: the last-ulp libm difference sits
        #  on every vertex from the start -- the scalar kernels' bounds, test_gpu_parity.py)
        moving = name.startswith(("camera motion blur", "moving", "halton everything"))
        assert (rel(gext["throughput"][m, k], oext["throughput"][m, k]).max(axis=1) >= (2e-2 if moving else 1e-3)).sum() <= outliers, k
        assert (rel(gext["pdf"][m, k], oext["pdf"][m, k]).max(axis=1) >= (2e-2 if moving else 5e-3)).sum() <= outliers, k
        for f in ("rd", "rg", "em", "eta"):
            assert (rel(gext[f][m, k], oext[f][m, k]).max(axis=1) >= 1e-4).sum() <= outliers, (f, k)
        # which components a vertex has zeroed (specular transmission: all but the last) is a decision, not arithmetic
        assert ((gext["throughput"][m, k] == 0) != (oext["throughput"][m, k] == 0)).any(axis=1).sum() <= allowed
    m = same & (gpu["num_splats"] == ora["num_splats"]) & (ora["num_splats"] > 0)
    if m.sum():
        for k in range(int(ora["num_splats"][m].max())):
            mk = m & (ora["num_splats"] > k)
            a, b = gext["splat_value"][mk, k], oext["splat_value"][mk, k]
            # deep connections: the reference's own MIS products leave the float range (inf / inf = NaN, view_splat drops them) on both sides alike --
            # where a product sits AT the limit, expf / the order of a product decides (device libm against glibc): a few rows per thousand at depth 32
            assert (np.isnan(a) != np.isnan(b)).any(axis=1).sum() <= max(2, int(np.ceil((1e-3 if mv <= 8 else 5e-3) * len(a)))), k
            fin = np.isfinite(a) & np.isfinite(b)
            if fin.sum() >= 200:
                assert np.quantile(rel(a[fin], b[fin]), 0.99) < 1e-3, k
            ca, cb = gpu["splat"]["col"][mk, k], ora["splat"]["col"][mk, k]
            f3 = np.isfinite(ca).all(axis=1) & np.isfinite(cb).all(axis=1)
            if f3.sum() >= 200:
                d = np.abs(ca[f3] - cb[f3]).max(axis=1) / np.maximum(np.abs(cb[f3]).max(axis=1), 1e-20)
                assert np.quantile(d, 0.99) < 1e-3, k


def test_hero_image_matches_oracle():
    """the framebuffer of a hero render = the oracle's, splat for splat (one sample per pixel of a small film; float atomics reorder the sums)"""
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    n = scene.width * scene.height
    be = pkg.Backend(scene)
    be.set_counters(True)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.render(0, n)
    fb = be.fb_read()
    cnt = be.counters()
    ofb = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    import ctypes as C
    ocnt = (C.c_uint64 * 8)()
    oracle_lib().oracle_hero_trace(scene.desc_ptr, 0, n, None, None, ofb.ctypes.data, ocnt)
    assert cnt[4] == n
    assert abs(cnt[5] - ocnt[5]) <= max(3, 2e-4 * ocnt[5]), (cnt[5], ocnt[5])          # splats
    assert abs(cnt[6] - ocnt[6]) <= max(3, 1e-4 * ocnt[6]), (cnt[6], ocnt[6])          # vertices
    # the traversal work of the four-wavelength render: one ray per path segment and connection, the exact rounds' node visits / box hits / primitive tests
    # (shadow rays towards planar emitters stop at the first occluder, MI_ANYHIT: a little LESS work than the oracle's closest-hit loop; measured 0.18 %)
    assert abs(cnt[0] - ocnt[0]) <= 1e-3 * ocnt[0], (cnt[0], ocnt[0])
    for k in (1, 2, 3):
        assert 0.98 * ocnt[k] <= cnt[k] <= (1 + 1e-3) * ocnt[k], (k, cnt[k], ocnt[k])
    tot, otot = fb.reshape(-1, 3).sum(axis=0), ofb.reshape(-1, 3).sum(axis=0)
    assert np.abs(tot / otot - 1.0).max() < 2e-3, (tot, otot)
    d = np.abs(fb - ofb).max(axis=2)
    assert (d > 1e-3 * max(1.0, float(ofb.max()))).mean() < 2e-3
    # and it is a different estimate than the scalar render of the same indices, with the same expectation (tests/golden/mf4_vs_mf1_measured.json)
    be.set_wavelengths(1)
    be.fb_clear()
    be.render(0, n)
    fb1 = be.fb_read()
    assert np.abs(fb1 - fb).max() > 0
    be.close()


def test_hero_render_converges_to_the_scalar_render():
    """same expectation, two estimators: the means of two 512-sample renders agree within noise (the reference's own MF_COUNT = 4 / 1 images at
    1024 x 576 x 64: 0.7 %; measured here at 64 samples of this small film: 2.2 / 1.5 / 0.9 %)"""
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    n = scene.width * scene.height * 512
    be = pkg.Backend(scene)
    be.render(0, n)
    m1 = be.fb_read().reshape(-1, 3).mean(axis=0)
    be.fb_clear()
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.render(0, n)
    m4 = be.fb_read().reshape(-1, 3).mean(axis=0)
    assert np.abs(m4 / m1 - 1.0).max() < 0.02, (m4, m1)
    be.close()


def test_hero_switch():
    """a scalar scene says so at the hero entry point; only 1 and 4 wavelengths exist; switching back and forth leaves the scalar kernels' paths untouched"""
    scene = make_scene(SCENE_FOG, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PT)
    be = pkg.Backend(scene)
    with pytest.raises(RuntimeError):
        be.trace_paths_hero(0, 16)
    be.close()
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PT)
    be = pkg.Backend(scene)
    with pytest.raises(RuntimeError):
        be.set_wavelengths(3)
    # switching back and forth leaves the scalar kernels' paths untouched
    a = be.trace_paths(0, 2000)
    be.set_wavelengths(4)
    h, _ = be.trace_paths_hero(0, 2000)
    be.set_wavelengths(1)
    b = be.trace_paths(0, 2000)
    assert a.tobytes() == b.tobytes() and (h["pixel_i"] != a["pixel_i"]).mean() > 0.99
    assert be.kernel_name().endswith("false>")
    be.set_wavelengths(4)
    assert be.kernel_name().endswith("true>")
    be.close()


def test_hero_sharding_by_tiles_and_by_group(monkeypatch):
    """the two ways a job is shared between GPUs, with hero paths: tile-owned sharding (mi_render_tiles: two members' renders add up to the single
    render, each splatting into its own pixels only up to the filter's reach) and index ranges behind the C ABI (mi_group of two members)"""
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    frames = 4
    be = pkg.Backend(scene, counters=False)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.set_pixels(True)
    be.render_tiles(0, frames)
    whole = be.fb_read()
    be.fb_clear()
    be.render_tiles(0, frames, 0, 2)
    be.render_tiles(0, frames, 1, 2)
    parts = be.fb_read()
    assert np.abs(parts - whole).max() <= 2e-4 * whole.max() and whole.max() > 0
    # ... and it is the index-range render of the same frames in pixel mode
    be.fb_clear()
    be.render(0, frames * scene.width * scene.height)
    assert np.abs(be.fb_read() - whole).max() <= 2e-4 * whole.max()
    be.set_pixels(False)
    be.fb_clear()
    n = frames * scene.width * scene.height
    be.render(0, n)
    ref = be.fb_read()
    be.close()
    monkeypatch.setenv("CORONA_MI_GROUP_REDUCE", "peer")
    two = pkg.Group(scene, [0, 0])
    two.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    two.render(0, n)
    got = two.fb_read()
    assert np.abs(got - ref).max() <= 2e-4 * ref.max() and two.counters()[4] == n
    two.close()


def test_command_line_renderer_with_hero_wavelengths(tmp_path):
    """corona-mi --wavelengths 4 writes the image of the library's hero render of the same path indices (pfmdiff-mi, like the reference's regression
    scripts)"""
    import shutil
    import subprocess
    from helpers import REPO
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    cli = REPO / "corona-13_amd" / "host" / "corona-mi"
    scene_file = tmp_path / "scenes" / "0010_pt" / "test.nra2"
    out = subprocess.run([str(cli), str(scene_file), "-s", "16", "--batch", "16", "-w", "256", "-h", "256", "--max-verts", "8", "--sampler", "ptdl", "-x", "_hero",
                          "--wavelengths", "4"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    scene = make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    be = pkg.Backend(scene, counters=False)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.render(0, 16 * scene.width * scene.height)
    img = be.fb_read() * scene.gain(16)
    be.close()
    with open(tmp_path / "lib.pfm", "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, dtype=np.float32).tobytes())
    d = subprocess.run([str(REPO / "corona-13_amd" / "host" / "pfmdiff-mi"), str(tmp_path / "scenes" / "0010_pt" / "test_hero_fb00.pfm"), str(tmp_path / "lib.pfm")],
                       capture_output=True, text=True)
    if float(d.stdout.split("rmse:")[1]) >= 1e-3:
        # diagnostics (round 5: this comparison failed on some GPU boxes of one afternoon with rmse 0.6-1.4 and could not be reproduced afterwards, alone or in
        # the full file): which of the two images is off -- both are rendered again -- and where
        def _rp(path):
            with open(path, "rb") as f:
                f.readline(); w, h = (int(x) for x in f.readline().split()); f.readline(); raw = f.read()
            return np.frombuffer(raw[-12 * w * h:], dtype="<f4").reshape(h, w, 3)
        icli = _rp(tmp_path / "scenes" / "0010_pt" / "test_hero_fb00.pfm")
        subprocess.run([str(cli), str(scene_file), "-s", "16", "--batch", "16", "-w", "256", "-h", "256", "--max-verts", "8", "--sampler", "ptdl", "-x", "_hero2", "--wavelengths", "4"], capture_output=True, text=True)
        icli2 = _rp(tmp_path / "scenes" / "0010_pt" / "test_hero2_fb00.pfm")
        be = pkg.Backend(scene, counters=False); be.set_wavelengths(4); be.render(0, 16 * scene.width * scene.height); img2 = be.fb_read() * scene.gain(16); be.close()
        dd = np.abs(icli - img).max(axis=2)
        worst = np.argsort(-dd.ravel())[:5]
        print("\nDBGFAIL sums cli", icli.sum(), "cli2", icli2.sum(), "lib", img.sum(), "lib2", img2.sum(), "rmse cli-cli2", float(np.sqrt(((icli-icli2)**2).sum()/65536)), "lib-lib2", float(np.sqrt(((img-img2)**2).sum()/65536)),
              "cli-lib2", float(np.sqrt(((icli-img2)**2).sum()/65536)), "pixels differing >1e-3:", int((dd > 1e-3).sum()), "worst", [(int(w // 256), int(w % 256), float(dd.ravel()[w]), icli.reshape(-1, 3)[w].tolist(), img.reshape(-1, 3)[w].tolist()) for w in worst])
    assert d.returncode == 0 and float(d.stdout.split("rmse:")[1]) < 1e-3, d.stdout + d.stderr
    fog = subprocess.run([str(cli), str(tmp_path / "scenes" / "0056_fog" / "test.nra2"), "-s", "1", "-w", "64", "-h", "64", "--wavelengths", "4"], capture_output=True, text=True)
    assert fog.returncode == 0, fog.stdout + fog.stderr          # the extended kernels carry four wavelengths too
    assert subprocess.run([str(cli), str(scene_file), "--wavelengths", "3"], capture_output=True, text=True).returncode == 1


def test_hero_soak_half_a_million_paths():
    """500 000 ptdl paths against the oracle, chunk by chunk: primitive sequence, vertex count and splat count of every path and the four
    components of every vertex' throughput; the scalar kernels' bound (2 in 100 000 paths may differ)"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    be = pkg.Backend(scene)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    total, chunk, bad, bad_splats = 500000, 125000, 0, 0
    worst = 0.0
    for first in range(999, 999 + total, chunk):
        g, ge = be.trace_paths_hero(first, chunk)
        o, oe = oracle_hero_records(scene, first, chunk)
        k = np.arange(8)[None, :]
        valid = k < np.minimum(o["length"], 8)[:, None]
        ok = (g["length"] == o["length"]) & ((g["v"]["prim"] == o["v"]["prim"]) | ~valid).all(axis=1)
        bad += int((~ok).sum())
        bad_splats += int((ok & (g["num_splats"] != o["num_splats"])).sum())
        m = ok[:, None] & valid & (k >= 1)
        worst = max(worst, float(np.quantile(rel(ge["throughput"][m], oe["throughput"][m]), 0.9999)))
    be.close()
    assert bad <= 2e-5 * total, bad
    assert bad_splats <= 2e-4 * total, bad_splats          # a grazing shadow ray or a weight at the underflow limit flips one connection (test_gpu_parity.py)
    assert worst < 1e-3, worst


@pytest.mark.parametrize("scene_dir", ["0055_media", "0056_fog", "0057_nested", "0058_cam_mb", "0059_mb", "0066_smooth"])
def test_hero_image_mean_against_the_mf4_reference_render(scene_dir):
    """End to end against the reference built with -DMF_COUNT=4 itself: its own CPU renders of the extended scenes (pt, 256 x 256, 128 spp, three frames:
    tests/golden/mf4_scene_means.json, written by tests/golden/measure_mf4_scenes.py in the build container) and the device's hero render of the same
    scene have the same mean image -- media (free flight from the hero's medium), fog, nested media, moving camera, moving geometry, smooth glass (one
    component survives a specular transmission)."""
    import json
    from helpers import REPO
    with open(GOLDEN / "mf4_scene_means.json") as f:
        measured = json.load(f)
    ref = measured["scenes"][scene_dir]
    scene = make_scene(REPO / "scenes" / scene_dir / "test.nra2", width=measured["size"], height=measured["size"], max_verts=8, sampler=pkg.MI_SAMPLER_PT)
    spp = 4 * measured["spp"]
    be = pkg.Backend(scene, counters=False)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.render(0, spp * scene.width * scene.height)
    mean = (be.fb_read() * scene.gain(spp)).reshape(-1, 3).mean(axis=0)
    be.close()
    rmean, sd = np.array(ref["mean"]), np.array(ref["sd_between_frames"])
    # the reference's mean of three frames has sd / sqrt(3); the device's 512 spp a quarter of a frame's variance; 4 sigma of the difference, at least 1 %
    tol = np.maximum(4.0 * sd * np.sqrt(1.0 / 3.0 + 0.25), 0.01 * rmean)
    assert (np.abs(mean - rmean) <= tol).all(), (mean, rmean, tol)


def test_hero_on_a_device_built_tree_and_a_tree_beyond_lds():
    """the traversal is the scalar kernels': hero paths do not depend on where the tree comes from (built on the device: same hits, so the same
    paths as on the host-built tree) nor on where it lives (262 144-quad backdrop: the top of the tree in LDS, the rest from HBM) -- against the oracle"""
    from helpers import SCENE_LARGE
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    n = 20000
    host = pkg.Backend(scene)
    host.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    a, ae = host.trace_paths_hero(0, n)
    host.close()
    dev = pkg.Backend(scene, device_build=True)
    dev.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    b, be_ = dev.trace_paths_hero(0, n)
    dev.close()
    same = (a["length"] == b["length"]) & (a["num_splats"] == b["num_splats"])
    assert (~same).sum() <= 2
    k = np.arange(8)[None, :]
    valid = k < np.minimum(a["length"], 8)[:, None]
    assert (((a["v"]["prim"] != b["v"]["prim"]) & valid).any(axis=1) & same).sum() <= 2
    assert np.array_equal(ae["throughput"][same], be_["throughput"][same], equal_nan=True) or (rel(ae["throughput"][same], be_["throughput"][same]) > 1e-5).mean() < 1e-4
    big = make_scene(SCENE_LARGE, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PT)
    be = pkg.Backend(big)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    assert 0 < be.lds_nodes() < big.desc.num_nodes
    g, ge = be.trace_paths_hero(100, 8000)
    be.close()
    o, oe = oracle_hero_records(big, 100, 8000)
    ok = g["length"] == o["length"]
    assert (~ok).sum() <= 2
    valid = k < np.minimum(o["length"], 8)[:, None]
    assert (((g["v"]["prim"] != o["v"]["prim"]) & valid).any(axis=1) & ok).sum() <= 2
    m = ok[:, None] & valid & (k >= 1)
    assert np.quantile(rel(ge["throughput"][m], oe["throughput"][m]), 0.999) < 1e-3


def test_hero_entry_points_at_the_c_abi():
    """the two entry points as a C host calls them: null arguments and a wrong count are errors with a message, the extension block is optional, an empty
    range is no work, and the records of a call without the extension block are the records of a call with it"""
    import ctypes as C
    m = pkg.mi_lib()
    assert m.mi_scene_set_wavelengths(None, 4) < 0 and m.mi_last_error()
    assert m.mi_trace_paths_hero(None, 0, 1, None, None) < 0
    scene = make_scene(SCENE_0010, width=256, height=256, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    be = pkg.Backend(scene)
    assert m.mi_scene_set_wavelengths(be._ptr, 0) < 0 and m.mi_scene_set_wavelengths(be._ptr, 8) < 0 and b"1 or 4" in m.mi_last_error()
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    rec = np.zeros(3000, dtype=pkg.record_dtype())
    assert m.mi_trace_paths_hero(be._ptr, 0, 3000, None, None) < 0                  # records are not optional
    assert m.mi_trace_paths_hero(be._ptr, 77, 3000, rec.ctypes.data, None) == 0     # the extension block is
    assert m.mi_trace_paths_hero(be._ptr, 77, 0, rec.ctypes.data, None) == 0
    both, ext = be.trace_paths_hero(77, 3000)
    assert rec.tobytes() == both.tobytes() and np.array_equal(ext["lambda"][:, 0], rec["lambda"])
    be.close()


def test_hero_ignores_the_traversal_switch():
    """the HERO kernels have the exact rounds only: a scene switched to the FAST rounds renders its hero paths with them all the same (and says so in the
    kernel's name), its scalar paths with the FAST rounds"""
    scene = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=pkg.MI_SAMPLER_PT)
    a = pkg.Backend(scene, traversal="exact")
    a.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    ra, ea = a.trace_paths_hero(0, 5000)
    a.close()
    b = pkg.Backend(scene, traversal="fast", counters=False)
    assert b.kernel_name().split(", ")[7] == "true"           # scalar: the FAST instantiation
    b.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    assert b.kernel_name().split(", ")[7] == "false" and b.kernel_name().endswith("true>")
    rb, eb = b.trace_paths_hero(0, 5000)
    b.close()
    assert ra.tobytes() == rb.tobytes() and ea.tobytes() == eb.tobytes()


def test_hero_pixels_from_path_indices():
    """MI_PIXELS_FROM_INDEX with four wavelengths per path: path i starts inside pixel (i mod W H) with its generator seeded through the hash of its index
    (the mode tile-owned sharding renders in) -- path for path against the oracle's hero lanes in that mode"""
    from helpers import oracle_pixels
    scene = make_scene(SCENE_0010, width=640, height=352, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    W, H = scene.width, scene.height
    be = pkg.Backend(scene)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    be.set_pixels(True)
    first, n = 2 * W * H - 5000, 20000
    gpu, gext = be.trace_paths_hero(first, n)
    be.close()
    with oracle_pixels():
        ora, oext = oracle_hero_records(scene, first, n)
    idx = (first + np.arange(n)) % (W * H)
    # inside the pixel its index names (pixel + a number in [0, 1): in float the sum may come out as the next pixel's edge, one path in 20 000 here)
    di, dj = gpu["pixel_i"] - (idx % W), gpu["pixel_j"] - (idx // W)
    assert (di >= 0).all() and (di <= 1).all() and (dj >= 0).all() and (dj <= 1).all() and (di == 1).sum() + (dj == 1).sum() <= 3
    assert np.abs(gpu["pixel_i"] - ora["pixel_i"]).max() <= 1e-4 and np.abs(gpu["pixel_j"] - ora["pixel_j"]).max() <= 1e-4 and np.abs(gext["lambda"] - oext["lambda"]).max() <= 1e-4
    same = gpu["length"] == ora["length"]
    assert (~same).sum() <= 2 and (gpu["num_splats"] != ora["num_splats"]).sum() <= 8
    for k in range(1, 8):
        m = same & (ora["length"] > k)
        if m.sum():
            assert (gpu["v"]["prim"][m, k] != ora["v"]["prim"][m, k]).sum() <= 2
            assert (rel(gext["throughput"][m, k], oext["throughput"][m, k]).max(axis=1) >= 1e-3).sum() <= max(3, int(1e-3 * m.sum()))


def test_hero_other_frame_seed_large_indices_and_edge_cases():
    """another frame number, path indices beyond 2^32, a range of one path, a range that does not fill a workgroup, max depth 2 (no bounce) and 32"""
    scene = make_scene(SCENE_0010, width=640, height=352, max_verts=8, frame=7, sampler=pkg.MI_SAMPLER_PTDL)
    be = pkg.Backend(scene)
    be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
    for first, n in ((0, 4000), ((1 << 33) + 12345, 4000), (5, 1), (99, 777)):
        gpu, gext = be.trace_paths_hero(first, n)
        ora, oext = oracle_hero_records(scene, first, n)
        assert np.array_equal(gpu["index"], ora["index"]) and len(gpu) == n
        assert np.abs(gpu["pixel_i"] - ora["pixel_i"]).max() <= 1e-5 and np.abs(gext["lambda"] - oext["lambda"]).max() <= 1e-4
        assert (gpu["length"] != ora["length"]).sum() <= max(1, int(1e-3 * n))
        m = (gpu["length"] == ora["length"]) & (ora["length"] > 2)
        if m.sum():
            assert (gpu["v"]["prim"][m, 2] != ora["v"]["prim"][m, 2]).sum() <= max(1, int(1e-3 * n))
    be.close()
    for mv in (2, 32):
        s2 = make_scene(SCENE_0010, width=640, height=352, max_verts=mv, sampler=pkg.MI_SAMPLER_PTDL)
        b2 = pkg.Backend(s2)
        b2.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
        gpu, gext = b2.trace_paths_hero(1000, 6000)
        b2.close()
        ora, oext = oracle_hero_records(s2, 1000, 6000)
        assert gpu["length"].max() <= mv and (gpu["length"] != ora["length"]).sum() <= 6 and (gpu["num_splats"] != ora["num_splats"]).sum() <= 12


def test_hero_exchange_between_waves_changes_no_path(monkeypatch):
    """The pools of the exchange between waves carry components 1..3 of a hero path in eight more words per vertex (csrc/mi_regroup.h): which lane finishes a
    path must not matter to the path. Records and extension blocks with the exchange (the default) and without it (CORONA_MI_REGROUP=0) are the same BYTES --
    plain pt / ptdl, the extended kernels (volume vertices as a class), Halton, a tree in HBM, launches smaller than a wave and than a workgroup; rendered
    frames agree to the order of the float atomics and count every path."""
    from helpers import SCENE_CAM_MB
    cases = [(SCENE_0010, pkg.MI_SAMPLER_PT, "rand", 8, 300000), (SCENE_0010, pkg.MI_SAMPLER_PTDL, "rand", 8, 200000), (SCENE_METAL, pkg.MI_SAMPLER_PTDL, "rand", 8, 150000),
             (SCENE_MEDIA, pkg.MI_SAMPLER_PTDL, "rand", 32, 100000), (SCENE_CAM_MB, pkg.MI_SAMPLER_PT, "rand", 8, 100000),
             (SCENE_0010, pkg.MI_SAMPLER_PTDL, "halton", 8, 100000), (SCENE_FINE, pkg.MI_SAMPLER_PT, "rand", 8, 100000)]
    for path, sampler, points, mv, n in cases:
        scene = make_scene(path, width=1280, height=720, max_verts=mv, sampler=sampler, pointsampler=pkg.MI_POINTS_HALTON if points == "halton" else pkg.MI_POINTS_RAND)
        monkeypatch.setenv("CORONA_MI_REGROUP", "0")
        off = pkg.Backend(scene)
        monkeypatch.delenv("CORONA_MI_REGROUP")
        on = pkg.Backend(scene)
        off.set_wavelengths(pkg.MI_WAVELENGTHS_HERO); on.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
        for first, count in ((5, n), (123456789, 1), (77, 63), (1000, 1000), (2 ** 33 + 9, 50000)):
            (a, ae), (b, be_) = off.trace_paths_hero(first, count), on.trace_paths_hero(first, count)
            assert a.tobytes() == b.tobytes() and ae.tobytes() == be_.tobytes(), (str(path), sampler, points, first, count)
        off.close(); on.close()
        per = 2 * scene.width * scene.height
        frames = []
        for env in ("0", None):
            if env is None:
                monkeypatch.delenv("CORONA_MI_REGROUP", raising=False)
            else:
                monkeypatch.setenv("CORONA_MI_REGROUP", env)
            be = pkg.Backend(scene, counters=False)
            be.set_wavelengths(pkg.MI_WAVELENGTHS_HERO)
            c0 = be.counters()
            be.render(9 * per, per)
            frames.append(be.fb_read())
            assert be.counters()[4] - c0[4] == per
            be.close()
        monkeypatch.delenv("CORONA_MI_REGROUP", raising=False)
        assert np.abs(frames[0] - frames[1]).max() <= 2e-4 * np.abs(frames[0]).max(), (str(path), sampler)
