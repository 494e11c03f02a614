"""Pin the oracle (oracle/liboracle.so, our CPU restatement) against the REAL reference.

tests/golden/paths_*.npz were dumped from hanatos/corona-13 itself (oracle/_ref, built from
/root/reference by oracle/Makefile, run with MOD_points=xorshift128p -t 1 => bit-reproducible) by
tests/golden/make_golden.py. The oracle traces the same path indices with the same random numbers.

Floating point: the reference is compiled -O3 -ffast-math (reassociation, FMA, rsqrtss in
rgb2spec_eval_fast), so agreement is not bitwise. Stated tolerances:
  * camera sample (pixel, wavelength, time)      exact to 1e-4 absolute
  * >= 99.8 % of paths: identical vertex count and identical primitive per vertex
  * >= 99.5 % of paths: identical number of splats (ptdl: shadow rays grazing the emitter may flip)
  * median relative deviation of splat values < 5e-4, 99th percentile < 5e-2 (GGX roughness 0.04 on
    a sphere 17 dm away amplifies the 1e-4 hit-point noise of the float quadratic)
  * total splatted energy of the set within 1.5e-3 relative (pt) / 1e-2 (ptdl)
"""
import json

import numpy as np
import pytest

from helpers import GOLDEN, SCENE_0010, SCENE_ALL, SCENE_CAM_MB, SCENE_FINE, SCENE_FOG, SCENE_MB, SCENE_MB_LIGHT, SCENE_MB_ROUND, SCENE_MB_ROUND_LIGHT, SCENE_MEDIA, SCENE_NESTED, SCENE_METAL, SCENE_ROUGH, SCENE_SMOOTH, load_pkg, make_scene, oracle_lib, oracle_records, oracle_render

pkg = load_pkg()

CASES = [
    ("pt_mv8", pkg.MI_SAMPLER_PT, SCENE_0010, 1.5e-3),
    ("ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_0010, 1e-2),
    ("pt_mv4_256", pkg.MI_SAMPLER_PT, SCENE_0010, 1.5e-3),       # BASELINE config 1
    ("rough_mv32", pkg.MI_SAMPLER_PT, SCENE_ROUGH, 1.5e-3),      # BASELINE config 4 (0052 parameters)
    # smooth glass (roughness 0 <= GLOSSY_THR): the specular branches of dielectric.c:303-343 (sample), :425-440 (brdf: zero), pdf
    ("smooth_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_SMOOTH, 1.5e-3),
    ("smooth_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_SMOOTH, 1e-2),
    ("fine_mv8", pkg.MI_SAMPLER_PT, SCENE_FINE, 1.5e-3),         # 16 384-quad backdrop: QBVH of 1711 nodes (host builder vs reference builder)
    ("metal_mv8", pkg.MI_SAMPLER_PT, SCENE_METAL, 1.5e-3),       # row a19: metal.c sample
    ("metal_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_METAL, 1e-2),  # row a19: metal.c brdf / pdf through next event estimation
    # SURVEY 8(f) row 2, MOD_pointsampler=halton ("halton_" cases run with MI_POINTS_HALTON): dimension bookkeeping
    # (rand_beg / rand_cnt incl. the four next-event dimensions path_pop hands to the previous vertex), permutation tables
    # from srand48(frame), and -- max depth 32 with ptdl -- the fall-back to the per-path generator from dimension 256 on
    # SURVEY 8(f) row 3, homogeneous medium inside the glass sphere (`interior`, `medium_rgb`, `color v`): free-flight sampling,
    # volume vertices, Henyey-Greenstein, transmittance on next-event connections, volume pdfs in the MIS weights. The energy
    # tolerance is wider: mu_t(lambda) is a sigmoid evaluated with rsqrtss by the reference (include/rgb2spec.h:145-149) and
    # above 800 nm, where the sigmoid of this colour is small, that approximation is worth several per cent in exp(-mu_t d)
    ("media_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_MEDIA, 5e-3),
    ("media_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_MEDIA, 1e-2),
    ("media_pt_mv32", pkg.MI_SAMPLER_PT, SCENE_MEDIA, 5e-3),
    # thin global fog (`exterior <medium> 0`, scenes/0056_fog): nearly every path scatters in the open, so next event estimation
    # from volume vertices reaches the emitters. mu_t is a SMALL sigmoid value here, where the reference's rsqrtss is worth
    # 0.1-0.4 %: these two cases run the oracle with its reference-build emulation of that instruction (see reference_rsqrt)
    ("fog_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_FOG, 1.5e-3),
    ("fog_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_FOG, 1e-2),
    # fog outside, a scattering medium in the sphere, a purely absorbing one (albedo 0: src/shader.c:99-100) in the cone:
    # every transition of the nested-medium stack between exterior and interior volumes
    ("nested_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_NESTED, 1.5e-3),
    ("nested_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_NESTED, 1e-2),
    # camera motion blur (cf. regression/0003_cam_mb): per-path camera frame, view_cam_init_frame src/view.c:903-919
    ("cam_mb_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_CAM_MB, 1.5e-3),
    ("cam_mb_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_CAM_MB, 1e-2),
    # motion-blurred geometry (cf. regression/0002_mb): vertices and vertex normals interpolated at the path's time
    ("mb_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_MB, 1.5e-3),
    ("mb_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_MB, 1e-2),
    ("mb_round_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_MB_ROUND, 1.5e-3),          # moving sphere, cone and cylinder (centre / end points at the ray's time)
    ("mb_round_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_MB_ROUND, 1e-2),
    ("mb_light_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_MB_LIGHT, 1.5e-3),          # the emitter moves too: prims_sample / prims_retime at the path's time
    ("mb_light_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_MB_LIGHT, 1e-2),
    # moving sphere and cone AS EMITTERS (round 4; src/prims.c:216-252 prims_sample -> geo_sphere_retime / geo_line_retime at the path's time)
    ("mb_round_light_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_MB_ROUND_LIGHT, 1.5e-3),
    ("mb_round_light_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_MB_ROUND_LIGHT, 1e-2),
    ("halton_pt_mv8", pkg.MI_SAMPLER_PT, SCENE_0010, 1.5e-3),
    ("halton_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_0010, 1e-2),
    ("halton_ptdl_rough_mv32", pkg.MI_SAMPLER_PTDL, SCENE_ROUGH, 1e-2),
    # everything at once (scenes/0061_all): fog + interior media, moving camera, moving geometry and emitter -- with Halton ptdl, and pt at depth 32
    ("halton_all_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_ALL, 1e-2),
    ("all_pt_mv32", pkg.MI_SAMPLER_PT, SCENE_ALL, 5e-3),
    ("halton_fog_ptdl_mv8", pkg.MI_SAMPLER_PTDL, SCENE_FOG, 1e-2),          # the free-flight dimension (rand_beg + 0) from the Halton sampler
]

# Fraction of paths that must have the reference's vertex count. The reference's metal Fresnel term (metal.c:79-157)
# takes sqrt(0.5*(len - cost2r)) with len = |cost2|; for gold above ~690 nm len and cost2r agree to the last float bit, and
# under the reference build's -ffast-math (reciprocal-sqrt refinement) the difference comes out one ulp negative in a
# few percent of the samples: NaN -> clamp -> R = 0, a black sample that ends the path. The oracle evaluates the same
# formula with IEEE sqrt and keeps those samples; every such mismatch is a path the REFERENCE ends at a metal vertex
# with lambda > 600 nm (asserted below).
MIN_SAME_LENGTH = {"metal_mv8": 0.99, "metal_ptdl_mv8": 0.99,
                   # with the emulation of the reference build's NaN (METAL_REFERENCE_CASES below): measured 0.9963 / 0.9953. Not the 0.998 of the
                   # other cases, and it cannot be: the NaN is the SIGN OF A ROUNDING ERROR of cost2r (oracle/oracle_shade.c:638-663), the
                   # reference's cosr and the oracle's differ in the last bit wherever -ffast-math contracted something on the way, so sample by
                   # sample the verdict is a coin that both sides toss -- what can be pinned is that they toss it equally often and in the same
                   # place (test_metal_reference_build_ends_the_references_share_of_paths)
                   "metal_mv8@reference": 0.994, "metal_ptdl_mv8@reference": 0.994}
_ = [
]


class reference_rsqrt:
    """context: the oracle evaluates rgb2spec_eval_fast like the reference build on this host (FMA + hardware rsqrtss,
    oracle/oracle_shade.c). The fixtures were dumped on an Intel Xeon; rsqrtss tables differ between vendors, so the emulation
    only counts as exact when the host returns that CPU's values for a few arguments."""
    KNOWN = {1.5: 1062273024, 2.0: 1060435968, 3.0: 1058260992, 1.0001: 1065349120}

    def __enter__(self):
        import ctypes as C
        o = oracle_lib()
        o.oracle_rsqrtss.restype = C.c_float
        o.oracle_rsqrtss.argtypes = [C.c_float]
        self.exact = bool(o.oracle_set_reference_rsqrt(1)) and all(
            int(np.float32(o.oracle_rsqrtss(x)).view(np.uint32)) == bits for x, bits in self.KNOWN.items())
        return self

    def __exit__(self, *a):
        oracle_lib().oracle_set_reference_rsqrt(0)


def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-20, np.maximum(np.abs(a), np.abs(b)))


# Row a19 pinned to the reference BUILD (round 4): the same two metal fixtures with the oracle's emulation of the reference build's
# NaN (oracle_set_reference_metal: the compiled Fresnel sequence of src/shaders/metal.c:79-157 restated operation by operation,
# oracle/oracle_shade.c) -- at the thresholds every other case meets, not the 0.99 the formula as written needs.
METAL_REFERENCE_CASES = [
    ("metal_mv8@reference", pkg.MI_SAMPLER_PT, SCENE_METAL, 1.5e-3),
    ("metal_ptdl_mv8@reference", pkg.MI_SAMPLER_PTDL, SCENE_METAL, 1e-2),
]


class reference_metal:
    def __enter__(self):
        oracle_lib().oracle_set_reference_metal(1)
        return self

    def __exit__(self, *a):
        oracle_lib().oracle_set_reference_metal(0)


def oracle_for_case(name, sampler, scene_path):
    """(reference records, oracle records) of one case, or None when this host cannot emulate the reference build's rsqrtss"""
    fixture = name.split("@")[0]
    g = np.load(GOLDEN / f"paths_{fixture}.npz")
    ref = g["records"]
    s = make_scene(scene_path, width=int(g["width"]), height=int(g["height"]), max_verts=int(g["max_verts"]), sampler=sampler,
                   pointsampler=pkg.MI_POINTS_HALTON if name.startswith("halton_") else pkg.MI_POINTS_RAND)
    if name.startswith(("fog_", "nested_", "halton_fog_", "halton_all_", "all_")):
        with reference_rsqrt() as emu:
            ora = oracle_records(s, 0, len(ref))
        if not emu.exact:
            return None
    elif name.endswith("@reference"):
        with reference_metal():
            ora = oracle_records(s, 0, len(ref))
    else:
        ora = oracle_records(s, 0, len(ref))
    return ref, ora


def measure(ref, ora):
    """what the oracle achieves against the reference's dump: the quantities the test bounds"""
    m = {}
    same_len = ref["length"] == ora["length"]
    m["same_length"] = float(same_len.mean())
    bad_p = bad_m = 0.0
    for k in range(1, 8):
        sel = same_len & (ref["length"] > k)
        if sel.sum():
            bad_p = max(bad_p, float((ref["v"]["prim"][sel, k] != ora["v"]["prim"][sel, k]).sum()) / float(sel.sum()))
            bad_m = max(bad_m, float((ref["v"]["mode"][sel, k] != ora["v"]["mode"][sel, k]).sum()) / float(sel.sum()))
    m["worst_prim_mismatch"] = bad_p
    m["worst_mode_mismatch"] = bad_m
    same_splats = ref["num_splats"] == ora["num_splats"]
    m["same_splats"] = float(same_splats.mean())
    both = same_len & same_splats
    devs, nan_same = [], []
    for k in range(ref["splat"].shape[1]):
        sel = both & (ref["num_splats"] > k)
        if sel.sum():
            a, b = ref["splat"]["value"][sel, k], ora["splat"]["value"][sel, k]
            nan_same.append(np.isnan(a) == np.isnan(b))
            fin = np.isfinite(a) & np.isfinite(b)
            devs.append(rel(a[fin], b[fin]))
    devs = np.concatenate(devs)
    m["nan_same"] = float(np.concatenate(nan_same).mean())
    m["splat_dev_median"] = float(np.median(devs))
    m["splat_dev_p99"] = float(np.quantile(devs, 0.99))
    e_ref, e_ora = np.nan_to_num(ref["splat"]["col"][both]).sum(axis=(0, 1)), np.nan_to_num(ora["splat"]["col"][both]).sum(axis=(0, 1))
    m["energy_dev"] = float(np.abs(e_ref - e_ora).max() / np.abs(e_ref).max())
    return m


def measure_case(name, sampler, scene_path):
    r = oracle_for_case(name, sampler, scene_path)
    return None if r is None else measure(*r)


with open(GOLDEN / "oracle_vs_reference_measured.json") as _f:
    MEASURED = json.load(_f)        # written by tests/golden/measure_oracle_vs_reference.py (committed next to it)


def bound_fraction(measured, floor):
    """a fraction of agreeing paths may drop by a quarter of what the measurement misses (other libm, other compiler), never below `floor`"""
    return max(floor, 1.0 - 1.25 * (1.0 - measured) - 2e-4)


def bound_dev(measured, ceiling, slack=1.5):
    return min(ceiling, slack * measured + 1e-7)


@pytest.mark.parametrize("name,sampler,scene_path,etol", CASES + METAL_REFERENCE_CASES)
def test_oracle_matches_reference_paths(name, sampler, scene_path, etol):
    r = oracle_for_case(name, sampler, scene_path)
    if r is None:
        pytest.skip("rsqrtss of this host differs from the CPU the fixtures were dumped on; thin media need it (see CASES)")
    ref, ora = r
    for f, tol in (("pixel_i", 1e-4), ("pixel_j", 1e-4), ("lambda", 1e-4), ("time", 1e-6), ("scramble", 1e-6)):
        assert np.abs(ref[f] - ora[f]).max() <= tol, f
    got, was = measure(ref, ora), MEASURED[name]
    same_len = ref["length"] == ora["length"]
    # floors: what every case must reach whatever was measured (the module's stated tolerances); bounds: the measurement plus a margin
    assert got["same_length"] >= bound_fraction(was["same_length"], MIN_SAME_LENGTH.get(name, 0.998)), (got["same_length"], was["same_length"])
    if name in MIN_SAME_LENGTH and not name.endswith("@reference"):
        bad = np.where(~same_len)[0]
        shorter = ref["length"][bad] < ora["length"][bad]
        last = np.minimum(ref["length"][bad] - 1, 7)
        at_metal = ref["v"]["shader"][bad, last] == 10
        assert (shorter & at_metal & (ref["lambda"][bad] > 600)).mean() >= 0.8
    for k in range(1, 8):
        m = same_len & (ref["length"] > k)
        if m.sum():
            # rough-metal bounce chains: a few hundred paths per depth, one neighbouring backdrop quad is 0.3 %
            # at least the stated fraction, but one stray path is allowed at depths only a few hundred of the fixture's paths reach
            bad_p, bad_m = (ref["v"]["prim"][m, k] != ora["v"]["prim"][m, k]).sum(), (ref["v"]["mode"][m, k] != ora["v"]["mode"][m, k]).sum()
            metal = name.split("@")[0] in ("metal_mv8", "metal_ptdl_mv8")
            assert bad_p <= max(1, min(0.005 if metal else 0.001, 1.5 * was["worst_prim_mismatch"] + 2e-4) * m.sum())
            assert bad_m <= max(1, min(0.005 if metal else 0.002, 1.5 * was["worst_mode_mismatch"] + 2e-4) * m.sum())
    assert got["same_splats"] >= bound_fraction(was["same_splats"], 0.995), (got["same_splats"], was["same_splats"])
    same_splats = ref["num_splats"] == ora["num_splats"]
    both = same_len & same_splats
    for k in range(ref["splat"].shape[1]):
        m = both & (ref["num_splats"] > k)
        if m.sum():
            assert (ref["splat"]["length"][m, k] == ora["splat"]["length"][m, k]).all()
    # the reference's own weights are NaN for a few connections (inf/inf in its MIS products; view_splat drops them)
    assert got["nan_same"] >= 0.999
    assert got["splat_dev_median"] < bound_dev(was["splat_dev_median"], 5e-4), (got["splat_dev_median"], was["splat_dev_median"])
    assert got["splat_dev_p99"] < bound_dev(was["splat_dev_p99"], 5e-2), (got["splat_dev_p99"], was["splat_dev_p99"])
    assert got["energy_dev"] <= min(etol, 2.0 * was["energy_dev"] + 1e-4), (got["energy_dev"], was["energy_dev"])


@pytest.mark.parametrize("name,sampler", [("metal_mv8", pkg.MI_SAMPLER_PT), ("metal_ptdl_mv8", pkg.MI_SAMPLER_PTDL)])
def test_metal_reference_build_ends_the_references_share_of_paths(name, sampler):
    """Row a19 against the reference BUILD. Its metal sample() ends 2-4 % of the samples at a gold vertex with a NaN (src/shaders/metal.c:
    79-157 as gcc -O3 -ffast-math compiles it); `oracle_set_reference_metal(1)` restates that instruction sequence. The verdict on one
    sample is the sign of a rounding error, so it cannot agree path for path (see MIN_SAME_LENGTH) -- pinned here: with the formula as
    written the oracle ends clearly FEWER paths at a metal vertex than the reference's dump does and nearly every mismatching path is
    one the reference cut short; with the emulation the two end equally many there (within three standard deviations of the
    number of such verdicts), the mismatches go both ways, and fewer paths mismatch than before."""
    ref, plain = oracle_for_case(name, sampler, SCENE_METAL)
    _, emu = oracle_for_case(name + "@reference", sampler, SCENE_METAL)

    def ends_at_metal(r):
        last = np.minimum(r["length"] - 1, 7)
        return int(((r["v"]["shader"][np.arange(len(r)), last] == 10) & (r["length"] < 8)).sum())

    n_ref, n_plain, n_emu = ends_at_metal(ref), ends_at_metal(plain), ends_at_metal(emu)
    killed = n_ref - n_plain                                    # the paths the reference build's NaN ended in this fixture
    assert killed >= 8, (n_ref, n_plain)
    assert abs(n_emu - n_ref) <= 3.0 * np.sqrt(2.0 * killed), (n_ref, n_emu, killed)
    d_plain, d_emu = ref["length"] != plain["length"], ref["length"] != emu["length"]
    assert (ref["length"][d_plain] < plain["length"][d_plain]).mean() >= 0.85          # one-sided: the reference is the shorter one
    shorter = (ref["length"][d_emu] < emu["length"][d_emu]).mean()
    assert 0.2 <= shorter <= 0.8, shorter                                                # both ways now
    assert d_emu.sum() <= d_plain.sum()


def test_first_vertex_geometry_close():
    g = np.load(GOLDEN / "paths_pt_mv8.npz")
    ref = g["records"]
    s = make_scene(SCENE_0010, width=1280, height=720, max_verts=8)
    ora = oracle_records(s, 0, len(ref))
    m = (ref["length"] > 1) & (ora["length"] > 1)
    dx = np.abs(ref["v"]["x"][m, 1] - ora["v"]["x"][m, 1]).max(axis=1)
    assert np.quantile(dx, 0.999) < 1e-3 and dx.max() < 5e-3       # dm; sphere hits 17 dm away carry ~1e-4 noise
    assert rel(ref["v"]["throughput"][m, 1], ora["v"]["throughput"][m, 1]).max() < 1e-5


def test_rng_known_answers():
    """xorshift128+ seeded per path index (src/points.d/xorshift128p.c:53-74): the camera sample of the golden
    records is a direct function of draws 6..9, so the generator is pinned by them."""
    g = np.load(GOLDEN / "paths_pt_mv8.npz")
    ref = g["records"][:64]
    o = oracle_lib()
    for r in ref:
        seq = np.zeros(9, dtype=np.float32)
        o.oracle_rand_sequence(int(r["index"]), 1, 9, seq.ctypes.data)
        assert abs(0.1 + seq[0] * 0.8 - r["scramble"]) < 1e-6
        assert abs(360 + 470 * seq[1] - r["lambda"]) < 1e-4
        assert abs(min(seq[5] * 1280, 1280 - 1e-4) - r["pixel_i"]) < 1e-4
        assert abs(min(seq[6] * 736, 736 - 1e-4) - r["pixel_j"]) < 1e-4
    assert ((0 <= seq) & (seq < 1)).all()


def test_traversal_work_counters_match_reference():
    """-DACCEL_DEBUG totals of the reference for path indices [0, 1280*736) (tests/golden/counters.json)
    vs the oracle over the same indices: rays, node visits, box hits, prim tests within 0.2 %."""
    gold = json.loads((GOLDEN / "counters.json").read_text())
    for name, sampler in (("pt_mv8", pkg.MI_SAMPLER_PT), ("ptdl_mv8", pkg.MI_SAMPLER_PTDL)):
        s = make_scene(SCENE_0010, width=1280, height=720, max_verts=8, sampler=sampler)
        n = gold[name]["paths"]
        _, cnt, _ = oracle_render(s, 0, n, threads=8)
        assert cnt[4] == n
        for k, key in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
            assert abs(cnt[k] - gold[name][key]) <= 2e-3 * gold[name][key], (name, key, cnt[k], gold[name][key])


def _tile_stats(fb, gain):
    H, W, _ = fb.shape
    return (fb * gain).reshape(H // 32, 32, W // 32, 32, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("name,sampler,scene_path,mv,w,h,spp", [
    ("pt_mv4_256", pkg.MI_SAMPLER_PT, SCENE_0010, 4, 256, 256, 64),
])
@pytest.mark.parametrize("pixels", ["sampled", "from path indices"])
def test_oracle_image_statistics_vs_reference_render(name, sampler, scene_path, mv, w, h, spp, pixels):
    """Statistical oracle: high-spp render of the real reference (sfmt, all cores) reduced to 32x32 tile means.
    The oracle's `spp` render must agree in the image mean within 3 standard errors estimated from the
    tile-to-tile scatter (pure pt with a tiny emitter is firefly dominated, hence the robust statistic)."""
    fn = GOLDEN / f"tilemeans_{name}.npz"
    if not fn.exists():
        pytest.skip("tile-mean fixture not generated")
    g = np.load(fn)
    s = make_scene(scene_path, width=w, height=h, max_verts=mv, sampler=sampler)
    if pixels == "sampled":
        fb, _, _ = oracle_render(s, 0, spp * s.width * s.height, threads=8)
    else:
        # the tiled branch of render_sample_path (gi.c:88-95, `#if 0` in the reference's default build: no path dumps exist for it) is pinned
        # against the same converged reference render: two tile owners, every pixel sampled spp times
        from helpers import oracle_render_tiles
        from helpers import oracle_pixels
        with oracle_pixels():
            fb = sum(oracle_render_tiles(s, 0, spp, g, 2, threads=8)[0] for g in range(2))
    tiles = _tile_stats(fb, s.gain(spp))
    ref = g["tiles"]
    # compare medians of tile luminance (robust against fireflies)
    assert abs(np.median(tiles[..., 1]) - np.median(ref[..., 1])) < 0.1 * np.median(ref[..., 1])
    # and the plain mean within a generous factor given the noise floor at this spp
    assert abs(tiles[..., 1].mean() - ref[..., 1].mean()) < 0.35 * ref[..., 1].mean()


def test_halton_falls_back_to_the_generator_from_dimension_256():
    """ptdl at max depth 32 owns 9 dimensions per vertex (5 extension + 4 next event), so vertex 29 onwards asks for
    dimensions >= 256 and gets the per-path generator instead (src/pointsampler.d/halton.c:78-80). The fixture holds the
    198 paths of 29+ vertices among the reference's first 2 000 000 (0052 rough-dielectric scene)."""
    g = np.load(GOLDEN / "paths_halton_long_mv32.npz")
    ref = g["records"]
    assert len(ref) > 100 and ref["length"].min() >= 29
    s = make_scene(SCENE_ROUGH, width=1280, height=720, max_verts=32, sampler=pkg.MI_SAMPLER_PTDL, pointsampler=pkg.MI_POINTS_HALTON)
    ora = np.concatenate([oracle_records(s, int(i), 1) for i in ref["index"]])
    # chains of 29+ rough-dielectric bounces amplify the reference's fast-math noise: most, not all, stay on the same path
    same = ref["length"] == ora["length"]
    assert same.mean() >= 0.97, same.mean()
    assert (ref["num_splats"][same] == ora["num_splats"][same]).mean() >= 0.95
    m = same & (ref["num_splats"] == ora["num_splats"]) & (ref["num_splats"] > 0)
    assert m.sum() > 100 and (ref["splat"]["length"][m] == ora["splat"]["length"][m]).all()
    assert np.median(rel(ref["splat"]["value"][m, 0], ora["splat"]["value"][m, 0])) < 1e-2
    for k in range(8):
        assert (ref["v"]["prim"][same, k] == ora["v"]["prim"][same, k]).mean() >= 0.99


def test_halton_tables_match_libc_drand48():
    """the oracle restates srand48/lrand48 (POSIX LCG) to draw the digit permutations (ext/halton/halton.h:3244-3274);
    rebuild the tables of a few bases with the C library's generator"""
    import ctypes as C
    o = oracle_lib()
    o.oracle_halton_tables.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    o.oracle_halton_tables.restype = C.c_uint32
    base, off = np.zeros(256, dtype=np.uint32), np.zeros(256, dtype=np.uint32)
    n = o.oracle_halton_tables(7, None, base.ctypes.data, off.ctypes.data)
    table = np.zeros(n, dtype=np.uint16)
    o.oracle_halton_tables(7, table.ctypes.data, None, None)
    assert list(base[:6]) == [2, 3, 5, 7, 11, 13] and base[255] == 1619 and n == 193329   # 387 KB of 16-bit entries
    libc = C.CDLL(None)
    libc.lrand48.restype = C.c_long
    libc.srand48(C.c_long(7))
    perms = {}
    for b in range(4, 30):                                   # bases are shuffled in order, prime or not
        perm = list(range(b))
        for i in range(b - 1):
            j = i + libc.lrand48() // ((1 << 31) // (b - i) + 1)
            perm[i], perm[j] = perm[j], perm[i]
        perms[b] = perm
    for dim, (b, digits) in {2: (5, 3), 3: (7, 3), 4: (11, 2), 8: (23, 1), 9: (29, 1)}.items():
        assert base[dim] == b
        size = b ** digits
        want = []
        for i in range(size):
            r, idx = 0, i
            for _ in range(digits):
                r = r * b + perms[b][idx % b]
                idx //= b
            want.append(r)
        assert list(table[off[dim]:off[dim] + size]) == want, b
    # base 3 keeps the identity permutation: five digits reversed
    assert table[off[1] + 1] == 81 and table[off[1] + 3] == 27


def test_motion_blur_work_counters_equal_the_reference():
    """scenes/0059_mb with the node boxes interpolated at every ray's time (mi_scene_desc.nodes_t1; src/accel.d/qbvhmp.c:1208-1224):
    the oracle's rays, node visits, box hits and primitive tests on a quarter of the reference's 1-spp frame scale to the
    reference's own -DACCEL_DEBUG totals (tests/golden/counters.json, corona_pt_xs_mv8_dbg) within 1 % -- with one static box
    around each primitive's whole motion the same frame costs 7 % more node visits, 14 % more box hits, 33 % more primitive tests."""
    import json
    from helpers import SCENE_MB, oracle_render
    gold = json.loads((GOLDEN / "counters.json").read_text())["mb_pt_mv8"]
    scene = make_scene(SCENE_MB, width=1280, height=720, max_verts=8)
    n = scene.width * scene.height // 4
    _, cnt, _ = oracle_render(scene, 0, n, threads=8)
    assert cnt[4] == n
    for k, key in ((0, "rays"), (1, "node_visits"), (2, "box_hits"), (3, "prim_tests")):
        assert abs(4 * cnt[k] / gold[key] - 1) < 1e-2, (key, 4 * cnt[k], gold[key])


def test_pixels_from_indices_why_not_the_reference_branch_literally():
    """render_sample_path's tiled branch (src/render.d/gi.c:88-95) is dead code in the reference's builds; MI_PIXELS_FROM_INDEX uses its hook
    (path_set_pixel) but departs from it in two ways (corona_mi.h). The oracle restates BOTH (oracle_set_pixels_from_index 1 / 2); this is the
    measurement behind the departure, on the reference's own generator:
      - points_set_state seeds path i with 1 + i and ten warm-up rounds: the first numbers of consecutive indices stay correlated;
      - with the branch's row-by-row pixels that puts (nearly) one wavelength on a whole stretch of neighbouring pixels per frame (on the
        GPU, cfg 3: the 256-spp image mean 1.5 % off in X and Z, profiles/r05_tiles.txt);
      - the product's mode (hashed seed, position inside the pixel) has neither."""
    from helpers import oracle_pixels
    s = make_scene(SCENE_0010, width=160, height=96, max_verts=8, sampler=pkg.MI_SAMPLER_PTDL)
    W, H = s.width, s.height
    lam = {}
    for mode in (0, 1, 2):
        with oracle_pixels(mode):
            lam[mode] = oracle_records(s, 0, 4000)["lambda"]
    corr = {m: abs(np.corrcoef(l[:-1], l[1:])[0, 1]) for m, l in lam.items()}
    assert corr[0] > 0.9 and corr[1] > 0.9 and corr[2] < 0.05, corr            # the reference's seeding / the hashed one
    # what that does to a 32-pixel stretch of a tile's row: the literal branch lights it with (nearly) one wavelength per frame -- a spread of a few
    # nanometres where independent samples of [360, 830] nm spread 136 nm
    spread = {m: l[:3968].reshape(-1, 32).std(axis=1).mean() for m, l in lam.items()}
    assert spread[1] < 20.0 and spread[2] > 110.0, spread
