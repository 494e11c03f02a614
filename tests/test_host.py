"""Host side (plain C): scene files, QBVH builder, emitters, camera, rgb2spec, PFM. CPU only."""
import ctypes as C
import json
import sys
import os
import struct
import tempfile

import numpy as np
import pytest

from helpers import GOLDEN, REPO, SCENE_0010, SCENE_MEDIA, SCENE_ROUGH, golden_coeffs, load_pkg, make_scene

pkg = load_pkg()


def test_scene_loads_and_pads_film():
    s = make_scene(SCENE_0010, inject=False, width=1280, height=720, max_verts=8)
    d = s.desc
    assert (d.width, d.height) == (1280, 736)          # src/view.c:294-296: padded to multiples of 32
    assert d.num_prims == 4108 and d.num_shapes == 6   # 4105 quads + sphere + cone + cylinder, 6 of 7 shapes
    assert d.max_verts == 8 and d.frame == 1
    kinds = [pkg.MI_SAMPLER_PT]
    assert d.sampler in kinds
    # regression/0010_pt/test01.cam is the legacy 152-byte layout: 60mm f/4 1/125 iso 400, focus 18.986 dm
    assert abs(d.cam.focus - 18.986294) < 1e-4
    assert abs(d.cam.focal_length - 0.6) < 1e-6 and d.cam.f_stop == 4.0
    assert abs(d.cam.exposure_time - 1 / 125) < 1e-7 and d.cam.iso == 400
    assert abs(d.cam.film_width - 0.35) < 1e-6 and abs(d.cam.film_height - 0.35 * 736 / 1280) < 1e-6
    for v in (d.cam.a, d.cam.b, d.cam.n):
        assert abs(sum(x * x for x in v) - 1) < 1e-5


def test_qbvh_equals_reference_tree():
    """The builder must reproduce the reference's tree exactly (tests/golden/tree_0010.npz was dumped from
    the real reference): same prim order, boxes, split axes and leaf ranges."""
    g = np.load(GOLDEN / "tree_0010.npz")
    s = make_scene(SCENE_0010, inject=False, width=1280, height=720, max_verts=8)
    d = s.desc
    assert d.num_nodes == len(g["box"]) == 428
    assert np.array_equal(np.ctypeslib.as_array(d.primid, (d.num_prims,)), g["primid"])
    assert np.allclose(list(d.aabb), g["aabb"], rtol=0, atol=0)
    LEAF = 1 << 63
    stack = [(0, 0)]
    seen = 0
    while stack:
        rn, mn = stack.pop()
        seen += 1
        node = d.nodes[mn]
        box = np.array([[node.aabb[k][c] for c in range(4)] for k in range(6)], dtype=np.float32)
        assert np.array_equal(box, g["box"][rn])
        assert (node.axis0, node.axis00, node.axis01) == tuple(int(x) for x in g["ax"][rn][:3])
        for c in range(4):
            rc, mc = int(g["child"][rn][c]), int(node.child[c])
            assert (rc >> 63) == (mc >> 63)
            if rc >> 63:
                assert rc == mc          # first prim and count
            else:
                stack.append((rc, mc))
    assert seen == 428


def test_qbvh_of_a_motion_blurred_scene_equals_the_reference_tree():
    """scenes/0059_mb (the backdrop's 4096 quads and the cylinder cap move): the reference builds on the shutter-OPEN boxes and refits
    a second box set to the shutter-close state (src/accel.d/qbvhmp.c:259-283,854-873, 1034-1065); tests/golden/tree_0059_mb.npz is
    that tree dumped from the real reference. Same topology, primitive order and shutter-open boxes to the bit, and the same
    shutter-close boxes (mi_scene_desc.nodes_t1) to the bit."""
    from helpers import SCENE_MB
    g = np.load(GOLDEN / "tree_0059_mb.npz")
    s = make_scene(SCENE_MB, inject=False, width=256, height=256, max_verts=8)
    d = s.desc
    assert d.num_nodes == len(g["box"]) and bool(d.nodes_t1)
    assert np.array_equal(np.ctypeslib.as_array(d.primid, (d.num_prims,)), g["primid"])
    stack, seen = [(0, 0)], 0
    while stack:
        rn, mn = stack.pop()
        seen += 1
        node, t1 = d.nodes[mn], d.nodes_t1[mn]
        box = np.array([[node.aabb[k][c] for c in range(4)] for k in range(6)], dtype=np.float32)
        box1 = np.array([[t1.aabb[k][c] for c in range(4)] for k in range(6)], dtype=np.float32)
        assert np.array_equal(box, g["box"][rn]) and np.array_equal(box1, g["box1"][rn]), (rn, box1, g["box1"][rn])
        assert (node.axis0, node.axis00, node.axis01) == tuple(int(x) for x in g["ax"][rn][:3])
        for c in range(4):
            rc, mc = int(g["child"][rn][c]), int(node.child[c])
            assert (rc >> 63) == (mc >> 63)
            if rc >> 63:
                assert rc == mc
            else:
                stack.append((rc, mc))
    assert seen == d.num_nodes
    assert (g["box1"] != g["box"]).any()                      # the two box sets do differ
    # a static scene carries no second set
    assert not make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=8).desc.nodes_t1


def test_leaves_cover_all_prims_once():
    s = make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=4)
    d = s.desc
    covered = np.zeros(d.num_prims, dtype=int)
    for n in range(d.num_nodes):
        for c in range(4):
            ch = int(d.nodes[n].child[c])
            if ch >> 63:
                first, cnt = (ch ^ (1 << 63)) >> 5, ch & 31
                assert cnt <= 6
                covered[first:first + cnt] += 1
    assert (covered == 1).all()


def test_emitters():
    s = make_scene(SCENE_0010, width=256, height=256, max_verts=4)
    l = s.desc.lights
    assert l.num_prims == 3 and l.p_geo == 1.0 and l.p_sky == 0.0
    cdf = [l.cdf[i] for i in range(3)]
    assert cdf[2] == 1.0 and abs(cdf[0] - 1 / 3) < 1e-4 and abs(cdf[1] - 2 / 3) < 1e-4
    # pdf = 1/total area for a single emissive shape of three 0.16 dm^2 quads
    assert abs(l.L[0] - 1 / 0.48) < 2e-3


def test_materials_compiled():
    s = make_scene(SCENE_0010, width=256, height=256, max_verts=4)
    m = s.desc.materials
    assert m[2].bsdf == 0 and m[2].num_ops == 1 and m[2].op[0].kind == 1          # plane: checker -> diffuse
    assert m[5].bsdf == 0 and m[5].num_ops == 2 and m[5].op[1].slot == 3          # light: rd=0, emission
    assert abs(m[5].op[1].mul - 3200) < 1e-3
    assert m[10].bsdf == 1 and abs(m[10].param[0] - 1.3) < 1e-6 and m[10].param[1] == 23   # dielectric 1.3 23
    assert abs(m[10].op[0].roughness - 0.04) < 1e-7
    assert m[8].bsdf == 3 and abs(m[8].mean_cos - 0.85) < 1e-7 and m[8].interior == -1   # medium_rgb (unused in 0010): mu_t colour, mean cosine
    assert m[9].bsdf == 255                                                         # a bare `color` line is no material
    med_scene = make_scene(SCENE_MEDIA, width=256, height=256, max_verts=4)      # keep the owner of the descriptor alive
    med = med_scene.desc.materials
    assert med[14].bsdf == 1 and med[14].interior == 13 and med[14].num_ops == 1    # interior 10 13: glass surface + medium link
    assert med[13].bsdf == 3 and med[13].num_ops == 1 and med[13].op[0].slot == 4 and abs(med[13].mean_cos - 0.6) < 1e-7
    assert abs(med[13].param[3] - 5.0) < 1e-5                                       # scale of mu_t = 1 / 0.2 dm
    r = make_scene(SCENE_ROUGH, width=256, height=256, max_verts=32)
    assert abs(r.desc.materials[10].param[0] - 1.7) < 1e-6 and abs(r.desc.materials[10].op[0].roughness - 0.4) < 1e-7


def test_rgb2spec_direct_fit_close_to_reference_lut():
    """Without the 9.4 MB LUT the host fits coefficients directly; the spectra must agree with the reference
    LUT's (golden) within 4e-3 absolute reflectance (1e-2 for saturated, scaled colours) over the wavelengths that carry weight (360..740 nm)."""
    h = pkg.host_lib()
    # the chromatic fit integrates against the CIE table, which a loaded scene provides
    s = make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=4)
    for e in golden_coeffs():
        rgb = (C.c_float * 3)(*e["rgb"])
        out = (C.c_float * 3)()
        mul = h.ch_rgb_to_coeff(rgb, out, None)
        assert abs(mul - e["mul"]) < 1e-6 * max(1, e["mul"])
        for k, lam in enumerate(range(360, 760, 50)):
            x = (out[0] * lam + out[1]) * lam + out[2]
            val = 0.5 * x / np.sqrt(x * x + 1) + 0.5
            # the two collision-coefficient colours of the media scenes are far more saturated than any reflectance in the tests
            tol = 4e-3 if e["mul"] <= 1 or min(e["rgb"]) == max(e["rgb"]) else 1e-2
            assert abs(val - e["eval_precise"][k]) < tol, (e["rgb"], lam, val, e["eval_precise"][k])
    del s


REF_LUT = REPO / "oracle" / "_ref" / "data" / "ergb2spec.coeff"


@pytest.mark.skipif(not REF_LUT.exists(), reason="the reference's coefficient table is only there where oracle/_ref was built")
def test_lut_reader_equals_reference_fetch_bit_for_bit():
    """ch_rgb_to_coeff(rgb, out, lut) -- the host's reader of the reference-format table -- against coefficients dumped from
    the reference's own rgb2spec_fetch (include/rgb2spec.h:87-128 via oracle/refharness/unit_harness.c): every colour, every bit."""
    h = pkg.host_lib()
    for e in golden_coeffs():
        out = (C.c_float * 3)()
        mul = h.ch_rgb_to_coeff((C.c_float * 3)(*e["rgb"]), out, str(REF_LUT).encode())
        assert np.float32(mul) == np.float32(e["mul"])
        assert [np.float32(x) for x in out] == [np.float32(x) for x in e["coeff"]], (e["rgb"], list(out), e["coeff"])


def test_shipped_scenes_carry_the_reference_coefficients(monkeypatch):
    """scenes/*/test.rgb2spec pin the colour shaders to the reference table's coefficients: a plainly loaded scene (no test-side
    injection) has bit-for-bit the golden coefficients on every colour the reference dump knows; CORONA_MI_RGB2SPEC=fit
    switches back to the host's own fit."""
    table = {tuple(np.float32(e["rgb"])): e for e in golden_coeffs()}
    seen = 0
    for path in (SCENE_0010, SCENE_ROUGH, SCENE_MEDIA):
        s = make_scene(path, width=64, height=64, max_verts=4)
        mats = s.desc.materials
        lines = path.read_text().splitlines()
        for sid in range(int(lines[1].split()[0])):
            tok = lines[2 + sid].split("#")[0].split()
            if not tok or tok[0] != "color":
                continue
            rgb = tuple(np.float32(x) for x in tok[2:5])
            if max(rgb) == 0 or rgb not in table:
                continue
            # a bare colour line is compiled into the materials that use it: look for its op
            ops = [m.op[k] for m in (mats[i] for i in range(s.desc.num_materials)) for k in range(m.num_ops) if m.op[k].kind == 0]
            want = [np.float32(x) for x in table[rgb]["coeff"]]
            used = any(str(sid) in l.split("#")[0].split()[1:] for l in lines[2:2 + int(lines[1].split()[0])] if l.split()[0] in ("mult", "interior"))
            if not used:
                continue                                      # e.g. the medium lines of 0010, which no material refers to
            assert any([np.float32(c) for c in op.coeff] == want and np.float32(op.mul) == np.float32(table[rgb]["mul"]) for op in ops), (path, rgb)
            seen += 1
    assert seen >= 9
    monkeypatch.setenv("CORONA_MI_RGB2SPEC", "fit")
    s = make_scene(SCENE_0010, width=64, height=64, max_verts=4)
    white = [np.float32(x) for x in table[(np.float32(1), np.float32(1), np.float32(1))]["coeff"]]
    ops = [m.op[k] for m in (s.desc.materials[i] for i in range(s.desc.num_materials)) for k in range(m.num_ops) if m.op[k].kind == 0]
    assert ops and not any([np.float32(c) for c in op.coeff] == white for op in ops)       # the closed form for greys, not the table


@pytest.mark.skipif(not REF_LUT.exists(), reason="needs the reference's coefficient table (oracle/_ref)")
def test_scene_caches_are_up_to_date():
    sys.path.insert(0, str(GOLDEN))
    import make_rgb2spec_cache as mk
    for nra2 in sorted((REPO / "scenes").glob("*/test.nra2")):
        assert (nra2.parent / "test.rgb2spec").read_text().splitlines() == mk.cache_lines(nra2), nra2


def test_black_is_special_cased():
    h = pkg.host_lib()
    out = (C.c_float * 3)()
    mul = h.ch_rgb_to_coeff((C.c_float * 3)(0, 0, 0), out, None)
    assert mul == 0.0 and list(out) == [0.0, 0.0, 0.0]


def test_pfm_layout_matches_reference_writer(tmp_path):
    h = pkg.host_lib()
    w, hh = 64, 32
    fb = np.arange(3 * w * hh, dtype=np.float32).reshape(hh, w, 3)
    fn = str(tmp_path / "t.pfm").encode()
    assert h.ch_pfm_write(fn, fb.ctypes.data, w, hh, 0.5) == 0
    raw = open(fn, "rb").read()
    header_end = raw.index(b"\n", raw.index(b"-1.0")) + 1
    assert raw.startswith(b"PF\n64 32\n-1.0") and header_end % 16 == 0        # fb_export pads to 16 bytes
    data = np.frombuffer(raw[header_end:], dtype="<f4").reshape(hh, w, 3)
    assert np.array_equal(data, fb * 0.5)                                       # rows j=0 first, scaled by gain


def test_bad_scenes_fail_loudly(tmp_path):
    bad = tmp_path / "bad.nra2"
    bad.write_text("daylight 1 2 3\n1\ndiffuse\n0\n")
    with pytest.raises(RuntimeError):
        pkg.Scene(bad)                              # only the black sky is in scope
    bad.write_text("black\n1\nhair 1 2\n1\n0 nonexistent_geo\n")
    s = pkg.Scene(bad)                              # missing .geo: shape is skipped like src/prims.c:783-788
    assert s.desc.num_shapes == 0 and s.desc.num_prims == 0
    with pytest.raises(RuntimeError):
        pkg.Scene(tmp_path / "does_not_exist.nra2")


def test_gain_formula():
    s = make_scene(SCENE_0010, inject=False, width=256, height=256, max_verts=4)
    assert abs(s.gain(64) - 400 / (100 * 64)) < 1e-9   # src/view.c:651-657: gain * iso / (100 * spp)


def test_pfmdiff_tool_matches_reference_tool(tmp_path):
    """host/pfmdiff-mi prints the reference's figure (tools/img/pfmdiff.c:75-86) for two PFM images; where the reference's
    own tool has been built (oracle/_ref/pfmdiff, build container only) the two outputs are compared directly"""
    import subprocess
    rng = np.random.default_rng(5)
    a = rng.random((24, 32, 3)).astype(np.float32)
    b = (a + 0.01 * rng.standard_normal(a.shape)).astype(np.float32)

    def write(fn, img):
        with open(fn, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
            f.write(img[::-1].tobytes())
    write(tmp_path / "a.pfm", a); write(tmp_path / "b.pfm", b)
    tool = REPO / "corona-13_amd" / "host" / "pfmdiff-mi"
    out = subprocess.run([str(tool), str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")], capture_output=True, text=True, check=True).stdout
    rmse = float(out.split("rmse:")[1])
    want = float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum() / (a.shape[0] * a.shape[1])))
    assert abs(rmse - want) < 1e-5 * want
    ref = REPO / "oracle" / "_ref" / "pfmdiff"
    if ref.exists():
        rout = subprocess.run([str(ref), str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm"), str(tmp_path / "d.pfm")], capture_output=True, text=True).stdout
        if "rmse:" in rout:
            assert abs(float(rout.split("rmse:")[1].split()[0]) - rmse) < 1e-5 * want


def test_cli_info_validates_scene_files(tmp_path):
    """corona-mi --info: the host loaders as a validator of .nra2 / .geo / .cam (no GPU involved)"""
    import shutil
    import subprocess
    cli = REPO / "corona-13_amd" / "host" / "corona-mi"
    out = subprocess.run([str(cli), str(SCENE_0010), "-w", "1280", "-h", "720", "--max-verts", "8", "--info"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "film     : 1280x736" in out.stdout and "4108 (spheres 1, lines 2, triangles 0, quads 4105)" in out.stdout
    assert "qbvh, 428 nodes, 1283 leaves" in out.stdout and "(max 6)" in out.stdout and "emitters : 3 primitives" in out.stdout
    # a truncated geometry file is reported, the shape is skipped like the reference does (src/prims.c:783-788)
    shutil.copytree(REPO / "scenes", tmp_path / "scenes")
    geo = tmp_path / "scenes" / "geo" / "sphere.geo"
    geo.write_bytes(geo.read_bytes()[:40])
    bad = subprocess.run([str(cli), str(tmp_path / "scenes" / "0010_pt" / "test.nra2"), "--info"], capture_output=True, text=True)
    assert "sphere" in bad.stderr and ("spheres 0" in bad.stdout or bad.returncode != 0)
    # a .geo header that promises 2^61 primitives (the products in the bounds checks would wrap around): rejected, shape skipped
    geo = tmp_path / "scenes" / "geo" / "cone.geo"
    raw = bytearray(geo.read_bytes())
    raw[8:16] = struct.pack("<Q", 1 << 61)
    geo.write_bytes(bytes(raw))
    bad = subprocess.run([str(cli), str(tmp_path / "scenes" / "0010_pt" / "test.nra2"), "--info"], capture_output=True, text=True)
    assert "cone" in bad.stderr and "bad magic/version/offsets" in bad.stderr and bad.returncode in (0, 2)
    # a missing scene file is an error
    assert subprocess.run([str(cli), str(tmp_path / "nothing.nra2"), "--info"], capture_output=True, text=True).returncode != 0


def test_camera_files_static_and_moving():
    """legacy 152-byte camera (0010) resolves to a static frame; the CCAM v1 file of scenes/0058_cam_mb carries distinct shutter-open
    and shutter-close states: both quaternions and positions are handed over, the kernel interpolates per path"""
    from helpers import SCENE_CAM_MB
    st_scene = make_scene(SCENE_0010, width=256, height=256, max_verts=4)        # keep the owners of the descriptors alive
    st = st_scene.desc.cam
    assert st.moving == 0 and list(st.pos) == list(st.pos_t1) and list(st.orient) == list(st.orient_t1)
    n = np.array(st.n)
    assert abs(np.linalg.norm(n) - 1) < 1e-6 and abs(st.time_scale - (1 / 125) / (1 / 30)) < 1e-6
    mv_scene = make_scene(SCENE_CAM_MB, width=256, height=256, max_verts=4)
    mv = mv_scene.desc.cam
    assert mv.moving == 1 and abs(mv.time_scale - 1.0) < 1e-7
    assert np.allclose(np.array(mv.pos_t1) - np.array(mv.pos), [0.6, 0.3, 0.15], atol=1e-5)
    q0, q1 = np.array(mv.orient), np.array(mv.orient_t1)
    assert abs(np.linalg.norm(q0) - 1) < 1e-4 and abs(np.linalg.norm(q1) - 1) < 1e-4      # the 0010 camera file itself is normalised to 3e-5
    assert abs(2 * np.degrees(np.arccos(min(1.0, abs(float(q0 @ q1) / float(np.linalg.norm(q0) * np.linalg.norm(q1)))))) - 4.0) < 5e-2   # turned by 4 degrees


def test_motion_blurred_geo_is_loaded_with_enclosing_boxes():
    """tools/make_geo.py mb: primid bit 60 set, vertices interleaved shutter open / close; the host tree's leaf boxes enclose both states"""
    from helpers import SCENE_MB
    s = make_scene(SCENE_MB, width=256, height=256, max_verts=4)
    d = s.desc
    prim = np.ctypeslib.as_array(d.primid, shape=(d.num_prims,))
    mb = (prim >> np.uint64(60)) & np.uint64(1)
    assert mb.sum() == 4096 + 6 and d.num_prims == 4108            # backdrop + cylinder cap move, emitter / sphere / lines do not
    assert d.num_vtx > 2 * 4096                                     # two states per vertex of the moving shapes
    # scene box: the backdrop rises by 0.15 dm, the box must contain the shutter-close state
    st_scene = make_scene(SCENE_0010, width=256, height=256, max_verts=4)
    st = st_scene.desc
    assert d.aabb[5] >= st.aabb[5] - 1e-6 and d.aabb[2] <= st.aabb[2] + 1e-6
    assert max(d.aabb[3] - st.aabb[3], d.aabb[5] - st.aabb[5]) > 0.1


def test_regression_report_pieces(tmp_path):
    """tests/regression_report.py (the reference's regression/createres.sh flow): per-test files exist for every scene, the PNG
    writer produces a decodable image and the report lists every test with its verdict"""
    import importlib.util, zlib, struct
    spec = importlib.util.spec_from_file_location("regression_report", REPO / "tests" / "regression_report.py")
    rr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rr)
    for d in sorted((REPO / "scenes").iterdir()):
        if d.is_dir() and d.name[:4].isdigit():
            assert (d / "title").exists() and (d / "config.mk").read_text().startswith("MOD_sampler=")
            # a scene without `args` is not a regression test (scenes/0065_huge: bench and residency probe only; 0066_smooth: path fixtures only)
            assert not (d / "args").exists() or (d / "maxerror").exists(), d.name
    img = np.random.default_rng(0).uniform(0, 1, size=(8, 16, 3)).astype(np.float32)
    rr.write_png(tmp_path / "a.png", img)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (16, 8)
    idat = raw[raw.index(b"IDAT") + 4:raw.index(b"IEND") - 8]
    assert len(zlib.decompress(idat)) == 8 * (1 + 16 * 3)
    (tmp_path / "0010_pt").mkdir()
    rr.write_png(tmp_path / "0010_pt" / "testrender.png", img)
    rr.write_report(tmp_path, [dict(name="0010_pt", title="simplemost path tracing", maxerror=4.0, rmse=1.25, status="pass_1", log="ok", ref_seconds=7.5, seconds=0.1),
                               dict(name="0059_mb", title="motion-blurred geometry", maxerror=4.0, rmse=None, status="pass_crash", log="boom", ref_seconds=None)])
    page = (tmp_path / "report.html").read_text()
    assert "0010_pt" in page and "rmse=1.25" in page and 'class="pass_crash"' in page and "testrender.png" in page
