#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REAL reference (hanatos/corona-13).

Runs only in the build container: needs oracle/_ref/ (built by `make -C oracle ref` from the
sources under /root/reference) and writes small data files that are committed:

  rgb2spec_coeffs.json       reference LUT coefficients for the colours used by the test scenes
  tree_0010.npz              QBVH of regression/0010_pt as built by the reference (topology, boxes, prim order)
  paths_<cfg>.npz            per-path records of the first N path indices (xorshift128p, -t 1: bit-reproducible)
  counters.json              -DACCEL_DEBUG work counters (rays, node visits, box hits, prim tests) for 1 spp
  tilemeans_<cfg>.npz        32x32 tile means + image means of high-spp reference renders (statistical oracle)

Usage: python3 tests/golden/make_golden.py [quick|paths|mb|mbrl|smooth|images|all]
"""
import json
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
REF = REPO / "oracle" / "_ref"
GOLD = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO / "tests"))


def run_ref(binary, mv, scene, args, env=None, work=None):
    """run a reference binary on a scratch copy of scenes/ (it writes next to the scene file)"""
    work = Path(work or tempfile.mkdtemp(prefix="corona_ref_"))
    if not (work / "scenes").exists():
        shutil.copytree(REPO / "scenes", work / "scenes")
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = str(REF / f"shaders_mv{mv}")
    e.update(env or {})
    cmd = [str(REF / binary), str(work / "scenes" / scene / "test.nra2")] + args
    out = subprocess.run(cmd, cwd=REF, env=e, capture_output=True, text=True)
    if out.returncode:
        raise RuntimeError(out.stdout + out.stderr)
    return work, out.stdout + out.stderr


def read_pfm(fn):
    with open(fn, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)


def dump_paths(name, binary, mv, scene, w, h, n):
    from helpers import load_pkg
    pkg = load_pkg()
    work = Path(tempfile.mkdtemp(prefix="corona_ref_"))
    fn = work / "paths.bin"
    run_ref(binary, mv, scene, ["-s", "1", "-w", str(w), "-h", str(h), "-t", "1", "-x", "_dump"],
            env={"CORONA_DUMP_N": str(n), "CORONA_DUMP_FILE": str(fn), "CORONA_DUMP_TREE": str(work / "tree.bin")}, work=work)
    raw = fn.read_bytes()
    hdr = np.frombuffer(raw, dtype="<u4", count=4)
    dt = pkg.record_dtype()
    assert hdr[1] == dt.itemsize
    rec = np.frombuffer(raw, dtype=dt, offset=16)[:n]
    np.savez_compressed(GOLD / f"paths_{name}.npz", records=rec, width=w, height=h, max_verts=mv)
    print("wrote", name, len(rec), "records")
    return work


def dump_tree(work, name="tree_0010"):
    d = (work / "tree.bin").read_bytes()
    magic, nn, npr = struct.unpack("<QQQ", d[:24])
    off = 24
    aabb = np.frombuffer(d, dtype="<f4", count=6, offset=off); off += 24
    node_dt = np.dtype([("box", "<f4", (6, 4)), ("child", "<u8", 4), ("ax", "<i4", 4)])
    nodes = np.frombuffer(d, dtype=node_dt, count=nn, offset=off); off += nn * node_dt.itemsize
    prim = np.frombuffer(d, dtype="<u8", count=npr, offset=off); off += 8 * npr
    box1 = np.frombuffer(d, dtype="<f4", count=nn * 24, offset=off).reshape(nn, 6, 4)      # shutter-close boxes (qbvh_node_t.aabb1)
    np.savez_compressed(GOLD / f"{name}.npz", aabb=aabb, box=nodes["box"], child=nodes["child"], ax=nodes["ax"], primid=prim, box1=box1)
    print("wrote", name, nn, "nodes")


def dump_mb_tree():
    """the reference's tree of scenes/0059_mb (moving backdrop and cylinder cap): built on the shutter-open boxes, second box set
    refitted to the shutter-close state (src/accel.d/qbvhmp.c:259-283,854-873)"""
    work = Path(tempfile.mkdtemp(prefix="corona_ref_"))
    run_ref("dump_pt_xs_mv8", 8, "0059_mb", ["-s", "1", "-w", "256", "-h", "256", "-t", "1", "-x", "_dump"],
            env={"CORONA_DUMP_N": "16", "CORONA_DUMP_FILE": str(work / "paths.bin"), "CORONA_DUMP_TREE": str(work / "tree.bin")}, work=work)
    dump_tree(work, "tree_0059_mb")


def counters():
    out = {}
    for name, binary, scene in [("pt_mv8", "corona_pt_xs_mv8_dbg", "0010_pt"), ("ptdl_mv8", "corona_ptdl_xs_mv8_dbg", "0010_pt"),
                                # moving geometry: the reference interpolates the node boxes per ray (qbvhmp.c:1208-1224)
                                ("mb_pt_mv8", "corona_pt_xs_mv8_dbg", "0059_mb"), ("mb_round_pt_mv8", "corona_pt_xs_mv8_dbg", "0062_mb_round")]:
        _, log = run_ref(binary, 8, scene, ["-s", "1", "-w", "1280", "-h", "720", "-t", "1", "-x", "_dbg"])
        m = re.search(r"accel_intersect: (\d+) aabb_intersect (\d+) / (\d+) prims_intersect (\d+)", log)
        out[name] = {"paths": 1280 * 736, "rays": int(m.group(1)), "box_hits": int(m.group(2)),
                     "node_visits": int(m.group(3)), "prim_tests": int(m.group(4))}
    (GOLD / "counters.json").write_text(json.dumps(out, indent=1))
    print(out)


def tilemeans(name, binary, mv, scene, w, h, spp, threads=8):
    work, log = run_ref(binary, mv, scene, ["-s", str(spp), "--batch", "16", "-w", str(w), "-h", str(h), "-t", str(threads), "-x", "_img"])
    img = read_pfm(work / "scenes" / scene / "test_img_fb00.pfm")
    H, W, _ = img.shape
    tiles = img.reshape(H // 32, 32, W // 32, 32, 3).mean(axis=(1, 3))
    side = (work / "scenes" / scene / "test_img_fb00.pfm.txt").read_text()
    m = re.search(r"elapsed wallclock prog ([\d.]+)s", side)
    np.savez_compressed(GOLD / f"tilemeans_{name}.npz", tiles=tiles.astype(np.float32), mean=img.mean(axis=(0, 1)),
                        spp=spp, width=W, height=H, max_verts=mv, seconds=float(m.group(1)) if m else 0.0, threads=threads)
    print("wrote tilemeans", name, img.mean(axis=(0, 1)), "in", m.group(1) if m else "?", "s")
    shutil.rmtree(work, ignore_errors=True)


def image_pair(name, binary, mv, scene, w, h, spp, threads=8):
    """two independent high-spp renders of a tiny film (frames 1 and 2 seed the generators differently): per-pixel golden
    image plus the reference's own per-pixel noise floor at that sample count"""
    imgs = []
    for frame in (1, 2):
        work, log = run_ref(binary, mv, scene, ["-s", str(spp), "--batch", "1024", "-w", str(w), "-h", str(h), "-t", str(threads),
                                                "--frame", str(frame), "-x", "_img"])
        imgs.append(read_pfm(work / "scenes" / scene / "test_img_fb00.pfm").copy())
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(GOLD / f"image_{name}.npz", a=imgs[0].astype(np.float32), b=imgs[1].astype(np.float32), spp=spp, max_verts=mv)
    a, b = imgs
    print("wrote image pair", name, a.shape, "means", a.mean(axis=(0, 1)), b.mean(axis=(0, 1)),
          "rmse(a,b)/mean", np.sqrt(((a - b) ** 2).mean()) / a.mean())


COLOURS = [(0.3, 0.3, 0.3), (1, 1, 1), (3200, 3200, 3200), (10, 10, 10), (0.99, 0.96, 0.94), (0.5, 0.5, 0.5), (0.8, 0.2, 0.1), (0.1, 0.7, 0.3),
           (1 / 0.5, 1 / 0.3, 1 / 0.2), (1 / 0.014, 1 / 0.005, 1 / 0.003), (1 / 30, 1 / 40, 1 / 60), (1 / 1.5, 1 / 0.8, 1 / 0.4)]
# the last three: mu_t of `medium_rgb 0.5 0.3 0.2` (scenes/0055_media), `medium_rgb 30 40 60` (scenes/0056_fog), `medium_rgb 1.5 0.8 0.4` (scenes/0057_nested) and of the `medium_rgb 0.014 0.005 0.003` line every 0010-based scene carries
# unused (regression/0010_pt/test.nra2:10), medium_rgb.c:113-119


def coeffs():
    """rgb2spec_coeffs.json: the reference LUT's coefficients for every colour the test scenes use (oracle/refharness/unit_harness.c)"""
    args = [f"{np.float32(c):.9g}" for rgb in COLOURS for c in rgb]
    out = subprocess.run([str(REF / "unit_harness"), "coeff", "data/ergb2spec.coeff"] + args, cwd=REF, capture_output=True, text=True, check=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == len(COLOURS)
    (GOLD / "rgb2spec_coeffs.json").write_text("\n".join(lines) + "\n")
    print("wrote", len(lines), "colours")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "quick"
    if what in ("coeffs", "all"):
        coeffs()
    if what in ("quick", "paths", "all"):
        work = dump_paths("pt_mv8", "dump_pt_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump_tree(work)
        dump_paths("ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump_paths("pt_mv4_256", "dump_pt_xs_mv4", 4, "0010_pt", 256, 256, 2000)
        dump_paths("rough_mv32", "dump_pt_xs_mv32", 32, "0052_rough", 1280, 720, 2000)
        dump_paths("fine_mv8", "dump_pt_xs_mv8", 8, "0054_fine", 1280, 720, 3000)     # needs scenes/geo/plane_fine.geo (tools/make_geo.py)
        dump_paths("metal_mv8", "dump_pt_xs_mv8", 8, "0053_metal", 1280, 720, 3000)
        dump_paths("metal_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0053_metal", 1280, 720, 3000)
        # homogeneous medium inside the glass sphere (`interior`, `medium_rgb`)
        dump_paths("media_pt_mv8", "dump_pt_xs_mv8", 8, "0055_media", 1280, 720, 10000)
        dump_paths("media_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0055_media", 1280, 720, 10000)
        dump_paths("media_pt_mv32", "dump_pt_xs_mv32", 32, "0055_media", 1280, 720, 6000)
        dump_paths("fog_pt_mv8", "dump_pt_xs_mv8", 8, "0056_fog", 1280, 720, 10000)
        dump_paths("fog_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0056_fog", 1280, 720, 10000)
        dump_paths("media_ptdl_mv32", "dump_ptdl_xs_mv32", 32, "0055_media", 1280, 720, 6000)
        dump_paths("nested_pt_mv8", "dump_pt_xs_mv8", 8, "0057_nested", 1280, 720, 10000)
        dump_paths("nested_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0057_nested", 1280, 720, 10000)
        dump_paths("cam_mb_pt_mv8", "dump_pt_xs_mv8", 8, "0058_cam_mb", 1280, 720, 4000)        # camera motion blur
        dump_paths("cam_mb_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0058_cam_mb", 1280, 720, 4000)
        dump_paths("mb_pt_mv8", "dump_pt_xs_mv8", 8, "0059_mb", 1280, 720, 6000)                # motion-blurred backdrop and cylinder cap (tools/make_geo.py mb)
        dump_paths("mb_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0059_mb", 1280, 720, 6000)
        dump_paths("mb_round_pt_mv8", "dump_pt_xs_mv8", 8, "0062_mb_round", 1280, 720, 6000)        # moving sphere / cone / cylinder
        dump_paths("mb_round_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0062_mb_round", 1280, 720, 6000)
        dump_paths("mb_light_pt_mv8", "dump_pt_xs_mv8", 8, "0060_mb_light", 1280, 720, 6000)        # moving emitter
        dump_paths("mb_light_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0060_mb_light", 1280, 720, 6000)
    if what in ("paths", "all", "mbrl"):
        dump_paths("mb_round_light_pt_mv8", "dump_pt_xs_mv8", 8, "0063_mb_round_light", 1280, 720, 6000)   # moving sphere and cone as emitters
        dump_paths("mb_round_light_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0063_mb_round_light", 1280, 720, 6000)
    if what in ("paths", "all"):
        dump_paths("halton_all_ptdl_mv8", "dump_ptdl_halton_mv8", 8, "0061_all", 1280, 720, 6000)    # every feature in one scene
        dump_paths("all_pt_mv32", "dump_pt_xs_mv32", 32, "0061_all", 1280, 720, 4000)
        # MOD_pointsampler=halton; the mv32 ptdl case reaches dimensions >= 256 (fallback to the per-path generator)
        dump_paths("halton_pt_mv8", "dump_pt_halton_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump_paths("halton_ptdl_mv8", "dump_ptdl_halton_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump_paths("halton_fog_ptdl_mv8", "dump_ptdl_halton_mv8", 8, "0056_fog", 1280, 720, 6000)   # free-flight dimension from the Halton sampler
        dump_paths("halton_ptdl_rough_mv32", "dump_ptdl_halton_mv32", 32, "0052_rough", 1280, 720, 2000)
        counters()
    if what in ("smooth", "paths", "all"):
        # smooth glass (scenes/0066_smooth = 0010 with roughness 0): the specular branches of dielectric.c sample / brdf / pdf
        dump_paths("smooth_pt_mv8", "dump_pt_xs_mv8", 8, "0066_smooth", 1280, 720, 3000)
        dump_paths("smooth_ptdl_mv8", "dump_ptdl_xs_mv8", 8, "0066_smooth", 1280, 720, 3000)
    if what in ("mb", "quick", "paths", "all"):
        dump_mb_tree()
        if what == "mb":
            counters()
    if what in ("images", "all"):
        tilemeans("pt_mv8", "corona_pt_sfmt_mv8", 8, "0010_pt", 1280, 720, 2048)
        tilemeans("ptdl_mv8", "corona_ptdl_sfmt_mv8", 8, "0010_pt", 1280, 720, 512)
        tilemeans("rough_mv32", "corona_pt_sfmt_mv32", 32, "0052_rough", 1280, 720, 512)
        tilemeans("pt_mv4_256", "corona_pt_sfmt_mv4", 4, "0010_pt", 256, 256, 4096)
        image_pair("pt_mv8_64", "corona_pt_sfmt_mv8", 8, "0010_pt", 64, 64, 65536)
        tilemeans("metal_ptdl_mv8", "corona_ptdl_sfmt_mv8", 8, "0053_metal", 1280, 720, 1024)


if __name__ == "__main__":
    main()
