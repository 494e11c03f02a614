#!/usr/bin/env python3
"""Regression report in the manner of the reference's regression/Makefile + createres.sh, for a GPU box:

    python3 tests/regression_report.py [--out gpurun_out/regression] [scene ...]

For every scenes/NNNN_*/ with an `args` file (same per-test files as the reference keeps: args, maxerror, title, config.mk with
MOD_sampler): render with the stand-alone renderer corona-13_amd/host/corona-mi, render the same test with the REAL reference
binary (oracle/_ref/, CPU; skipped when it is not there), compare with pfmdiff-mi (per-pixel RMSE on the gain-scaled images,
createres.sh:21-23) against `maxerror`, and write report.html with tone-mapped previews. Lives under tests/ because it runs the
reference build under oracle/, which only test code may do."""
import argparse
import html
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time
import zlib
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
CLI = REPO / "corona-13_amd" / "host" / "corona-mi"
PFMDIFF = REPO / "corona-13_amd" / "host" / "pfmdiff-mi"
REF = REPO / "oracle" / "_ref"


def read_pfm(fn):
    with open(fn, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), dtype="<f4", count=3 * w * h).reshape(h, w, 3)


def write_png(fn, xyz):
    """XYZ -> sRGB (D65 matrix, gamma), 8 bit; a minimal PNG writer (zlib only)"""
    m = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]], dtype=np.float32)
    rgb = np.clip(xyz @ m.T, 0.0, 1.0)
    srgb = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(rgb, 1 / 2.4) - 0.055)
    img = (np.clip(srgb, 0, 1) * 255 + 0.5).astype(np.uint8)
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    Path(fn).write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def run_test(scene_dir, out, keep_pfm=False, wavelengths=1):
    name = scene_dir.name
    args = (scene_dir / "args").read_text().split()
    maxerror = float((scene_dir / "maxerror").read_text()) if (scene_dir / "maxerror").exists() else 0.11
    title = (scene_dir / "title").read_text().strip() if (scene_dir / "title").exists() else name
    sampler = "pt"
    if (scene_dir / "config.mk").exists():
        for line in (scene_dir / "config.mk").read_text().splitlines():
            if line.startswith("MOD_sampler="):
                sampler = line.split("=", 1)[1].strip()
    work = out / name
    work.mkdir(parents=True, exist_ok=True)
    res = {"name": name, "title": title, "maxerror": maxerror, "rmse": None, "status": "pass_crash", "log": "", "ref_seconds": None}
    # like the reference, the renderer writes <basename><postfix>_fb00.pfm next to the scene file: work on a scratch copy of scenes/
    tmp = Path(tempfile.mkdtemp(prefix="corona_reg_"))
    shutil.copytree(REPO / "scenes", tmp / "scenes")
    t0 = time.time()
    p = subprocess.run([str(CLI), str(tmp / "scenes" / name / "test.nra2")] + args + ["--sampler", sampler, "--max-verts", "8", "-x", "_mi"] +
                       (["--wavelengths", str(wavelengths)] if wavelengths != 1 else []), capture_output=True, text=True)
    res["seconds"] = time.time() - t0
    res["log"] = (p.stdout + p.stderr)[-2000:]
    render = work / "testrender_fb00.pfm"
    made = tmp / "scenes" / name / "test_mi_fb00.pfm"
    if p.returncode or not made.exists():
        shutil.rmtree(tmp, ignore_errors=True)
        return res
    shutil.copy(made, render)
    write_png(work / "testrender.png", read_pfm(render))
    # the reference itself, same sampler, PATHSPACE_MAX_VERTS 8, its default generator (sfmt)
    binary = REF / f"corona_{sampler}_sfmt_mv8"
    ref_args = list(args)
    if "-s" in ref_args:          # the stored reference of the original flow is a converged image: four times the samples here
        ref_args[ref_args.index("-s") + 1] = str(4 * int(ref_args[ref_args.index("-s") + 1]))
    if binary.exists():
        env = dict(os.environ, LD_LIBRARY_PATH=str(REF / "shaders_mv8"))
        t0 = time.time()
        r = subprocess.run([str(binary), str(tmp / "scenes" / name / "test.nra2")] + ref_args + ["--batch", "16", "-t", str(min(32, os.cpu_count() or 8)), "-x", "_ref"],
                           cwd=REF, env=env, capture_output=True, text=True)
        res["ref_seconds"] = time.time() - t0
        refpfm = tmp / "scenes" / name / "test_ref_fb00.pfm"
        if r.returncode == 0 and refpfm.exists():
            shutil.copy(refpfm, work / "reference.pfm")
            write_png(work / "reference.png", read_pfm(refpfm))
    elif (scene_dir / "reference.pfm").exists():
        shutil.copy(scene_dir / "reference.pfm", work / "reference.pfm")
        write_png(work / "reference.png", read_pfm(work / "reference.pfm"))
    shutil.rmtree(tmp, ignore_errors=True)
    if (work / "reference.pfm").exists():
        d = subprocess.run([str(PFMDIFF), str(render), str(work / "reference.pfm")], capture_output=True, text=True)
        try:
            res["rmse"] = float(d.stdout.split(":")[1].split()[0])
            res["status"] = "pass_1" if res["rmse"] < maxerror else "pass_0"
        except (IndexError, ValueError):
            res["log"] += d.stdout + d.stderr
    if not keep_pfm:
        for f in (render, work / "reference.pfm"):
            if f.exists():
                f.unlink()
    else:
        res["status"] = "pass_noref"
    return res


def write_report(out, results):
    css = ("body{font-family:sans-serif;background:#222;color:#ddd} h1{font-size:1.1em;padding:.4em;cursor:pointer;margin:.2em 0}"
           ".pass_1{background:#264d26}.pass_0{background:#6b2222}.pass_crash{background:#6b2222}.pass_noref{background:#444}"
           ".content{display:none;padding:.5em 1em} img{max-width:48%;margin:.5%} pre{white-space:pre-wrap;color:#aaa}")
    body = []
    for r in results:
        n = html.escape(r["name"])
        imgs = "".join(f'<img src="{n}/{f}.png" alt="{f}" title="{f}"/>' for f in ("testrender", "reference") if (out / r["name"] / f"{f}.png").exists())
        rm = "no reference image" if r["rmse"] is None else f"rmse={r['rmse']:.4g}, max allowed={r['maxerror']:g}"
        timing = f"corona-mi {r.get('seconds', 0):.2f} s" + (f", reference binary {r['ref_seconds']:.1f} s" if r["ref_seconds"] else "")
        body.append(f'<h1 class="{r["status"]}" onclick="toggle(\'{n}\');">{n} &mdash; {html.escape(r["title"])} &mdash; {rm}</h1>'
                    f'<div id="{n}" class="content">{imgs}<h3>{timing}</h3><pre>{html.escape(r["log"])}</pre></div>')
    page = ("<!DOCTYPE html><html><head><meta charset='utf-8'/><title>corona-mi regression tests</title><style>" + css + "</style><script>"
            "function toggle(id){var e=document.getElementById(id);e.style.display=(e.style.display=='block')?'none':'block';}</script></head><body>"
            "<h2>corona-mi regression tests</h2><p>click on the headings to expand the detailed reports.</p>" + "\n".join(body) + "</body></html>")
    (out / "report.html").write_text(page)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "regression"))
    ap.add_argument("--keep-pfm", action="store_true", help="keep the float images next to the previews (7 MB each)")
    ap.add_argument("--wavelengths", type=int, default=1, choices=[1, 4],
                    help="4: the test renders carry four wavelengths per path (corona-mi --wavelengths 4, the reference's MF_COUNT = 4); the references stay the "
                         "MF_COUNT = 1 reference's renders -- same expected image, the thresholds are the scalar tests' noise floors")
    ap.add_argument("scenes", nargs="*")
    a = ap.parse_args()
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    dirs = sorted(d for d in (REPO / "scenes").iterdir() if d.is_dir() and (d / "args").exists() and (not a.scenes or d.name in a.scenes))
    results = [run_test(d, out, a.keep_pfm, a.wavelengths) for d in dirs]
    write_report(out, results)
    for r in results:
        print("%-14s %-11s rmse %-10s max %-5g %s" % (r["name"], r["status"], "-" if r["rmse"] is None else "%.4g" % r["rmse"], r["maxerror"], r["title"]))
    return 0 if all(r["status"] in ("pass_1", "pass_noref") for r in results) else 1


if __name__ == "__main__":
    sys.exit(main())
