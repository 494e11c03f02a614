#!/usr/bin/env python3
"""Writes scenes/*/test.rgb2spec: the RGB -> spectrum coefficients of every `color` / `medium_rgb` line of a scene as the
REFERENCE's coefficient table yields them (data/ergb2spec.coeff, made by the reference's own tools/img/rgb2spec_opt.cpp in
oracle/_ref, 9.4 MB, not shipped). The host loader applies such a file when it finds one next to the scene
(corona-13_amd/host/ch_scene.c, apply_coeff_cache), so a scene carries the reference's init-time constants
(include/spectrum.h:29-38, include/rgb2spec.h:87-128) without the table. Build container only (needs oracle/_ref).

The fetch itself is the host library's (ch_rgb2spec_lut.c); tests/test_host.py pins it bit-for-bit against coefficients
dumped from the reference binary (tests/golden/rgb2spec_coeffs.json), and checks these files against the table.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO / "tests"))
from helpers import load_pkg  # noqa: E402

LUT = REPO / "oracle" / "_ref" / "data" / "ergb2spec.coeff"


def scene_colours(nra2):
    """rgb triples (as the host parser reads them: float32) of the color / medium_rgb shaders of a scene file"""
    lines = nra2.read_text().splitlines()
    n = int(lines[1].split()[0])
    out = []
    for sid in range(n):
        tok = lines[2 + sid].split("#")[0].split()
        if not tok:
            continue
        if tok[0] == "color":
            rgb = [np.float32(x) for x in tok[2:5]]
        elif tok[0] == "medium_rgb":            # collision coefficient = 1 / mean free path, src/shaders/medium_rgb.c:113-119
            rgb = [np.float32(1) / np.float32(x) for x in tok[1:4]]
        else:
            continue
        if max(rgb) > 0 and rgb not in out:
            out.append(rgb)
    return out


def cache_lines(nra2, lut=LUT):
    h = load_pkg().host_lib()
    lines = ["# r g b  c0 c1 c2 mul -- coefficients of the reference's ergb2spec.coeff (64^3) for this scene's colours;",
             "# written by tests/golden/make_rgb2spec_cache.py, read by ch_scene.c (apply_coeff_cache)"]
    for rgb in scene_colours(nra2):
        arr = (C.c_float * 3)(*[float(x) for x in rgb])
        out = (C.c_float * 3)()
        mul = h.ch_rgb_to_coeff(arr, out, str(lut).encode())
        lines.append(" ".join("%.9g" % float(np.float32(v)) for v in list(rgb) + list(out) + [mul]))
    return lines


if __name__ == "__main__":
    if not LUT.exists():
        raise SystemExit(f"{LUT} missing: build the reference first (make -C oracle ref)")
    for nra2 in sorted((REPO / "scenes").glob("*/test.nra2")):
        (nra2.parent / "test.rgb2spec").write_text("\n".join(cache_lines(nra2)) + "\n")
        print("wrote", nra2.parent / "test.rgb2spec")
