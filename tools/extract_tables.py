#!/usr/bin/env python3
"""Extract the *measured data tables* the hot path needs from the reference tree into
small binary float32 files under corona-13_amd/data/ (run in the build container only;
/root/reference does not exist on the GPU box).

These are physical measurement data, not code:
  cie1931_xyz.f32     96 x 3   CIE 1931 2-deg colour matching functions, 360..830 nm, 5 nm,
                               plus one zero row (include/spectrum.h:66-170)
  colorchecker_sg.f32 140 x 36 ColorChecker SG reflectances, 380..730 nm, 10 nm
                               (src/shaders/colorcheckersg.c:51-190)
  metal_ior.f32       5 x 95 x 2 (n,k) of Ti,Cu,Fe,Au,Ag, 360..830 nm, 5 nm
                               (src/shaders/fresnel.h:27-516)
"""
import re, sys, os
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "corona-13_amd", "data")
os.makedirs(OUT, exist_ok=True)

def numbers_between(text, start_pat, end_pat):
    s = re.search(start_pat, text).end()
    e = text.index(end_pat, s)
    body = re.sub(r"//[^\n]*", "", text[s:e])
    return np.array([float(x) for x in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", body)], dtype=np.float64)

spec = open(os.path.join(REF, "include/spectrum.h")).read()
cie = numbers_between(spec, r"spectrum_xyz_lut\[\]\s*=\s*\{", "};")
assert cie.size == 96 * 3, cie.size
cie.astype(np.float32).tofile(os.path.join(OUT, "cie1931_xyz.f32"))

cc = open(os.path.join(REF, "src/shaders/colorcheckersg.c")).read()
cobs = numbers_between(cc, r"cobs\[140\]\[36\]\s*=\s*\{", "};")
assert cobs.size == 140 * 36, cobs.size
cobs.astype(np.float32).tofile(os.path.join(OUT, "colorchecker_sg.f32"))

fr = open(os.path.join(REF, "src/shaders/fresnel.h")).read()
ior = numbers_between(fr, r"fresnel_ior\[\]\[95\]\[2\]\s*=\s*\{", "}; ")
assert ior.size == 5 * 95 * 2, ior.size
ior.astype(np.float32).tofile(os.path.join(OUT, "metal_ior.f32"))
print("wrote", OUT, cie.size, cobs.size, ior.size)
