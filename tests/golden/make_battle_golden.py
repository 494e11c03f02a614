#!/usr/bin/env python3
"""Generate tests/golden/battle.json from the REFERENCE's own BSDF battle test (tools/battle-test.c:57-266 -- what
regression/0052_dielectric and 0053_dielectric run), built in oracle/_ref by `make -C oracle ref` through
oracle/refharness/battle_main.c. Build container only (needs /root/reference).

Per case four incidence angles; per angle the tool prints
    ebsdf  = sum of the weights sample() returned, over 8 * 512^2 samples          (estimate of the integral of bsdf cos)
    bsdf   = brdf() summed over a 512^2 grid on the projected hemisphere             (the same integral from the evaluation)
    epdf   = fraction of samples that landed in the tested hemisphere                (estimate of the integral of the pdf)
    pdf    = pdf() summed over the same grid
and regression/makebattletest.sh:13-14 passes a case iff (bsdf - ebsdf)^2 < 1e-5 and (pdf - epdf)^2 < 1e-5 for every angle."""
import json
import re
import subprocess
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent.parent
REF = REPO / "oracle" / "_ref"

CASES = [
    # name, plugin, the shader line piped to init(), roughness, reflect, MI_BSDF_* kind and parameters for the device hook
    dict(name="0052_dielectric reflect", plugin="libdielectric.so", line="1.7 73 #", roughness=0.4, reflect=1, bsdf="dielectric", param=[1.7, 73.0]),
    dict(name="0053_dielectric transmit", plugin="libdielectric.so", line="1.7 73 #", roughness=0.4, reflect=0, bsdf="dielectric", param=[1.7, 73.0]),
    dict(name="metal Au reflect", plugin="libmetal.so", line="Au #", roughness=0.3, reflect=1, bsdf="metal", param=[3.0, 0.0]),   # fresnel.h:21-27: Ti Cu Fe Au Ag
]

out = []
for c in CASES:
    with tempfile.TemporaryDirectory() as td:       # the tool writes its .pgm images into the working directory
        r = subprocess.run([str(REF / "battle_test"), str(REF / "shaders_mv32" / c["plugin"]), str(c["roughness"]), str(c["reflect"]), "4"],
                           input=c["line"] + "\n", capture_output=True, text=True, cwd=td, check=True)
    rows = [[float(x) for x in m.groups()] for m in re.finditer(r"ebsdf-bsdf-epdf-pdf\[\d+\] (\S+) (\S+) (\S+) (\S+)", r.stdout)]
    assert len(rows) == 4, r.stdout + r.stderr
    passes = [[(row[1] - row[0]) ** 2 < 1e-5, (row[3] - row[2]) ** 2 < 1e-5] for row in rows]
    out.append(dict(c, count=4, **{"lambda": 525.0}, size=512, spp=8, rows=rows, reference_passes_bsdf_pdf=passes))
    print(c["name"], rows, passes)
(REPO / "tests" / "golden" / "battle.json").write_text(json.dumps(out, indent=1) + "\n")
