#!/usr/bin/env python3
"""development helper (build container): registers, spills and scratch of every kernel in a gfx950 code object, read from the
.amdgpu_metadata note.  tools/kstat.py <file.s | libcorona_mi.so> [filter]
For a .so the device code object is extracted first (clang-offload-bundler)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def metadata_text(path):
    p = Path(path)
    if p.suffix == ".s":
        return p.read_text()
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fatbin"
        subprocess.check_call([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(p), str(fat)])
        blob = fat.read_bytes()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        for i in range(len(starts) - 1):          # one bundle per translation unit
            b = Path(td) / f"bundle{i}"
            co = Path(td) / f"dev{i}.co"
            b.write_bytes(blob[starts[i]:starts[i + 1]])
            subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={b}",
                                   f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
            out.append(subprocess.check_output([str(LLVM / "llvm-readelf"), "--notes", str(co)], text=True))
    return "\n".join(out)


def demangle(names):
    out = subprocess.check_output(["c++filt"], input="\n".join(names), text=True)
    return out.splitlines()


def main():
    text = metadata_text(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for blk in re.split(r"\n\s*- \.agpr_count:", text)[1:]:
        def g(key):
            m = re.search(r"\.%s:\s*(\S+)" % key, blk)
            return m.group(1) if m else "?"
        rows.append((g("name"), g("vgpr_count"), "%s" % blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                     g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = demangle([r[0] for r in rows])
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt in n:
            print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>6} {r[5]:>6} {r[6]:>7} {r[7]:>6}  {n}")


if __name__ == "__main__":
    main()
