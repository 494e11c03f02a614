#!/usr/bin/env python3
"""(build container) golden per-path dumps of the reference built with -DMF_COUNT=4 (hero wavelengths, include/mf.h:280-423; `make -C oracle mf4`):
oracle/refharness/render_dump.c writes the usual record with the hero component (index 0) of every spectral quantity plus an extension block with all
four. tests/golden/paths_mf4_*.npz: records (the layout of mi_path_record), ext_* arrays."""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden import GOLD, REF, REPO, run_ref      # noqa: E402

MV_DUMP, NS_DUMP = 8, 8
MF = 8 if "--mf=8" in sys.argv or ("--mf" in sys.argv and sys.argv[sys.argv.index("--mf") + 1] == "8") else 4       # --mf 8: the AVX build (`make -C oracle mf8`)
EXT = np.dtype([("lambda", "<f4", MF), ("throughput", "<f4", (MV_DUMP, MF)), ("pdf", "<f4", (MV_DUMP, MF)), ("rd", "<f4", (MV_DUMP, MF)), ("rg", "<f4", (MV_DUMP, MF)),
                ("em", "<f4", (MV_DUMP, MF)), ("eta", "<f4", (MV_DUMP, MF)), ("splat_value", "<f4", (NS_DUMP, MF))])


def dump(name, binary, mv, scene, w, h, n):
    from helpers import load_pkg
    pkg = load_pkg()
    work = Path(tempfile.mkdtemp(prefix="corona_mf4_"))
    fn = work / "paths.bin"
    run_ref(binary, mv, scene, ["-s", "1", "-w", str(w), "-h", str(h), "-t", "1", "-x", "_dump"],
            env={"CORONA_DUMP_N": str(n), "CORONA_DUMP_FILE": str(fn), "LD_LIBRARY_PATH": str(REF / f"mf{MF}" / f"shaders_mv{mv}")}, work=work)
    raw = fn.read_bytes()
    hdr = np.frombuffer(raw, dtype="<u4", count=4)
    rdt = pkg.record_dtype()
    both = np.dtype([("rec", rdt), ("ext", EXT)])
    assert hdr[1] == both.itemsize and (hdr[2] >> 16) == MF, (hdr, both.itemsize)
    data = np.frombuffer(raw, dtype=both, offset=16)[:n]
    np.savez_compressed(GOLD / f"paths_{name}.npz", records=np.ascontiguousarray(data["rec"]), ext=np.ascontiguousarray(data["ext"]), width=w, height=h, max_verts=mv, mf_count=MF)
    print("wrote", name, len(data), "records; mean length", data["rec"]["length"].mean(), "splats", data["rec"]["num_splats"].sum())
    subprocess.run(["rm", "-rf", str(work)])


def main():
    if MF == 8:
        # eight wavelengths per path (include/mf.h:22-279): pt, ptdl and smooth glass (one component survives a specular transmission: the eighth)
        subprocess.check_call(["make", "-C", str(REPO / "oracle"), "mf8"], stdout=subprocess.DEVNULL)
        dump("mf8_pt_mv8", "mf8/dump_pt_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump("mf8_ptdl_mv8", "mf8/dump_ptdl_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
        dump("mf8_smooth_ptdl_mv8", "mf8/dump_ptdl_xs_mv8", 8, "0066_smooth", 1280, 720, 3000)
        dump("mf8_media_ptdl_mv8", "mf8/dump_ptdl_xs_mv8", 8, "0055_media", 1280, 720, 3000)      # mf_exp = exp256_ps in this build
        return
    subprocess.check_call(["make", "-C", str(REPO / "oracle"), "mf4"], stdout=subprocess.DEVNULL)
    dump("mf4_pt_mv8", "mf4/dump_pt_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
    dump("mf4_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0010_pt", 1280, 720, 3000)
    dump("mf4_rough_mv32", "mf4/dump_pt_xs_mv32", 32, "0052_rough", 1280, 720, 2000)
    dump("mf4_metal_mv8", "mf4/dump_pt_xs_mv8", 8, "0053_metal", 1280, 720, 2000)
    dump("mf4_smooth_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0066_smooth", 1280, 720, 3000)   # specular transmission: one component survives
    # MOD_pointsampler = halton: the four draws of path_init ask for the SAME dimension, so the components are a quarter of the range apart
    dump("mf4_halton_ptdl_mv8", "mf4/dump_ptdl_halton_mv8", 8, "0010_pt", 1280, 720, 3000)
    dump("mf4_halton_rough_mv32", "mf4/dump_pt_halton_mv32", 32, "0052_rough", 1280, 720, 2000)
    # the extended scenes: media (mu_t per component, the free-flight distance from the hero's), moving camera, moving geometry and emitters
    dump("mf4_media_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0055_media", 1280, 720, 4000)
    dump("mf4_media_pt_mv32", "mf4/dump_pt_xs_mv32", 32, "0055_media", 1280, 720, 3000)
    dump("mf4_fog_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0056_fog", 1280, 720, 3000)
    dump("mf4_nested_pt_mv8", "mf4/dump_pt_xs_mv8", 8, "0057_nested", 1280, 720, 3000)
    dump("mf4_cam_mb_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0058_cam_mb", 1280, 720, 3000)
    dump("mf4_mb_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0059_mb", 1280, 720, 3000)
    dump("mf4_all_ptdl_mv8", "mf4/dump_ptdl_xs_mv8", 8, "0061_all", 1280, 720, 2500)


if __name__ == "__main__":
    main()
