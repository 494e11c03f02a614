"""Test helpers: package loader, oracle (ctypes) binding, golden coefficients.

The oracle (oracle/liboracle.so) is the CPU restatement of the reference used as the CHECKER.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it.
"""
import ctypes as C
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
SCENE_0010 = REPO / "scenes" / "0010_pt" / "test.nra2"
SCENE_ROUGH = REPO / "scenes" / "0052_rough" / "test.nra2"
SCENE_SMOOTH = REPO / "scenes" / "0066_smooth" / "test.nra2"     # 0010 with roughness 0: specular reflection / transmission
SCENE_LARGE = REPO / "scenes" / "0064_large" / "test.nra2"     # 0010 with every backdrop quad split 8x8: 262 156 primitives, the top of the tree in LDS, the rest in HBM
SCENE_FINE = REPO / "scenes" / "0054_fine" / "test.nra2"       # 0010 with every backdrop quad split 2x2 (tools/make_geo.py): 1711 nodes, too big for LDS
SCENE_MEDIA = REPO / "scenes" / "0055_media" / "test.nra2"     # 0010 with a scattering medium inside the glass sphere (`interior`, `medium_rgb`)
SCENE_FOG = REPO / "scenes" / "0056_fog" / "test.nra2"         # 0010 in a thin global fog (`exterior <medium> 0`)
SCENE_NESTED = REPO / "scenes" / "0057_nested" / "test.nra2"   # fog outside, scattering medium in the sphere, absorbing medium in the cone
SCENE_CAM_MB = REPO / "scenes" / "0058_cam_mb" / "test.nra2"   # 0010 seen by a camera that moves and turns by 4 degrees during the 1/30 s exposure
SCENE_MB = REPO / "scenes" / "0059_mb" / "test.nra2"           # 0010 with the backdrop rising 0.15 dm and the cylinder cap sliding 0.8 dm during a 1/30 s exposure
SCENE_MB_LIGHT = REPO / "scenes" / "0060_mb_light" / "test.nra2"   # 0059 with the emitter moving and turning as well
SCENE_ALL = REPO / "scenes" / "0061_all" / "test.nra2"         # fog, media in sphere and cone, moving camera, moving backdrop / cap / emitter
SCENE_MB_ROUND = REPO / "scenes" / "0062_mb_round" / "test.nra2"   # 0059 with sphere, cone and cylinder moving as well
SCENE_MB_ROUND_LIGHT = REPO / "scenes" / "0063_mb_round_light" / "test.nra2"   # 0062 with the moving sphere and cone as emitters
SCENE_METAL = REPO / "scenes" / "0053_metal" / "test.nra2"     # 0052 with `metal Au`, roughness 0.3 on cone/sphere/cylinder


def load_pkg():
    if "corona13_amd" in sys.modules:
        return sys.modules["corona13_amd"]
    spec = importlib.util.spec_from_file_location("corona13_amd", REPO / "corona-13_amd" / "__init__.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["corona13_amd"] = m
    spec.loader.exec_module(m)
    return m


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        so = Path(os.environ.get("CORONA_ORACLE_LIB", REPO / "oracle" / "liboracle.so"))     # override: the sanitizer build (make sanitize)
        if not so.exists():
            subprocess.check_call(["make", "-C", str(REPO / "oracle"), "liboracle.so"])
        pkg = load_pkg()
        o = C.CDLL(str(so))
        o.oracle_trace_records.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_uint64, C.c_uint64, C.c_void_p]
        o.oracle_trace_records.restype = None
        o.oracle_render.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        o.oracle_render.restype = C.c_double
        o.oracle_intersect.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        o.oracle_intersect.restype = None
        o.oracle_rand_sequence.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        o.oracle_rand_sequence.restype = C.c_float
        o.oracle_set_pixels_from_index.argtypes = [C.c_int]
        o.oracle_hero_trace.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_hero_trace.restype = None
        o.oracle_hero_trace_n.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_hero_trace_n.restype = None
        o.oracle_set_pixels_from_index.restype = None
        o.oracle_render_tiles.argtypes = [C.POINTER(pkg.MiSceneDesc), C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        o.oracle_render_tiles.restype = C.c_double
        _oracle = o
    return _oracle


def oracle_records(scene, first, count):
    pkg = load_pkg()
    out = np.zeros(count, dtype=pkg.record_dtype())
    oracle_lib().oracle_trace_records(scene.desc_ptr, first, count, out.ctypes.data)
    return out


HERO_MF = 4
# oracle_hero_ext (oracle/oracle.h) == dump_ext_t of the dump harness built with -DMF_COUNT=4 (oracle/refharness/render_dump.c)
HERO_EXT = np.dtype([("lambda", "<f4", HERO_MF), ("throughput", "<f4", (8, HERO_MF)), ("pdf", "<f4", (8, HERO_MF)), ("rd", "<f4", (8, HERO_MF)),
                     ("rg", "<f4", (8, HERO_MF)), ("em", "<f4", (8, HERO_MF)), ("eta", "<f4", (8, HERO_MF)), ("splat_value", "<f4", (8, HERO_MF))])


def hero_ext_dtype(mf):
    """dump_ext_t of the dump harness built with -DMF_COUNT=mf"""
    return np.dtype([("lambda", "<f4", mf), ("throughput", "<f4", (8, mf)), ("pdf", "<f4", (8, mf)), ("rd", "<f4", (8, mf)),
                     ("rg", "<f4", (8, mf)), ("em", "<f4", (8, mf)), ("eta", "<f4", (8, mf)), ("splat_value", "<f4", (8, mf))])


def oracle_hero_records(scene, first, count, fb=None, mf=4):
    """hero wavelengths (MF_COUNT = 4, or 8: the AVX build): (records of the hero component, all mf components of every spectral quantity)"""
    pkg = load_pkg()
    out = np.zeros(count, dtype=pkg.record_dtype())
    ext = np.zeros(count, dtype=HERO_EXT if mf == 4 else hero_ext_dtype(mf))
    if mf == 4:
        oracle_lib().oracle_hero_trace(scene.desc_ptr, first, count, out.ctypes.data, ext.ctypes.data, None if fb is None else fb.ctypes.data, None)
    else:
        oracle_lib().oracle_hero_trace_n(scene.desc_ptr, mf, first, count, out.ctypes.data, ext.ctypes.data, None if fb is None else fb.ctypes.data, None)
    return out, ext


def oracle_intersect(scene, pos, direction, ignore_primid=None, max_dist=None):
    """closest hits of caller-supplied rays by the oracle's accel_intersect; returns (hits, counters)"""
    ray_dt = np.dtype({"names": ["pos", "dir", "ignore", "max_dist"], "formats": [("<f4", 3), ("<f4", 3), "<u8", "<f4"],
                       "offsets": [0, 12, 24, 32], "itemsize": 40})
    hit_dt = np.dtype({"names": ["prim", "dist", "u", "v"], "formats": ["<u8", "<f4", "<f4", "<f4"], "offsets": [0, 8, 12, 16], "itemsize": 24})
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
    n = len(pos)
    rays = np.zeros(n, dtype=ray_dt)
    rays["pos"] = pos
    rays["dir"] = np.asarray(direction, dtype=np.float32).reshape(-1, 3)
    rays["ignore"] = 0xffffffffffffffff if ignore_primid is None else ignore_primid
    rays["max_dist"] = np.float32(3.4028234663852886e38) if max_dist is None else max_dist
    out = np.zeros(n, dtype=hit_dt)
    cnt = (C.c_uint64 * 8)()
    oracle_lib().oracle_intersect(scene.desc_ptr, rays.ctypes.data, n, out.ctypes.data, cnt)
    return out, list(cnt)


def oracle_render(scene, first, count, threads=1):
    fb = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    cnt = (C.c_uint64 * 8)()
    secs = oracle_lib().oracle_render(scene.desc_ptr, first, count, fb.ctypes.data, threads, cnt)
    return fb, list(cnt), secs


def oracle_render_tiles(scene, first_frame, frames, member=0, members=1, threads=1):
    """the member's tiles of the frames, pixels from path indices (oracle.h); leaves that mode ON in the oracle: see oracle_pixels()"""
    fb = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    cnt = (C.c_uint64 * 8)()
    secs = oracle_lib().oracle_render_tiles(scene.desc_ptr, first_frame, frames, member, members, fb.ctypes.data, threads, cnt)
    return fb, list(cnt), secs


class oracle_pixels:
    """with oracle_pixels(mode): ... -- the oracle takes every path's pixel from its index inside (1: row by row, gi.c:88-95; 2: the scattered order of
    MI_PIXELS_SCATTERED), samples it again outside"""
    def __init__(self, mode=2):
        self.mode = mode
    def __enter__(self):
        oracle_lib().oracle_set_pixels_from_index(self.mode)
    def __exit__(self, *a):
        oracle_lib().oracle_set_pixels_from_index(0)


def golden_coeffs():
    """rgb -> (coeff[3], mul) exactly as the reference's LUT yields them (tests/golden/rgb2spec_coeffs.json)."""
    with open(GOLDEN / "rgb2spec_coeffs.json") as f:
        return [json.loads(l) for l in f if l.strip()]


def make_scene(path=SCENE_0010, inject=None, **kw):
    """pkg.Scene(path): the scenes under scenes/ carry the reference table's RGB -> spectrum coefficients for their colours
    (test.rgb2spec, written by tests/golden/make_rgb2spec_cache.py, applied by the host loader), so the shipped default IS
    the reference's init-time constants -- nothing is injected by the tests any more (`inject` is accepted and ignored)."""
    pkg = load_pkg()
    s = pkg.Scene(path, **kw)
    _scene_paths[id(s)] = path
    return s


_scene_paths = {}


def scene_path_of(scene):
    return _scene_paths[id(scene)]
