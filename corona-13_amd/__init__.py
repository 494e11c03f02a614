"""corona-13_amd -- MI355X-native backend for the pt/ptdl hot path of hanatos/corona-13.

The product is two native libraries behind a C ABI (include/corona_mi.h):

  host/libcorona_host.so   plain C: .nra2/.geo/.cam loaders, QBVH build, emitter CDF, PFM output
  csrc/libcorona_mi.so     HIP (gfx950): the path tracing kernels + the mi_* entry points

This Python module is only a ctypes view of that ABI for tests, bench.py and multi-GPU glue
(torch.distributed owns the RCCL reduce of the framebuffer). It contains no rendering logic
and no CPU fallback: if libcorona_mi.so is missing, `Backend()` raises.

The directory name contains '-' and '.', so import it through `load_package()` in
__graft_entry__.py (importlib) under the module name `corona13_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
HOST_LIB = Path(os.environ.get("CORONA_HOST_LIB", PKG_DIR / "host" / "libcorona_host.so"))   # override: the sanitizer build (make sanitize)
MI_LIB = Path(os.environ.get("CORONA_MI_LIB", PKG_DIR / "csrc" / "libcorona_mi.so"))   # override: kernel-variant experiments only

MI_PIXELS_SAMPLED, MI_PIXELS_FROM_INDEX = 0, 1
MI_SAMPLER_PT, MI_SAMPLER_PTDL = 0, 1
MI_POINTS_RAND, MI_POINTS_HALTON = 0, 1
MI_TRAVERSAL_EXACT, MI_TRAVERSAL_FAST = 0, 1
MI_REC_MAX_VERTS, MI_REC_MAX_SPLATS = 8, 8
MI_NODE_LEAF = 1 << 63          # mi_node.child: leaf link = MI_NODE_LEAF | first_prim << 5 | count (corona_mi.h)


# ---------------------------------------------------------------- ctypes mirrors of corona_mi.h
class MiVtxidx(C.Structure):
    _fields_ = [("v", C.c_uint32), ("uv", C.c_uint32)]


class MiVtx(C.Structure):
    _fields_ = [("v", C.c_float * 3), ("n", C.c_uint32)]


class MiShape(C.Structure):
    _fields_ = [("material", C.c_int32), ("num_prims", C.c_uint32),
                ("vtxidx_base", C.c_uint32), ("vtx_base", C.c_uint32)]


class MiNode(C.Structure):
    _fields_ = [("aabb", (C.c_float * 4) * 6), ("child", C.c_uint64 * 4),
                ("axis0", C.c_int32), ("axis00", C.c_int32), ("axis01", C.c_int32), ("parent", C.c_int32)]


class MiNodeAabb(C.Structure):
    _fields_ = [("aabb", (C.c_float * 4) * 6)]


class MiShadeOp(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("slot", C.c_uint32), ("coeff", C.c_float * 3),
                ("mul", C.c_float), ("roughness", C.c_float), ("pad", C.c_uint32)]


class MiMaterial(C.Structure):
    _fields_ = [("bsdf", C.c_uint32), ("num_ops", C.c_uint32), ("op", MiShadeOp * 4), ("param", C.c_float * 4),
                ("mean_cos", C.c_float), ("interior", C.c_int32)]


class MiCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("a", C.c_float * 3), ("b", C.c_float * 3), ("n", C.c_float * 3),
                ("focus", C.c_float), ("focal_length", C.c_float), ("film_width", C.c_float),
                ("film_height", C.c_float), ("f_stop", C.c_float), ("exposure_time", C.c_float),
                ("iso", C.c_float), ("time_scale", C.c_float),
                ("moving", C.c_uint32), ("pos_t1", C.c_float * 3), ("orient", C.c_float * 4), ("orient_t1", C.c_float * 4)]


class MiLights(C.Structure):
    _fields_ = [("num_prims", C.c_uint32), ("primid", C.POINTER(C.c_uint64)), ("cdf", C.POINTER(C.c_float)),
                ("L", C.POINTER(C.c_float)), ("p_sky", C.c_float), ("p_geo", C.c_float), ("p_vol", C.c_float)]


class MiSceneDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("max_verts", C.c_uint32), ("sampler", C.c_uint32),
                ("frame", C.c_uint64),
                ("num_nodes", C.c_uint32), ("nodes", C.POINTER(MiNode)), ("aabb", C.c_float * 6),
                ("num_prims", C.c_uint64), ("primid", C.POINTER(C.c_uint64)),
                ("num_shapes", C.c_uint32), ("shapes", C.POINTER(MiShape)),
                ("num_vtxidx", C.c_uint64), ("vtxidx", C.POINTER(MiVtxidx)),
                ("num_vtx", C.c_uint64), ("vtx", C.POINTER(MiVtx)),
                ("num_materials", C.c_uint32), ("materials", C.POINTER(MiMaterial)),
                ("lights", MiLights), ("cam", MiCamera),
                ("cie_xyz", C.POINTER(C.c_float)), ("checker", C.POINTER(C.c_float)), ("metal_ior", C.POINTER(C.c_float)),
                ("pointsampler", C.c_uint32), ("exterior", C.c_uint32), ("nodes_t1", C.POINTER(MiNodeAabb))]


class MiPathVertex(C.Structure):
    _fields_ = [("prim", C.c_uint64), ("dist", C.c_float), ("x", C.c_float * 3), ("n", C.c_float * 3),
                ("gn", C.c_float * 3), ("omega", C.c_float * 3), ("mode", C.c_uint32), ("flags", C.c_uint32),
                ("throughput", C.c_float), ("pdf", C.c_float), ("u", C.c_float), ("v", C.c_float),
                ("rd", C.c_float), ("rg", C.c_float), ("em", C.c_float), ("roughness", C.c_float),
                ("eta", C.c_float), ("shader", C.c_int32)]


class MiPathSplat(C.Structure):
    _fields_ = [("length", C.c_int32), ("tech", C.c_int32), ("value", C.c_float), ("col", C.c_float * 3)]


class MiPathRecord(C.Structure):
    _fields_ = [("index", C.c_uint64), ("pixel_i", C.c_float), ("pixel_j", C.c_float), ("lambda_", C.c_float),
                ("time", C.c_float), ("scramble", C.c_float), ("throughput", C.c_float),
                ("length", C.c_int32), ("num_splats", C.c_int32),
                ("splat", MiPathSplat * MI_REC_MAX_SPLATS), ("v", MiPathVertex * MI_REC_MAX_VERTS)]


class MiBsdfTest(C.Structure):
    _fields_ = [("bsdf", C.c_uint32), ("param", C.c_float * 2), ("roughness", C.c_float), ("reflect", C.c_uint32), ("count", C.c_uint32),
                ("lambda_", C.c_float), ("size", C.c_uint32), ("spp", C.c_uint32)]


class ChOptions(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_verts", C.c_uint32), ("sampler", C.c_uint32),
                ("frame", C.c_uint64), ("cam_file", C.c_char_p), ("rgb2spec_lut", C.c_char_p),
                ("data_dir", C.c_char_p), ("iso", C.c_float), ("build_threads", C.c_int), ("verbose", C.c_int),
                ("pointsampler", C.c_uint32)]


def record_dtype():
    """numpy dtype of mi_path_record (== the reference dump harness record)."""
    import numpy as np
    vert = np.dtype([("prim", "<u8"), ("dist", "<f4"), ("x", "<f4", 3), ("n", "<f4", 3), ("gn", "<f4", 3),
                     ("omega", "<f4", 3), ("mode", "<u4"), ("flags", "<u4"), ("throughput", "<f4"), ("pdf", "<f4"),
                     ("u", "<f4"), ("v", "<f4"), ("rd", "<f4"), ("rg", "<f4"), ("em", "<f4"), ("roughness", "<f4"),
                     ("eta", "<f4"), ("shader", "<i4")], align=True)
    splat = np.dtype([("length", "<i4"), ("tech", "<i4"), ("value", "<f4"), ("col", "<f4", 3)])
    rec = np.dtype([("index", "<u8"), ("pixel_i", "<f4"), ("pixel_j", "<f4"), ("lambda", "<f4"), ("time", "<f4"),
                    ("scramble", "<f4"), ("throughput", "<f4"), ("length", "<i4"), ("num_splats", "<i4"),
                    ("splat", splat, MI_REC_MAX_SPLATS), ("v", vert, MI_REC_MAX_VERTS)], align=True)
    assert rec.itemsize == C.sizeof(MiPathRecord), (rec.itemsize, C.sizeof(MiPathRecord))
    return rec


def shard_range(first, count, rank, world):
    """Contiguous block of path indices [first, first+count) owned by `rank` of `world` (paths are independent;
    SURVEY 8(e): partition by sample index range). Remainder indices go to the lowest ranks."""
    base, rem = divmod(count, world)
    my = base + (1 if rank < rem else 0)
    start = first + rank * base + min(rank, rem)
    return start, my


class FrameReducer:
    """Double-buffered framebuffer all-reduce for the multi-GPU path (one process per GPU, paths sharded by index range,
    SURVEY 8(e)): the reduce of frame k runs while frame k+1 is rendered into the other buffer, so the exchange
    (11.3 MB at 1280x736, one ring all-reduce over xGMI) is off the critical path except for the last frame.
    `dist` is torch.distributed (nccl = RCCL on the GPU box, gloo in the CPU tests); buffers are torch tensors."""

    def __init__(self, buffers, dist):
        assert len(buffers) == 2
        self.buffers, self.dist, self.pending = list(buffers), dist, [None, None]

    def begin(self, k):
        """the buffer frame k renders into: its previous reduce (frame k-2) has completed, it is cleared"""
        b = k & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self.buffers[b].zero_()
        return self.buffers[b]

    def end(self, k):
        """frame k has been enqueued: start its all-reduce (asynchronous; ordered after the render on the device)"""
        b = k & 1
        self.pending[b] = self.dist.all_reduce(self.buffers[b], op=self.dist.ReduceOp.SUM, async_op=True)

    def finished(self, k):
        """wait for frame k's reduce and return its buffer (the sum over all ranks)"""
        b = k & 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        return self.buffers[b]

    def drain(self):
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None


# ---------------------------------------------------------------- host library
_host = None


def host_lib():
    global _host
    if _host is None:
        if not HOST_LIB.exists():
            raise RuntimeError(f"{HOST_LIB} missing: run __graft_entry__.build() (make -C corona-13_amd)")
        os.environ.setdefault("CORONA_MI_DATA", str(PKG_DIR / "data"))
        h = C.CDLL(str(HOST_LIB))
        h.ch_scene_load.argtypes = [C.c_char_p, C.POINTER(ChOptions), C.POINTER(C.c_void_p)]
        h.ch_scene_load.restype = C.c_int
        h.ch_scene_desc.argtypes = [C.c_void_p]
        h.ch_scene_desc.restype = C.POINTER(MiSceneDesc)
        h.ch_scene_set_color_coeff.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float]
        h.ch_scene_set_color_coeff.restype = C.c_int
        h.ch_scene_num_shaders.argtypes = [C.c_void_p]
        h.ch_scene_shader_name.argtypes = [C.c_void_p, C.c_int]
        h.ch_scene_shader_name.restype = C.c_char_p
        h.ch_scene_free.argtypes = [C.c_void_p]
        h.ch_scene_free.restype = None
        h.ch_scene_gain.argtypes = [C.c_void_p, C.c_uint64]
        h.ch_scene_gain.restype = C.c_float
        h.ch_pfm_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float]
        h.ch_rgb_to_coeff.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_char_p]
        h.ch_rgb_to_coeff.restype = C.c_float
        _host = h
    return _host


class Scene:
    """A scene loaded by the host library (owns the mi_scene_desc)."""

    def __init__(self, nra2, width=1024, height=576, max_verts=32, sampler=MI_SAMPLER_PT, frame=1,
                 rgb2spec_lut=None, verbose=0, pointsampler=MI_POINTS_RAND):
        h = host_lib()
        opt = ChOptions(width=width, height=height, max_verts=max_verts, sampler=sampler, frame=frame,
                        cam_file=None, rgb2spec_lut=(str(rgb2spec_lut).encode() if rgb2spec_lut else None),
                        data_dir=str(PKG_DIR / "data").encode(), iso=0.0, build_threads=0, verbose=verbose,
                        pointsampler=pointsampler)
        self._ptr = C.c_void_p()
        err = h.ch_scene_load(str(nra2).encode(), C.byref(opt), C.byref(self._ptr))
        if err:
            raise RuntimeError(f"ch_scene_load({nra2}) failed: {err}")
        self._h = h

    @property
    def desc(self):
        return self._h.ch_scene_desc(self._ptr).contents

    @property
    def desc_ptr(self):
        return self._h.ch_scene_desc(self._ptr)

    @property
    def width(self):
        return self.desc.width

    @property
    def height(self):
        return self.desc.height

    def gain(self, spp):
        return self._h.ch_scene_gain(self._ptr, spp)

    def shader_names(self):
        return [self._h.ch_scene_shader_name(self._ptr, i).decode() for i in range(self._h.ch_scene_num_shaders(self._ptr))]

    def set_color_coeff(self, shader_id, coeff, mul):
        arr = (C.c_float * 3)(*coeff)
        err = self._h.ch_scene_set_color_coeff(self._ptr, shader_id, arr, mul)
        if err:
            raise RuntimeError(f"set_color_coeff({shader_id}) failed: {err}")

    def close(self):
        if self._ptr:
            self._h.ch_scene_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- HIP backend (C ABI)
_mi = None


def mi_lib():
    """Load libcorona_mi.so. There is deliberately no fallback: missing extension == hard error."""
    global _mi
    if _mi is None:
        if not MI_LIB.exists():
            raise RuntimeError(f"{MI_LIB} missing: the HIP extension is not built (run __graft_entry__.build())")
        m = C.CDLL(str(MI_LIB))
        m.mi_init.argtypes = [C.c_int]
        m.mi_scene_create.argtypes = [C.POINTER(MiSceneDesc), C.POINTER(C.c_void_p)]
        m.mi_scene_set_framebuffer.argtypes = [C.c_void_p, C.c_void_p]
        m.mi_scene_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        m.mi_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        m.mi_sync.argtypes = [C.c_void_p]
        m.mi_fb_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        m.mi_fb_clear.argtypes = [C.c_void_p]
        m.mi_fb_device_ptr.argtypes = [C.c_void_p]
        m.mi_fb_device_ptr.restype = C.c_void_p
        m.mi_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        m.mi_scene_set_counters.argtypes = [C.c_void_p, C.c_int]
        m.mi_scene_set_traversal.argtypes = [C.c_void_p, C.c_int]
        m.mi_scene_get_traversal.argtypes = [C.c_void_p]
        m.mi_scene_set_metal_reference.argtypes = [C.c_void_p, C.c_int]
        m.mi_trace_paths.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        m.mi_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        m.mi_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        m.mi_last_kernel_launches.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        m.mi_scene_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        try:
            m.mi_scene_set_pixels.argtypes = [C.c_void_p, C.c_int]
            m.mi_render_tiles.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
            m.mi_group_render_tiles.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        except AttributeError:
            if "CORONA_MI_LIB" not in os.environ:
                raise
        try:
            m.mi_scene_set_wavelengths.argtypes = [C.c_void_p, C.c_int]
            m.mi_trace_paths_hero.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        except AttributeError:
            if "CORONA_MI_LIB" not in os.environ:
                raise
        try:
            m.mi_scene_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
            m.mi_scene_lds_nodes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        except AttributeError:
            if "CORONA_MI_LIB" not in os.environ:      # only a kernel-variant library of an older round (same-box A/B) may lack these two getters
                raise
        m.mi_scene_destroy.argtypes = [C.c_void_p]
        m.mi_scene_destroy.restype = None
        m.mi_shutdown.restype = None
        m.mi_last_error.restype = C.c_char_p
        m.mi_bsdf_test_run.argtypes = [C.c_void_p, C.POINTER(MiBsdfTest), C.POINTER(C.c_double)]
        m.mi_current_device.restype = C.c_int
        m.mi_group_create.argtypes = [C.POINTER(MiSceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        m.mi_group_size.argtypes = [C.c_void_p]
        m.mi_group_scene.argtypes = [C.c_void_p, C.c_int]
        m.mi_group_scene.restype = C.c_void_p
        m.mi_group_uses_rccl.argtypes = [C.c_void_p]
        m.mi_group_render.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        m.mi_group_fb_reduce.argtypes = [C.c_void_p]
        m.mi_group_fb_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        m.mi_group_fb_clear.argtypes = [C.c_void_p]
        m.mi_group_sync.argtypes = [C.c_void_p]
        m.mi_group_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        m.mi_group_destroy.argtypes = [C.c_void_p]
        m.mi_group_destroy.restype = None
        _mi = m
    return _mi


MI_SYMBOLS = ["mi_init", "mi_scene_create", "mi_scene_set_framebuffer", "mi_scene_set_stream", "mi_render",
              "mi_sync", "mi_fb_read", "mi_fb_clear", "mi_fb_device_ptr", "mi_counters", "mi_scene_set_counters", "mi_scene_set_traversal", "mi_scene_get_traversal", "mi_scene_set_metal_reference", "mi_trace_paths", "mi_intersect", "mi_plan_launches",
              "mi_last_kernel_ms", "mi_last_kernel_launches", "mi_scene_stats", "mi_scene_lds_nodes", "mi_scene_kernel_name", "mi_scene_set_pixels", "mi_render_tiles", "mi_group_render_tiles", "mi_scene_set_wavelengths", "mi_trace_paths_hero", "mi_scene_destroy", "mi_shutdown", "mi_last_error", "mi_current_device", "mi_bsdf_test_run",
              "mi_group_create", "mi_group_size", "mi_group_scene", "mi_group_uses_rccl", "mi_group_render", "mi_group_fb_reduce", "mi_group_fb_read",
              "mi_group_fb_clear", "mi_group_sync", "mi_group_counters", "mi_group_destroy"]


MI_WAVELENGTHS_HERO = 4


def hero_ext_dtype():
    """numpy dtype of mi_hero_ext (include/corona_mi.h)"""
    import numpy as np
    n = MI_WAVELENGTHS_HERO
    return np.dtype([("lambda", "<f4", n), ("throughput", "<f4", (8, n)), ("pdf", "<f4", (8, n)), ("rd", "<f4", (8, n)), ("rg", "<f4", (8, n)),
                     ("em", "<f4", (8, n)), ("eta", "<f4", (8, n)), ("splat_value", "<f4", (8, n))])


def ray_dtypes():
    """numpy dtypes of mi_ray and mi_hit (32 B each)."""
    import numpy as np
    ray = np.dtype([("pos", "<f4", 3), ("dir", "<f4", 3), ("ignore", "<u4"), ("max_dist", "<f4")])
    hit = np.dtype([("primid", "<u8"), ("prim", "<u4"), ("dist", "<f4"), ("u", "<f4"), ("v", "<f4"), ("pad", "<u4", 2)])
    return ray, hit


class Backend:
    """Device-resident scene on one MI355X, driven through the C ABI."""

    def __init__(self, scene: Scene, device: int = -1, device_build: bool = False, counters: bool = True, traversal: str | None = None):
        """device_build: hand the scene over WITHOUT the host-built tree (mi_scene_desc.nodes = NULL); the backend then
        builds its own 4-wide BVH on the GPU (csrc/mi_build.h).
        counters: render with the counting kernels (mi_scene_set_counters) so that counters() reports the traversal work --
        this Python view is the test / measurement harness, so they are on unless asked otherwise; the C ABI's default is off."""
        self.m = mi_lib()
        self._check(self.m.mi_init(device), "mi_init")
        self._ptr = C.c_void_p()
        desc_ptr = scene.desc_ptr
        if device_build:
            self._desc = MiSceneDesc()
            C.memmove(C.byref(self._desc), scene.desc_ptr, C.sizeof(MiSceneDesc))
            self._desc.nodes = None
            self._desc.num_nodes = 0
            desc_ptr = C.pointer(self._desc)
        self._check(self.m.mi_scene_create(desc_ptr, C.byref(self._ptr)), "mi_scene_create")
        self.scene = scene
        self.set_counters(counters)
        if traversal is not None:
            self.set_traversal(traversal)

    def set_counters(self, enable):
        self._check(self.m.mi_scene_set_counters(self._ptr, 1 if enable else 0), "mi_scene_set_counters")

    def set_traversal(self, mode):
        """'exact': the reference's order of operations per ray (counters equal its -DACCEL_DEBUG totals); 'fast' (the library's
        default): leaves put aside while the lane descends on -- same hits, other work counters (corona_mi.h)"""
        self._check(self.m.mi_scene_set_traversal(self._ptr, {"exact": 0, "fast": 1}[mode]), "mi_scene_set_traversal")

    def traversal(self):
        """the traversal mode the next render uses ('exact' / 'fast')"""
        return "fast" if self.m.mi_scene_get_traversal(self._ptr) == 1 else "exact"

    def set_metal_reference(self, enable):
        """end the metal samples the reference BUILD's NaN ends (corona_mi.h: mi_scene_set_metal_reference)"""
        self._check(self.m.mi_scene_set_metal_reference(self._ptr, 1 if enable else 0), "mi_scene_set_metal_reference")

    def _check(self, err, what):
        if err:
            raise RuntimeError(f"{what} failed ({err}): {self.m.mi_last_error().decode()}")

    def set_framebuffer(self, device_ptr):
        self._check(self.m.mi_scene_set_framebuffer(self._ptr, C.c_void_p(device_ptr)), "mi_scene_set_framebuffer")

    def set_stream(self, stream_handle):
        """launch on this HIP stream (torch: torch.cuda.current_stream().cuda_stream). Handle 0 is the device's default
        stream -- torch's current stream unless told otherwise -- and is passed as MI_STREAM_DEFAULT: a NULL pointer would
        select the backend's own stream, which is not ordered with torch's clears and collectives."""
        ptr = C.c_void_p(stream_handle) if stream_handle else C.c_void_p(-1)
        self._check(self.m.mi_scene_set_stream(self._ptr, ptr), "mi_scene_set_stream")

    def render(self, first, count):
        self._check(self.m.mi_render(self._ptr, first, count), "mi_render")

    def set_pixels(self, from_index: bool):
        """True: path i starts inside pixel (i mod W H) (MI_PIXELS_FROM_INDEX, the hook of the reference's tiled branch of render_sample_path); False: sampled"""
        self._check(self.m.mi_scene_set_pixels(self._ptr, MI_PIXELS_FROM_INDEX if from_index else MI_PIXELS_SAMPLED), "mi_scene_set_pixels")

    def render_tiles(self, first_frame, frames, member=0, members=1):
        """the paths of frames [first_frame, first_frame + frames) whose pixel lies in a 32 x 32 tile t = member (mod members)"""
        self._check(self.m.mi_render_tiles(self._ptr, first_frame, frames, member, members), "mi_render_tiles")

    def sync(self):
        self._check(self.m.mi_sync(self._ptr), "mi_sync")

    def fb_clear(self):
        self._check(self.m.mi_fb_clear(self._ptr), "mi_fb_clear")

    def fb_read(self, accumulate_into=None):
        import numpy as np
        w, h = self.scene.width, self.scene.height
        if accumulate_into is None:
            out = np.zeros((h, w, 3), dtype=np.float32)
            self._check(self.m.mi_fb_read(self._ptr, out.ctypes.data, 0), "mi_fb_read")
            return out
        self._check(self.m.mi_fb_read(self._ptr, accumulate_into.ctypes.data, 1), "mi_fb_read")
        return accumulate_into

    def counters(self):
        arr = (C.c_uint64 * 8)()
        self._check(self.m.mi_counters(self._ptr, arr), "mi_counters")
        return list(arr)

    def trace_paths(self, first, count):
        import numpy as np
        out = np.zeros(count, dtype=record_dtype())
        self._check(self.m.mi_trace_paths(self._ptr, first, count, out.ctypes.data), "mi_trace_paths")
        return out

    def set_wavelengths(self, count):
        """4 (MI_WAVELENGTHS_HERO): four wavelengths per path, the reference's MF_COUNT=4 build (mi_scene_set_wavelengths); 1: back to one"""
        self._check(self.m.mi_scene_set_wavelengths(self._ptr, int(count)), "mi_scene_set_wavelengths")

    def trace_paths_hero(self, first, count):
        """(records with component 0 of every spectral quantity, mi_hero_ext with all four) of paths [first, first + count)"""
        import numpy as np
        out = np.zeros(count, dtype=record_dtype())
        ext = np.zeros(count, dtype=hero_ext_dtype())
        self._check(self.m.mi_trace_paths_hero(self._ptr, first, count, out.ctypes.data, ext.ctypes.data), "mi_trace_paths_hero")
        return out, ext

    def intersect(self, pos, direction, ignore=None, max_dist=None):
        """closest hits of caller-supplied rays (test hook, mi_intersect): returns a structured array (primid, prim, dist, u, v)"""
        import numpy as np
        RAY_DTYPE, HIT_DTYPE = ray_dtypes()
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float32).reshape(-1, 3)
        n = len(pos)
        rays = np.zeros(n, dtype=RAY_DTYPE)
        rays["pos"] = pos; rays["dir"] = direction
        rays["ignore"] = 0xffffffff if ignore is None else ignore
        rays["max_dist"] = np.float32(3.4028234663852886e38) if max_dist is None else max_dist
        out = np.zeros(n, dtype=HIT_DTYPE)
        self._check(self.m.mi_intersect(self._ptr, rays.ctypes.data, n, out.ctypes.data), "mi_intersect")
        return out

    def bsdf_test(self, bsdf, param, roughness, reflect, count=4, lambda_=525.0, size=512, spp=8):
        """the reference's BSDF battle test on the device (mi_bsdf_test_run): rows of (ebsdf, bsdf, epdf, pdf), one per incidence angle"""
        import numpy as np
        kind = {"diffuse": 0, "dielectric": 1, "metal": 2}[bsdf]
        t = MiBsdfTest(bsdf=kind, param=(C.c_float * 2)(*param), roughness=roughness, reflect=reflect, count=count, lambda_=lambda_, size=size, spp=spp)
        out = np.zeros((count, 4), dtype=np.float64)
        self._check(self.m.mi_bsdf_test_run(self._ptr, C.byref(t), out.ctypes.data_as(C.POINTER(C.c_double))), "mi_bsdf_test_run")
        return out

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.m.mi_last_kernel_ms(self._ptr, C.byref(ms)), "mi_last_kernel_ms")
        return ms.value

    def last_kernel_launches(self):
        n = C.c_uint64()
        self._check(self.m.mi_last_kernel_launches(self._ptr, C.byref(n)), "mi_last_kernel_launches")
        return n.value

    def stats(self):
        """dict: 4-wide nodes, tree staged in LDS, stack entries needed, tree built on the device"""
        out = (C.c_uint32 * 4)()
        self._check(self.m.mi_scene_stats(self._ptr, out), "mi_scene_stats")
        return {"nodes": out[0], "nodes_in_lds": bool(out[1]), "stack_need": out[2], "device_built": bool(out[3])}

    def nodes_in_lds(self):
        return self.stats()["nodes_in_lds"]

    def lds_nodes(self):
        """nodes of the tree (breadth first from the root) staged in LDS (mi_scene_lds_nodes)"""
        out = C.c_uint32(0)
        self._check(self.m.mi_scene_lds_nodes(self._ptr, C.byref(out)), "mi_scene_lds_nodes")
        return int(out.value)

    def kernel_name(self):
        """the instantiation the next render() launches, as rocprofv3 prints it (mi_scene_kernel_name)"""
        buf = C.create_string_buffer(256)
        if "CORONA_MI_LIB" in os.environ and not hasattr(self.m, "mi_scene_kernel_name"):
            return "mi_path_kernel<a library of an earlier round>"       # same-box A/B against an older build only
        self._check(self.m.mi_scene_kernel_name(self._ptr, buf, len(buf)), "mi_scene_kernel_name")
        return buf.value.decode()

    def close(self):
        if self._ptr:
            self.m.mi_scene_destroy(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """Several GPUs of one node behind the C ABI, driven by this one host thread (mi_group_*, corona_mi.h): one copy of the scene per
    device, path indices split by contiguous ranges, framebuffers added up on member 0 (RCCL ncclReduce over xGMI, or peer copies
    when a device is named twice). `devices` is a list of device indices."""

    def __init__(self, scene: Scene, devices, traversal: str | None = None, counters: bool = False):
        self.m = mi_lib()
        self._ptr = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        err = self.m.mi_group_create(scene.desc_ptr, arr, len(devices), C.byref(self._ptr))
        if err:
            raise RuntimeError(f"mi_group_create failed ({err}): {self.m.mi_last_error().decode()}")
        self.scene, self.n = scene, len(devices)
        for k in range(self.n):
            member = C.c_void_p(self.m.mi_group_scene(self._ptr, k))
            if traversal is not None:
                self._check(self.m.mi_scene_set_traversal(member, {"exact": 0, "fast": 1}[traversal]), "mi_scene_set_traversal")
            self._check(self.m.mi_scene_set_counters(member, 1 if counters else 0), "mi_scene_set_counters")

    def _check(self, err, what):
        if err:
            raise RuntimeError(f"{what} failed ({err}): {self.m.mi_last_error().decode()}")

    def uses_rccl(self):
        return bool(self.m.mi_group_uses_rccl(self._ptr))

    def set_wavelengths(self, count):
        """every member: mi_scene_set_wavelengths (4 = hero wavelengths)"""
        for k in range(self.n):
            self._check(self.m.mi_scene_set_wavelengths(C.c_void_p(self.m.mi_group_scene(self._ptr, k)), int(count)), "mi_scene_set_wavelengths")

    def render(self, first, count):
        self._check(self.m.mi_group_render(self._ptr, first, count), "mi_group_render")

    def render_tiles(self, first_frame, frames):
        """member k renders the tiles t = k (mod n) of every frame (mi_group_render_tiles)"""
        self._check(self.m.mi_group_render_tiles(self._ptr, first_frame, frames), "mi_group_render_tiles")

    def reduce(self):
        self._check(self.m.mi_group_fb_reduce(self._ptr), "mi_group_fb_reduce")

    def sync(self):
        self._check(self.m.mi_group_sync(self._ptr), "mi_group_sync")

    def fb_clear(self):
        self._check(self.m.mi_group_fb_clear(self._ptr), "mi_group_fb_clear")

    def fb_read(self, accumulate_into=None):
        import numpy as np
        if accumulate_into is None:
            out = np.zeros((self.scene.height, self.scene.width, 3), dtype=np.float32)
            self._check(self.m.mi_group_fb_read(self._ptr, out.ctypes.data, 0), "mi_group_fb_read")
            return out
        self._check(self.m.mi_group_fb_read(self._ptr, accumulate_into.ctypes.data, 1), "mi_group_fb_read")
        return accumulate_into

    def counters(self):
        arr = (C.c_uint64 * 8)()
        self._check(self.m.mi_group_counters(self._ptr, arr), "mi_group_counters")
        return list(arr)

    def member_kernel_ms(self, k):
        ms = C.c_float()
        self._check(self.m.mi_last_kernel_ms(C.c_void_p(self.m.mi_group_scene(self._ptr, k)), C.byref(ms)), "mi_last_kernel_ms")
        return ms.value

    def close(self):
        if self._ptr:
            self.m.mi_group_destroy(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
