#!/usr/bin/env python3
"""bench.py -- Msamples/s of the pt hot path on regression/0010_pt at 1280x720, 64 spp, max depth 8
(BASELINE.json config 2) on N MI355X of one node.

A "step" is one whole frame: 64 spp x 1280x736 (the film is padded to multiples of 32 like the
reference, src/view.c:294-296) = 60 293 120 camera paths traced by the HIP kernel into a
device-resident framebuffer, INCLUDING the framebuffer reduce across ranks (RCCL all-reduce over
xGMI, the only collective of this path) -- the timing window of the reference's "elapsed wallclock
prog" (src/view.c:634,687-688). Scene upload, BVH build and file output are outside, as in the reference.
Inputs (scene, tables) are resident in HBM when the timed region starts.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL). Paths are independent, so every
rank renders its own contiguous range of path indices with no data-path collective; scaling is WEAK
(each GPU renders a full 64-spp frame's worth of distinct indices, the job is N x 64 spp).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

W, H, SPP, MAX_VERTS = 1280, 720, 64, 8
# BASELINE.json configs; the metric is quoted on configs[1], which is what the driver's plain `bench.py` run measures.
# The others can be timed with --config (development / DESIGN.md table); their CPU baseline leg is not run.
CONFIGS = {
    "cfg1": dict(scene="0010_pt", sampler="pt", w=256, h=256, spp=4, mv=4, name="configs[0]: regression/0010_pt, pt, 256x256, 4 spp, max depth 4"),
    "cfg2": dict(scene="0010_pt", sampler="pt", w=W, h=H, spp=SPP, mv=MAX_VERTS,
                 name="configs[1]: regression/0010_pt test.nra2, pt sampler, 1280x720 (padded 1280x736), 64 spp, max depth 8"),
    "cfg3": dict(scene="0010_pt", sampler="ptdl", w=1280, h=720, spp=64, mv=8, name="configs[2]: regression/0011_ptdl (0010 scene, ptdl sampler), 1280x720, 64 spp"),
    "cfg4": dict(scene="0052_rough", sampler="pt", w=1280, h=720, spp=256, mv=32, name="configs[3]: regression/0052 parameters (rough dielectric), max depth 32, 1280x720, 256 spp"),
    "cfg5": dict(scene="0010_pt", sampler="pt", w=3840, h=2160, spp=128, mv=8, name="configs[4]: regression/0010_pt at 3840x2160, 1024 spp over 8 GPUs = 128 spp per GPU"),
    # not BASELINE.json configurations: the SURVEY 8(f) row 3 scenes (participating media), same film and depth as configs[1]
    "media": dict(scene="0055_media", sampler="pt", w=1280, h=720, spp=64, mv=8, name="0010 scene, scattering medium inside the glass sphere (scenes/0055_media), pt, 1280x720, 64 spp"),
    "media_ptdl": dict(scene="0055_media", sampler="ptdl", w=1280, h=720, spp=64, mv=8, name="scenes/0055_media, ptdl, 1280x720, 64 spp"),
    "mb": dict(scene="0059_mb", sampler="pt", w=1280, h=720, spp=64, mv=8, name="0010 scene with a moving backdrop and cylinder cap (scenes/0059_mb, motion blur), pt, 1280x720, 64 spp"),
    "cam_mb": dict(scene="0058_cam_mb", sampler="pt", w=1280, h=720, spp=64, mv=8, name="0010 scene seen by a moving camera (scenes/0058_cam_mb), pt, 1280x720, 64 spp"),
    "fog": dict(scene="0056_fog", sampler="pt", w=1280, h=720, spp=64, mv=8, name="0010 scene in a thin global fog (scenes/0056_fog), pt, 1280x720, 64 spp"),
    "fog_ptdl": dict(scene="0056_fog", sampler="ptdl", w=1280, h=720, spp=64, mv=8, name="scenes/0056_fog, ptdl, 1280x720, 64 spp"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(scene):
    """CPU number reported beside the GPU result, on a bounded sample of the same workload.
    kind "reference": the real corona-13 binary built from /root/reference in the build container
    (oracle/_ref/, shipped as a built artefact), sfmt + rand like regression/0010_pt/config.mk. The reference's
    pool shares one atomic path counter and a CAS framebuffer, so it stops scaling at a few dozen threads: it is
    run with all hardware threads and with 32, and the better one is reported (`cores` = threads of that run).
    kind "port": our CPU restatement (oracle/liboracle.so) if the reference binary is not there."""
    cores = os.cpu_count() or 1
    ref = REPO / "oracle" / "_ref"
    binary = ref / "corona_pt_sfmt_mv8"
    spp = 16
    if binary.exists() and (ref / "data" / "ergb2spec.coeff").exists():
        best = None
        for threads in sorted({cores, min(cores, 32)}):
            work = Path(tempfile.mkdtemp(prefix="corona_cpu_"))
            try:
                shutil.copytree(REPO / "scenes", work / "scenes")
                env = dict(os.environ, LD_LIBRARY_PATH=str(ref / "shaders_mv8"))
                subprocess.run([str(binary), str(work / "scenes" / "0010_pt" / "test.nra2"), "-s", str(spp), "--batch", str(spp),
                                "-w", str(W), "-h", str(H), "-t", str(threads), "-x", "_cpu"], cwd=ref, env=env,
                               capture_output=True, text=True, timeout=600, check=True)
                side = (work / "scenes" / "0010_pt" / "test_cpu_fb00.pfm.txt").read_text()
                secs = float(re.search(r"elapsed wallclock prog ([\d.]+)s", side).group(1))
                n = spp * scene.width * scene.height
                cand = {"value": n / secs / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "reference",
                        "sample": f"{spp} spp of the same 1280x736 frame ({n} paths), reference binary corona_pt_sfmt_mv8, "
                                  f"-t {threads} of {cores} hardware threads, {secs:.2f} s"}
                if best is None or cand["value"] > best["value"]:
                    best = cand
            except Exception as e:          # fall through to the port
                print(f"[bench] reference cpu baseline failed: {e}", file=sys.stderr)
            finally:
                shutil.rmtree(work, ignore_errors=True)
        if best:
            return best
    from helpers import oracle_render
    n = 4 * scene.width * scene.height
    _, _, secs = oracle_render(scene, 0, n, threads=cores)
    return {"value": n / secs / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"4 spp of the same 1280x736 frame ({n} paths), oracle/liboracle.so, {secs:.2f} s"}


def profiled_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/profile.sh ->
    profiles/rNN_pmc_summary.json): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE counts 128-B requests as 64 B
    on gfx950 (MI355X_MICROARCH.md, HBM section); counters are in KiB. None if no profile is committed."""
    files = sorted((REPO / "profiles").glob("r*_pmc_summary.json"))
    if not files:
        return None, None
    d = json.loads(files[-1].read_text())
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None, None
    return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, files[-1].name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--tree", default="reference", choices=["reference", "device"],
                    help="reference: the QBVH the reference's builder makes, handed over through the ABI (default, the drop-in contract); "
                         "device: no tree handed over, the backend builds its own (csrc/mi_build.h)")
    ap.add_argument("--points", default="rand", choices=["rand", "halton"],
                    help="MOD_pointsampler: rand (regression/0010_pt/config.mk, the default) or halton (SURVEY 8(f) row 2)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    from helpers import make_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK/WORLD_SIZE/MASTER_* in the environment) a process group is always formed, also
    # for a single rank, so that the RCCL path can be exercised on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a banner (ROCm version, hostname, library path) on stdout when the communicator comes up; stdout is
        # reserved for the one JSON line, so the banner goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    pkg = load_package()
    cfg = CONFIGS[args.config]
    scene = make_scene(REPO / "scenes" / cfg["scene"] / "test.nra2", width=cfg["w"], height=cfg["h"], max_verts=cfg["mv"],
                       sampler=pkg.MI_SAMPLER_PTDL if cfg["sampler"] == "ptdl" else pkg.MI_SAMPLER_PT,
                       pointsampler=pkg.MI_POINTS_HALTON if args.points == "halton" else pkg.MI_POINTS_RAND)
    be = pkg.Backend(scene, device=local_rank, device_build=args.tree == "device")
    per_frame = cfg["spp"] * scene.width * scene.height
    fb = torch.zeros((scene.height, scene.width, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream()
    be.set_framebuffer(fb.data_ptr())
    be.set_stream(stream.cuda_stream)      # torch's current stream (the default stream, passed as MI_STREAM_DEFAULT): clears and RCCL are ordered with the renders
    # N > 1: two framebuffers, the all-reduce of step k (RCCL over xGMI) overlaps the render of step k+1
    reducer = pkg.FrameReducer([fb, torch.zeros_like(fb)], dist) if use_dist else None

    def barrier():
        if use_dist:
            reducer.drain()                                # every step's reduce is part of the timed region
            dist.barrier()
        torch.cuda.synchronize()

    def step(k):
        # rank r renders its own contiguous block of path indices of "frame" k: no data-path collective
        first, count = pkg.shard_range(k * world * per_frame, world * per_frame, rank, world)
        if use_dist:
            be.set_framebuffer(reducer.begin(k).data_ptr())    # cleared: the reduce works on this step's partial sums only
        be.render(first, count)
        if use_dist:
            reducer.end(k)                                 # framebuffer reduce over xGMI (RCCL), asynchronous

    for k in range(args.warmup):
        step(k)
    barrier()
    kernel_ms = []
    c0 = be.counters()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
        kernel_ms.append(None)
    barrier()
    t1 = time.perf_counter()
    c1 = be.counters()
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel duration with HIP events on the launch stream: re-run K launches un-overlapped, outside the timed region
    durs = []
    for k in range(max(3, min(args.steps, 5))):
        be.render((1000 + k) * per_frame, per_frame)
        be.sync()
        durs.append(be.last_kernel_ms())
    kms = sum(durs) / len(durs)
    dc = [b - a for a, b in zip(c0, c1)]
    paths = dc[4]
    # algorithmic bytes per sample, SURVEY 8(d): 128 B per node visit + 104 B per primitive test + 384 B per splat
    bytes_per_sample = (128.0 * dc[1] + 104.0 * dc[3] + 384.0 * dc[5]) / max(paths, 1)
    achieved = bytes_per_sample * per_frame / (kms * 1e-3) / 1e9

    if rank == 0:
        traffic, traffic_src = profiled_traffic()
        total = args.steps * per_frame * world
        out = {
            "metric": "Msamples/sec (and ms/frame) at 1280x720, 64 spp, regression/0010_pt" if args.config == "cfg2" else "Msamples/sec (and ms/frame), " + args.config,
            "value": total / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: regression/0010_pt scene (6 of 7 shapes, scenes/0010_pt), per-path xorshift128+ seeds",
            "config": {"workload": cfg["name"], "tree": args.tree, "pointsampler": args.points,
                       "paths_per_step_per_gpu": per_frame, "sharding": f"path-index ranges x{world}, framebuffer all-reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic if args.config == "cfg2" else None, "traffic_source": traffic_src if args.config == "cfg2" else None,
                         "kernel": "mi_path_kernel<false,%s,%s,%s,%s,%s> (RECORD, PTDL, NODES_LDS, HALTON, MEDIA, MB)" % ("true" if cfg["sampler"] == "ptdl" else "false", "true" if be.nodes_in_lds() else "false", "true" if args.points == "halton" else "false", "true" if cfg["scene"] in ("0055_media", "0056_fog", "0058_cam_mb", "0059_mb") else "false", "true" if cfg["scene"] == "0059_mb" else "false"), "kernel_ms": kms,
                         "algorithmic_bytes_per_sample": bytes_per_sample,
                         "work_per_sample": {"rays": dc[0] / paths, "node_visits": dc[1] / paths, "prim_tests": dc[3] / paths, "splats": dc[5] / paths}},
        }
        if not args.no_cpu_baseline and world == 1 and args.config == "cfg2":
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out))
    be.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
