"""bench.py's multi-GPU entry point without a GPU: `python bench.py --gpus 2 --stub` must start TWO ranks on its own (children
of a process that never touches the GPU), shard the path indices, all-reduce the frame (gloo here, RCCL on the GPU box) and
report n_gpus == 2. The stub backend renders nothing -- it adds the number of indices it was handed to one framebuffer
element -- so the reduced last frame must sum to the size of the job."""
import json
import os
import subprocess
import sys

from helpers import REPO


def run_bench(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=600)


def json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_weak():
    r = run_bench("--gpus", "2", "--stub", "--steps", "2", "--warmup", "1", "--config", "cfg1")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 4 * 256 * 256
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["config"]["paths_per_step"] == 2 * per_frame and out["config"]["paths_per_step_per_gpu"] == per_frame
    assert out["stub"]["reduced_sum_last_frame"] == out["stub"]["expected"] == 2 * per_frame     # both ranks' shares arrived in the reduced frame
    assert out["metric"].startswith("STUB")                                                     # never mistaken for a measurement


def test_gpus_2_strong_scaling_keeps_the_job_fixed():
    r = run_bench("--gpus", "2", "--stub", "--steps", "2", "--warmup", "0", "--config", "cfg1", "--scaling", "strong")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 4 * 256 * 256
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["paths_per_step"] == per_frame and out["config"]["paths_per_step_per_gpu"] == per_frame // 2
    assert out["stub"]["reduced_sum_last_frame"] == per_frame


def test_single_rank_and_launcher_mismatch():
    r = run_bench("--gpus", "1", "--stub", "--steps", "1", "--warmup", "0", "--config", "cfg1")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["stub"]["reduced_sum_last_frame"] == 4 * 256 * 256
    # a launcher that started a different number of ranks than --gpus says: refuse instead of reporting a wrong n_gpus
    r = run_bench("--gpus", "4", "--stub", "--steps", "1", "--config", "cfg1", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stdout + r.stderr)


def test_gpus_8_weak_cfg2_and_strong_cfg5():
    """the driver's 1 -> 8 curve without hardware: eight spawned ranks (gloo), cfg 2 weak (every rank a whole 64-spp frame of its own
    indices) and cfg 5 strong (configs[4]: 3840x2160, 1024 spp = 8.56 G paths split eight ways, one framebuffer reduce of 100 MB)"""
    r = run_bench("--gpus", "8", "--stub", "--steps", "2", "--warmup", "1", "--config", "cfg2", "--scaling", "weak")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 64 * 1280 * 736
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "weak"
    assert out["config"]["paths_per_step"] == 8 * per_frame and out["config"]["paths_per_step_per_gpu"] == per_frame
    assert out["stub"]["reduced_sum_last_frame"] == out["stub"]["expected"] == 8 * per_frame
    r = run_bench("--gpus", "8", "--stub", "--steps", "1", "--warmup", "0", "--config", "cfg5")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    job = 1024 * 3840 * 2176
    assert out["n_gpus"] == 8 and out["scaling"] == "strong"
    assert out["config"]["paths_per_step"] == job and out["config"]["paths_per_step_per_gpu"] == job // 8
    assert out["stub"]["reduced_sum_last_frame"] == out["stub"]["expected"] == job        # exact: 8 556 380 160 paths


def test_cfg5_defaults_to_strong_scaling():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", REPO / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.CONFIGS["cfg5"]["scaling"] == "strong" and mod.CONFIGS["cfg5"]["spp"] == 1024
    assert mod.CONFIGS["cfg2"]["scaling"] == "strong"        # the headline configuration is one fixed 64-spp frame: N GPUs split it
    assert abs(128 * mod.REFERENCE_WORK["cfg2"]["node_visits"] + 104 * mod.REFERENCE_WORK["cfg2"]["prim_tests"] + 384 * mod.REFERENCE_WORK["cfg2"]["splats"] - 2822) < 2


def test_gpus_2_tile_owned_sharding():
    """--shard tiles: rank r renders the 32 x 32 film tiles t = r (mod N) of every frame of the step (mi_render_tiles): cfg 1's 256 x 256 film has
    64 tiles, 32 per rank; strong scaling splits the frame's 4 spp x 64 tiles, weak gives the step N x 4 frames"""
    r = run_bench("--gpus", "2", "--stub", "--steps", "2", "--warmup", "1", "--config", "cfg1", "--shard", "tiles", "--scaling", "strong")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 4 * 256 * 256
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "tiles" in out["config"]["sharding"]
    assert out["config"]["paths_per_step"] == per_frame and out["config"]["paths_per_step_per_gpu"] == per_frame // 2
    assert out["stub"]["reduced_sum_last_frame"] == per_frame
    r = run_bench("--gpus", "3", "--stub", "--steps", "1", "--warmup", "0", "--config", "cfg1", "--shard", "tiles")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    # weak: 3 x 4 frames; 64 tiles over three ranks = 22 + 21 + 21 -- rank 0 holds 22 of them
    assert out["scaling"] == "weak" and out["config"]["paths_per_step"] == 3 * per_frame and out["config"]["paths_per_step_per_gpu"] == 12 * 22 * 1024
    assert out["stub"]["reduced_sum_last_frame"] == 3 * per_frame


def test_gpus_4_default_is_strong_scaling_of_the_headline_frame():
    """python bench.py --gpus N as the driver launches it: configs[1] is ONE 64-spp frame, four ranks render a quarter of its path indices each"""
    r = run_bench("--gpus", "4", "--stub", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 64 * 1280 * 736
    assert out["n_gpus"] == 4 and out["scaling"] == "strong"
    assert out["config"]["paths_per_step"] == per_frame and out["config"]["paths_per_step_per_gpu"] == per_frame // 4
    assert out["stub"]["reduced_sum_last_frame"] == per_frame


def test_committed_profile_files_have_what_bench_reads():
    """bench.py's roofline record reads the newest committed PMC summaries and the floor file; a missing key there used to surface only on the GPU box, in the
    one run whose library matches the profile (round 5: the regenerated floor file had lost its `mix` block and bench.py raised KeyError)"""
    import json
    prof = REPO / "profiles"
    floor = json.load(open(sorted(prof.glob("r*_valu_floor.json"))[-1]))
    for key in ("blocks", "vertex_valu", "vertex_valu_ptdl", "mix"):
        assert key in floor, key
    for cfg, k in (("cfg2", "surface_vertices_per_ray"), ("cfg3", "surface_vertices_per_extension_ray")):
        assert {k, "diffuse_share", "dielectric_share"} <= set(floor["mix"][cfg])
    for b in ("node_visit", "prim_test", "generate", "shadow_resolve", "splat_pass_of_four"):
        assert "valu" in floor["blocks"][b], b
    for name in ("pmc_summary.json", "pmc_summary_ptdl.json"):
        s = json.load(open(sorted(prof.glob("r*_" + name))[-1]))
        for key in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "gpu_cycles_per_launch", "build_id", "kernel", "kernel_ms"):
            assert key in s, (name, key)


def test_gpus_2_hero_wavelengths():
    """--wavelengths 4 reaches every rank's backend; the line says so, reports no scalar-kernel roofline, and the job's paths are shared as usual"""
    r = run_bench("--gpus", "2", "--stub", "--steps", "2", "--warmup", "0", "--config", "cfg1", "--wavelengths", "4")
    assert r.returncode == 0, r.stdout + r.stderr
    out = json_line(r.stdout)
    per_frame = 4 * 256 * 256
    assert out["n_gpus"] == 2 and out["config"]["wavelengths_per_path"] == 4 and out["roofline"] is None and out["work_rate_vs_hbm"] is None
    assert out["stub"]["reduced_sum_last_frame"] == out["stub"]["expected"] == out["config"]["paths_per_step"]
    assert run_bench("--stub", "--wavelengths", "3").returncode != 0
