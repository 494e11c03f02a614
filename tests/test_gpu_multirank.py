"""The multi-rank path with the REAL kernels on a one-GPU box (-m gpu): bench.py --share-gpu starts N ranks (torch.distributed.run) that all render on
device 0 and reduce their framebuffers over gloo -- RCCL refuses two ranks on one device, and the pool has no multi-GPU node (the N > 1 RCCL
exchange itself has never run; everything around it has, here). Strong scaling: the job is the single-rank job, so the reduced frame must be the single-rank
frame -- index-range sharding and tile-owned sharding, scalar and hero paths, an odd number of ranks."""
import json
import os
import subprocess
import sys

import pytest

from helpers import REPO

pytestmark = pytest.mark.gpu


def run_bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", *argv],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--shard", "tiles"], ["--wavelengths", "4"]], ids=["indices", "tiles", "hero"])
def test_ranks_sharing_one_gpu_render_the_single_rank_frame(extra):
    one = run_bench("--gpus", "1", "--config", "cfg2", "--scaling", "strong", *extra)
    assert one["n_gpus"] == 1 and one["image"]["mean_xyz"]
    for n in (2, 3):
        many = run_bench("--gpus", str(n), "--share-gpu", "--config", "cfg2", "--scaling", "strong", *extra)
        assert many["n_gpus"] == n and many["rccl_ranks"] == 0 and many["metric"].startswith("SHARED GPU")
        assert many["config"]["paths_per_step"] == one["config"]["paths_per_step"]
        # bench.py itself exits non-zero on a wrong path count or an image mean off the reference's; the reduced frame is the single-rank frame
        for a, b in zip(many["image"]["mean_xyz"], one["image"]["mean_xyz"]):
            assert abs(a - b) <= 2e-5 * abs(b), (many["image"], one["image"])


def test_weak_scaling_ranks_sharing_one_gpu():
    """weak scaling (every rank a whole frame of its own indices): twice the paths, the reduced frame = the sum of two frames -- bench.py's own image and
    path-count checks pass, and the mean per frame is the single-frame mean within the two frames' noise"""
    one = run_bench("--gpus", "1", "--config", "cfg2", "--scaling", "weak")
    two = run_bench("--gpus", "2", "--share-gpu", "--config", "cfg2", "--scaling", "weak")
    assert two["scaling"] == "weak" and two["config"]["paths_per_step"] == 2 * one["config"]["paths_per_step"]
    for a, b in zip(two["image"]["mean_xyz"], one["image"]["mean_xyz"]):
        assert abs(a - b) <= 0.01 * abs(b), (two["image"], one["image"])


@pytest.mark.parametrize("shard", ["indices", "tiles"])
def test_cfg5_eight_rank_rehearsal(shard):
    """configs[4] as the driver would launch it on an 8-GPU node -- 3840 x 2160 (film 3840 x 2176: eight ranks x two 100 MB framebuffers of the double-buffered
    reduce), strong scaling, index ranges and tile ownership -- with eight ranks on the one GPU there is, at 2 spp instead of 1024: the reduced frame is the
    single-rank frame, every path is counted (bench.py exits non-zero otherwise)"""
    one = run_bench("--gpus", "1", "--config", "cfg5", "--spp", "2", "--shard", shard)
    many = run_bench("--gpus", "8", "--share-gpu", "--config", "cfg5", "--spp", "2", "--shard", shard)
    assert many["n_gpus"] == 8 and many["scaling"] == "strong" and "REHEARSAL" in many["config"]["workload"]
    assert many["config"]["paths_per_step"] == one["config"]["paths_per_step"] == 2 * 3840 * 2176
    assert 8 * many["config"]["paths_per_step_per_gpu"] >= many["config"]["paths_per_step"] > 7 * many["config"]["paths_per_step_per_gpu"]
    for a, b in zip(many["image"]["mean_xyz"], one["image"]["mean_xyz"]):
        assert abs(a - b) <= 2e-5 * abs(b), (many["image"], one["image"])
