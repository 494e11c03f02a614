cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r5j/tests.txt
{
echo "== bench cfg2 (+ secondary cfg3), index sharding / tile sharding (one GPU): value, ms/step, kernel ms, secondary ms/step, kernel ms, image mean, 256-spp mean"
python3 bench.py --steps 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['roofline']['kernel_ms'], d['image']['mean_xyz'], d['image'].get('mean_xyz_many'), d['secondary']['image'].get('mean_xyz_many'))"
python3 bench.py --steps 5 --no-cpu-baseline --shard tiles 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['roofline']['kernel_ms'], d['image']['mean_xyz'], d['image'].get('mean_xyz_many'), d['secondary']['image'].get('mean_xyz_many'), d['config']['sharding'])"
} > gpurun_out/r5j/bench.txt 2>&1
cat gpurun_out/r5j/tests.txt gpurun_out/r5j/bench.txt
