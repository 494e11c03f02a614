cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
(timeout 600 python -m pytest tests -m gpu -x -q -k "tile_sharding_statistics" 2>&1 | tail -40) > gpurun_out/r5i/tests.txt
(python3 bench.py --steps 3 --no-cpu-baseline --shard tiles 2>&1 | tail -5 | cut -c1-600) > gpurun_out/r5i/bench.txt
cat gpurun_out/r5i/tests.txt gpurun_out/r5i/bench.txt
