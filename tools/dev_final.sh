cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
bash tools/profile.sh $TAG > gpurun_out/$TAG/profile.out 2>&1
bash tools/all_configs.sh > gpurun_out/$TAG/all_configs.txt 2>&1
bash tools/ext_configs.sh > gpurun_out/$TAG/ext_configs.txt 2>&1
# hero wavelengths (four per path): bench lines of cfg 2 / cfg 3 and the rocprofv3 kernel statistics of the same commands
for c in cfg2 cfg3; do
  python3 bench.py --config $c --wavelengths 4 --steps 5 --no-cpu-baseline --no-secondary > gpurun_out/$TAG/hero_bench_$c.json 2> gpurun_out/$TAG/hero_bench_$c.err
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/hero_trace_$c -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --wavelengths 4 --steps 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1)
  cp gpurun_out/$TAG/hero_trace_$c/*/*_kernel_stats.csv gpurun_out/$TAG/hero_kernel_stats_$c.csv; rm -rf gpurun_out/$TAG/hero_trace_$c
done
python3 tools/hero_time.py --extended 64 > gpurun_out/$TAG/hero_time.txt 2>&1
python3 tests/regression_report.py --out gpurun_out/$TAG/regression > gpurun_out/$TAG/regression.txt 2>&1
rm -rf gpurun_out/$TAG/regression/*/*.png gpurun_out/$TAG/regression/report.html
tail -2 gpurun_out/$TAG/profile.out | cut -c1-3000; cat gpurun_out/$TAG/all_configs.txt gpurun_out/$TAG/regression.txt
