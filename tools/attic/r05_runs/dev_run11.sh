cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
(timeout 1200 python -m pytest tests -m gpu -x -q -k "tile or pixels" 2>&1 | tail -5) > gpurun_out/r5k/tests.txt
{
for sh in indices tiles indices tiles; do
python3 bench.py --steps 5 --no-cpu-baseline --shard $sh 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['roofline']['kernel_ms'], d['image'].get('mean_xyz_many'), d['secondary']['image'].get('mean_xyz_many'))"
done
} > gpurun_out/r5k/bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in indices tiles; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r5k/pmc_write_$mode -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 0 --no-cpu-baseline --shard $mode > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/r5k/pmc_write_$mode/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mi_path_kernel<false, true, true, false, false, false, false" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$mode", {k:(sum(v)/len(v), len(v)) for k,v in agg.items()})
PY
done > $R/gpurun_out/r5k/write.txt 2>&1
rm -rf $R/gpurun_out/r5k/pmc_write_*
cd $R; cat gpurun_out/r5k/tests.txt gpurun_out/r5k/bench.txt gpurun_out/r5k/write.txt
