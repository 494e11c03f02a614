cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r5h/tests.txt
(bash tests/dev/ab_env.sh aoslazy tile aoslazy tile 2>&1) > gpurun_out/r5h/ab.txt
{
echo "== bench cfg2, index sharding / tile sharding (one GPU)"
python3 bench.py --steps 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['roofline']['kernel_ms'], d['image']['mean_xyz'])"
python3 bench.py --steps 5 --no-cpu-baseline --shard tiles 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['roofline']['kernel_ms'], d['image']['mean_xyz'], d['config']['sharding'])"
} > gpurun_out/r5h/bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in indices tiles; do
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r5h/pmc_write_$mode -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 0 --no-cpu-baseline --shard $mode > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/r5h/pmc_write_$mode/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mi_path_kernel<false, true, true, false, false, false, false" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$mode", {k:(sum(v)/len(v), len(v)) for k,v in agg.items()})
PY
done > $R/gpurun_out/r5h/write.txt 2>&1
rm -rf $R/gpurun_out/r5h/pmc_write_*
cd $R; cat gpurun_out/r5h/tests.txt gpurun_out/r5h/bench.txt gpurun_out/r5h/write.txt; cut -c1-330 gpurun_out/r5h/ab.txt
