#!/bin/bash
# GPU box: rocprofv3 kernel-trace summaries of the non-headline kernels (ptdl, device-built tree, Halton, media, motion blur), one bench run each.
#   tools/profile_variants.sh <tag>  -> gpurun_out/<tag>/variants_kernel_stats.csv (+ variants_bench.jsonl)
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/variants_bench.jsonl
echo "variant,Name,Calls,TotalDurationNs,AverageNs,Percentage" > $OUT/variants_kernel_stats.csv
run() { name=$1; shift
  rm -rf /tmp/pv_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/pv_$name.log 2>/dev/null
  grep '^{' /tmp/pv_$name.log | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/variants_bench.jsonl
  python3 - "$name" /tmp/pv_$name $OUT/variants_kernel_stats.csv <<'PY'
import csv, glob, sys
name, d, out = sys.argv[1:4]
with open(out, "a") as f:
    for r in csv.DictReader(open(glob.glob(d + "/*/*_kernel_stats.csv")[0])):
        if "mi_path_kernel" in r["Name"]:
            f.write("%s,\"%s\",%s,%s,%s,%s\n" % (name, r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
}
run cfg3_ptdl --config cfg3
run cfg4_rough --config cfg4
run cfg2_device_tree --config cfg2 --tree device
run cfg2_halton --config cfg2 --points halton
run cfg3_halton --config cfg3 --points halton
run media_pt --config media
run media_ptdl --config media_ptdl
run fog_pt --config fog
run fog_ptdl --config fog_ptdl
run cam_mb --config cam_mb
run mb --config mb
cat $OUT/variants_kernel_stats.csv
