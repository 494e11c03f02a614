#!/bin/bash
# development helper (GPU box): instruction counts of the cfg-2 kernel, round-2 library against this one (exact and default rounds)
R=${GRAFT_REPO_ROOT:-$PWD}
V=$R/corona-13_amd/csrc/variants
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
export AB_ONLY="${AB_ONLY:-pt}"
run() { # tag, args...
  tag=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pk/$tag.a -- python3 $R/tests/dev/ab_r02.py "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pk/$tag.b -- python3 $R/tests/dev/ab_r02.py "$@" > /dev/null 2>&1
}
CORONA_MI_LIB=$V/libcorona_mi_r02.so run r02
run exact --exact
[ -z "$PMC_NO_FAST" ] && run fast
python3 - > $R/gpurun_out/pmc_ab.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pk/*/*/*_counter_collection.csv"):
    tag = f.split("/")[3].split(".")[0]
    for r in csv.DictReader(open(f)):
        if "mi_path_kernel" not in r["Kernel_Name"]: continue
        if int(r["Grid_Size"]) < 200000: continue
        agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for d in agg.values() for c in d})
print("%-24s" % "counter", *["%14s" % t for t in agg])
for c in names:
    print("%-24s" % c, *["%14.5g" % (max(agg[t][c]) if agg[t][c] else 0) for t in agg])
PY
cat $R/gpurun_out/pmc_ab.txt
