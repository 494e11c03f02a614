#!/bin/bash
# GPU box: which LDS user owns the bank conflicts of the cfg 2 / cfg 3 kernels (lever 1c)? One rocprofv3 --pmc pass per knock-out of an LDS user,
# each the same command (bench.py --steps 2): the tree read from L2 instead of LDS (CORONA_MI_NODES=global), the exchange between waves off
# (CORONA_MI_REGROUP=0: no pools), both.   tools/lds_conflicts.sh <tag>  ->  gpurun_out/<tag>/lds_conflicts.txt
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/lds_conflicts.txt
run() { name=$1; shift
  rm -rf /tmp/ldsc_$name
  env "$@" true
  ( export "$@"; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/ldsc_$name -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > /tmp/ldsc_$name.log 2>&1 )
  python3 - "$name" /tmp/ldsc_$name >> $OUT/lds_conflicts.txt <<'PY'
import csv, glob, sys, collections
name, d = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mi_path_kernel<false" in k and ", false, false, false, false, false>" in k.replace("true, false, false>", "false, false, false>")[-60:] or "mi_path_kernel<false" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    if len(c.get("SQ_LDS_IDX_ACTIVE", [])) < 2:      # the counting instantiation's single launch is not a timed kernel
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    paths = 64 * 1280 * 736
    print("%-22s %s\n    LDS instr/path %.2f  LDS active cycles/path %.1f  bank-conflict cycles/path %.2f (%.1f %% of active)  addr conflicts/path %.2f  wait_inst_lds/wave_cycles %.4f" %
          (name, k.split("mi_path_kernel")[1][:64], m["SQ_INSTS_LDS"] / paths, m["SQ_LDS_IDX_ACTIVE"] / paths, m["SQ_LDS_BANK_CONFLICT"] / paths,
           100.0 * m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_LDS_ADDR_CONFLICT", 0.0) / paths, m.get("SQ_WAIT_INST_LDS", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
PY
}
run default CORONA_MI_DUMMY=1
run nodes_from_l2 CORONA_MI_NODES=global
run no_exchange CORONA_MI_REGROUP=0
run both CORONA_MI_NODES=global CORONA_MI_REGROUP=0
cat $OUT/lds_conflicts.txt
