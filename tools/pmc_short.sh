#!/bin/bash
# GPU box: a few PMC sums of the path kernels for a probe script:  tools/pmc_short.sh <script.py> [args]
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
run() { rocprofv3 --pmc "$@" --output-format csv -d /tmp/pk/$1 -- python3 $R/$SCRIPT $ARGS > /dev/null 2>&1; }
SCRIPT=$1; shift; ARGS="$@"
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run FETCH_SIZE
run WRITE_SIZE
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pk/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "mi_path_kernel" not in k and "mi_wave_kernel" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    s = {c: sum(v) for c, v in d.items()}
    print(k)
    print("  " + " ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(s.items())))
    if "SQ_INSTS_VALU" in s:
        print("  valu/path(17spp) %.1f  lane_util %.3f  wait_any/wave_cycles %.3f  wait_inst/wave_cycles %.3f  vmem latency index %.2f" % (
            s["SQ_INSTS_VALU"] / (17 * 1280 * 736), s["SQ_THREAD_CYCLES_VALU"] / 64 / s["SQ_ACTIVE_INST_VALU"], s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
            s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], s["SQ_INST_LEVEL_VMEM"] / (s["SQ_INSTS_VMEM_RD"] + s["SQ_INSTS_VMEM_WR"])))
PY
