#!/bin/bash
# GPU box: per-kernel PMC averages for a probe script:  tools/pmc_kernels.sh <script.py>
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
run() { rocprofv3 --pmc "$@" --output-format csv -d /tmp/pk/$1 -- python3 $R/$SCRIPT > /dev/null 2>&1; }
SCRIPT=$1
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM
run SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL
run SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_VMEM
run FETCH_SIZE
run WRITE_SIZE
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pk/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_VGPR"] = [float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"])]
        agg[k]["_scratch"] = [float(r["Scratch_Size"])]
for k, d in agg.items():
    if "rocclr" in k: continue
    print(k, "launches", len(d.get("SQ_WAVES", [])))
    for c, v in sorted(d.items()):
        print("   %-26s sum %.4g  avg %.4g" % (c, sum(v), sum(v) / len(v)))
PY
