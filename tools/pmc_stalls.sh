R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
for w in "" "ptdl"; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH_LEVEL --output-format csv -d /tmp/pk/a$w -- python3 $R/tools/pc_workload.py $w 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pk/b$w -- python3 $R/tools/pc_workload.py $w 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS --output-format csv -d /tmp/pk/c$w -- python3 $R/tools/pc_workload.py $w 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pk/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mi_path_kernel" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s avg %.5g" % (c, sum(v) / len(v)))
PY
