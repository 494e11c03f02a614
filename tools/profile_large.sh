#!/bin/bash
# GPU box: rocprofv3 evidence for a tree that does not fit LDS (HBM / L2 traffic of the node fetches).
#   tools/profile_large.sh <tag> [config]   -> gpurun_out/<tag>/pmc_summary_<config>.json (+ the raw CSVs)
# Same rules as tools/profile.sh: counters in their own passes, never combined with tracing; every pass is the same command.
TAG=${1:-prof}; CFG=${2:-large}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
python3 $CMD > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -- python3 $CMD > $OUT/trace_$CFG.log 2>&1
cp $OUT/trace_$CFG/*/*_kernel_stats.csv $OUT/kernel_stats_$CFG.csv
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${CFG}_$name -- python3 $CMD > $OUT/pmc_${CFG}_$name.log 2>&1; cp $OUT/pmc_${CFG}_$name/*/*_counter_collection.csv $OUT/pmc_${CFG}_$name.csv; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc mem SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE
rm -rf $OUT/trace_$CFG $OUT/pmc_${CFG}_*/ $OUT/*.log
python3 - <<PY
import csv, glob, json, collections, hashlib
out, cfg = "$OUT", "$CFG"
bench = json.load(open(out + "/bench_%s.json" % cfg))
kname = bench["roofline"]["kernel"].split(" (")[0]
res = {"config": cfg, "workload": bench["config"]["workload"], "kernel": kname, "build_id": hashlib.sha256(open("$R/corona-13_amd/csrc/libcorona_mi.so", "rb").read()).hexdigest()[:16]}
for f in glob.glob(out + "/pmc_%s_*.csv" % cfg):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["LDS_Block_Size"] = r.get("LDS_Block_Size"); res["Scratch_Size"] = r.get("Scratch_Size")
    for k, v in agg.items():
        res[k] = sum(v) / len(v)
for r in csv.DictReader(open(out + "/kernel_stats_%s.csv" % cfg)):
    if kname in r["Name"]:
        res["kernel_ms"] = float(r["AverageNs"]) * 1e-6; res["kernel_calls"] = int(r["Calls"])
res["kernel_ms_bench"] = bench["roofline"]["kernel_ms"]
res["paths_per_launch"] = 64 * 1280 * 736
w = bench["work_rate_vs_hbm"]["live_work_per_sample"]
res["live_work_per_sample"] = w
# 128 B per node visit, 104 B per primitive test, 384 B per splat -- priced with the LIVE counters of this scene
res["algorithmic_bytes_per_launch"] = (128.0 * w["node_visits"] + 104.0 * w["prim_tests"] + 384.0 * w["splats"]) * res["paths_per_launch"]
if "FETCH_SIZE" in res and "WRITE_SIZE" in res and "kernel_ms" in res:
    # FETCH_SIZE counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM section); counters are in KiB
    res["hbm_traffic_bytes_per_launch"] = (2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0
    res["hbm_measured_gbs"] = res["hbm_traffic_bytes_per_launch"] / (res["kernel_ms"] * 1e-3) / 1e9
    res["hbm_traffic_over_algorithmic"] = res["hbm_traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    res["algorithmic_rate_over_hbm_peak"] = res["algorithmic_bytes_per_launch"] / (res["kernel_ms"] * 1e-3) / 8e12
if "TCC_HIT_sum" in res:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
    res["l2_requests_per_node_visit"] = (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]) / (w["node_visits"] * res["paths_per_launch"])
if "SQ_INSTS_VALU" in res:
    res["valu_instr_per_path"] = res["SQ_INSTS_VALU"] / res["paths_per_launch"]
    res["lane_utilisation"] = res["SQ_THREAD_CYCLES_VALU"] / (64.0 * res["SQ_ACTIVE_INST_VALU"]) if "SQ_THREAD_CYCLES_VALU" in res else None
json.dump(res, open(out + "/pmc_summary_%s.json" % cfg, "w"), indent=1)
print(json.dumps(res))
PY
