#!/bin/bash
# GPU box: collect the rocprofv3 evidence behind bench.py's roofline object.
#   tools/profile.sh <tag>   -> gpurun_out/<tag>/{kernel_stats.csv, pmc_*.csv, bench.json}
# PMC counters are collected in their own passes (never combined with tracing), as the guide prescribes.
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
# the PMC passes run bench.py WITH its secondary (configs[2], ptdl) leg: both timed kernels get their counters from the same command
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1; cp $OUT/pmc_$name/*/*_counter_collection.csv $OUT/pmc_$name.csv; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pmc mem SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pmc grbm GRBM_GUI_ACTIVE
# instruction classes: what tools/micro/valu_mix.py builds its calibration kernel from
pmc mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM
pmc mix2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS
python3 $R/tools/hbm_copy.py > $OUT/hbm_copy.json 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_*/ $OUT/*.log
python3 - <<PY
import csv, glob, json, collections, hashlib
out = "$OUT"
build_id = hashlib.sha256(open("$R/corona-13_amd/csrc/libcorona_mi.so", "rb").read()).hexdigest()[:16]
# the timed kernels: the instantiations the library chose for cfg 2 and cfg 3 (its `secondary`), by the names bench.py got from
# mi_scene_kernel_name; bench.py also launches the counting instantiations once, outside its timed regions
_bench = json.load(open(out + "/bench.json"))
KERNELS = {"pt": _bench["roofline"]["kernel"].split(" (")[0], "ptdl": _bench["secondary"]["roofline"]["kernel"].split(" (")[0]}
for tag, kname in KERNELS.items():
    res = {"kernel": kname, "build_id": build_id}
    for f in glob.glob(out + "/pmc_*.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                res["VGPR_Count"] = r.get("VGPR_Count"); res["LDS_Block_Size"] = r.get("LDS_Block_Size"); res["Scratch_Size"] = r.get("Scratch_Size")
                res["Grid_Size"] = r.get("Grid_Size"); res["Workgroup_Size"] = r.get("Workgroup_Size")
        for k, v in agg.items():
            res[k] = sum(v) / len(v)
    res["paths_per_launch"] = 64 * 1280 * 736
    for r in csv.DictReader(open(out + "/kernel_stats.csv")):
        if kname in r["Name"]:
            res["kernel_ms"] = float(r["AverageNs"]) * 1e-6
            res["kernel_calls"] = int(r["Calls"])
    try:
        res.update(json.load(open(out + "/hbm_copy.json")))
    except Exception:
        pass
    if "GRBM_GUI_ACTIVE" in res and "SQ_ACTIVE_INST_VALU" in res:
        # gfx950's SIMD issues the common f32 ops of a wave64 in 2 cycles (MI355X_MICROARCH.md)
        simds, xcds = 1024, 8
        cycles = res["GRBM_GUI_ACTIVE"] / xcds
        res["gpu_cycles_per_launch"] = cycles
        res["valu_busy_pct_simd32"] = 100.0 * res["SQ_ACTIVE_INST_VALU"] * 2 / simds / cycles
        res["lane_utilisation"] = res["SQ_THREAD_CYCLES_VALU"] / (64.0 * res["SQ_ACTIVE_INST_VALU"])
        res["valu_instr_per_path"] = res["SQ_INSTS_VALU"] / res["paths_per_launch"]
        res["salu_instr_per_path"] = res.get("SQ_INSTS_SALU", 0.0) / res["paths_per_launch"]
        if "SQ_INSTS_VALU_FMA_F32" in res:
            # the instruction mix (round 5 priced it with a 2 / 4 / 8-cycle model; round 6 measures the rate a kernel of this mix sustains: tools/micro/valu_mix.py below)
            f32 = res.get("SQ_INSTS_VALU_ADD_F32", 0.0) + res.get("SQ_INSTS_VALU_MUL_F32", 0.0) + res["SQ_INSTS_VALU_FMA_F32"]
            trans = res.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            f64 = res.get("SQ_INSTS_VALU_ADD_F64", 0.0) + res.get("SQ_INSTS_VALU_MUL_F64", 0.0) + res.get("SQ_INSTS_VALU_FMA_F64", 0.0)
            other = res["SQ_INSTS_VALU"] - f32 - trans - f64
            res["valu_mix"] = {"f32_add_mul_fma": f32 / res["SQ_INSTS_VALU"], "transcendental": trans / res["SQ_INSTS_VALU"], "f64": f64 / res["SQ_INSTS_VALU"], "other": other / res["SQ_INSTS_VALU"]}
    json.dump(res, open(out + ("/pmc_summary.json" if tag == "pt" else "/pmc_summary_ptdl.json"), "w"), indent=1)
    print(json.dumps(res))
PY
# the vector-issue peak for the two kernels' instruction mixes (round 6): a generated memory-free kernel per summary, tools/micro/valu_mix.py
# (the keys of its output are the summaries' file names: rename them with the files when they go to profiles/rNN_*)
python3 $R/tools/micro/valu_mix.py $OUT/pmc_summary.json $OUT/pmc_summary_ptdl.json > $OUT/valu_mix_peak.json 2> $OUT/valu_mix.err || echo "valu_mix.py failed: see $OUT/valu_mix.err"
cat $OUT/bench.json; head -3 $OUT/kernel_stats.csv
