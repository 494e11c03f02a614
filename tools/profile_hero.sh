#!/bin/bash
# GPU box: PMC passes of the HERO kernels (four wavelengths per path), the counters of tools/profile.sh that say how busy the vector pipes are and how full the waves:
#   tools/profile_hero.sh <tag>   -> gpurun_out/<tag>/pmc_summary_hero.json (+ the raw CSVs)
# Same rules as tools/profile.sh: counters in their own passes, never combined with tracing; every pass is the same command.
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg3; do
  CMD="$R/bench.py --config $CFG --wavelengths 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
  python3 $CMD > $OUT/hero_bench_$CFG.json 2> $OUT/hero_bench_$CFG.err          # un-profiled: the line names the kernel the library launches (config.kernel)
  pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_hero_${CFG}_$name -- python3 $CMD > $OUT/pmc_hero_${CFG}_$name.log 2>&1; cp $OUT/pmc_hero_${CFG}_$name/*/*_counter_collection.csv $OUT/pmc_hero_${CFG}_$name.csv; }
  pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
  pmc mem SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
  pmc grbm GRBM_GUI_ACTIVE
  pmc mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM
  pmc mix2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_WAIT_INST_LDS
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  rm -rf $OUT/pmc_hero_${CFG}_*/ $OUT/*.log
done
python3 - <<PY
import csv, glob, json, collections, hashlib
out = "$OUT"
res = {"build_id": hashlib.sha256(open("$R/corona-13_amd/csrc/libcorona_mi.so", "rb").read()).hexdigest()[:16], "paths_per_launch": 64 * 1280 * 736}
for cfg, ptdl in (("cfg2", "false"), ("cfg3", "true")):
    # the instantiation the library launched for this configuration, by its own name (bench.py: config.kernel from mi_scene_kernel_name) -- not re-derived here
    kname = json.load(open(out + "/hero_bench_%s.json" % cfg))["config"]["kernel"].split(" (")[0]
    r = {"kernel": kname}
    for f in glob.glob(out + "/pmc_hero_%s_*.csv" % cfg):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kname in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                r["VGPR_Count"] = row.get("VGPR_Count"); r["Scratch_Size"] = row.get("Scratch_Size"); r["Workgroup_Size"] = row.get("Workgroup_Size"); r["LDS_Block_Size"] = row.get("LDS_Block_Size")
        for k, v in agg.items():
            r[k] = sum(v) / len(v)
    if "SQ_INSTS_VALU" not in r:
        raise SystemExit("profile_hero.sh: no counter row matches kernel %s of %s" % (kname, cfg))
    n = res["paths_per_launch"]
    if "SQ_INSTS_VALU" in r:
        r["valu_instr_per_path"] = r["SQ_INSTS_VALU"] / n
        r["lane_utilisation"] = r["SQ_THREAD_CYCLES_VALU"] / (64.0 * r["SQ_ACTIVE_INST_VALU"])
        r["salu_instr_per_path"] = r.get("SQ_INSTS_SALU", 0.0) / n
        plain = sum(r.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
        trans = r.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        f64 = sum(r.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        other = r["SQ_INSTS_VALU"] - plain - trans - f64
        r["valu_mix"] = {"f32_add_mul_fma": plain / r["SQ_INSTS_VALU"], "transcendental": trans / r["SQ_INSTS_VALU"], "f64": f64 / r["SQ_INSTS_VALU"], "other": other / r["SQ_INSTS_VALU"]}
        if "GRBM_GUI_ACTIVE" in r:
            cycles = r["GRBM_GUI_ACTIVE"] / 8
            r["gpu_cycles_per_launch"] = cycles
    res[cfg] = r
json.dump(res, open(out + "/pmc_summary_hero.json", "w"), indent=1)
print(json.dumps({c: {k: res[c].get(k) for k in ("valu_instr_per_path", "lane_utilisation", "salu_instr_per_path", "Scratch_Size", "Workgroup_Size", "valu_mix")} for c in ("cfg2", "cfg3")}, indent=1))
PY
