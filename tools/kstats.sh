#!/bin/bash
# development helper (GPU box): per-kernel times of one probe run (tools/wf_probe.py: 1 + 16 spp of cfg 2)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $R/tools/wf_probe.py > /tmp/ks.log 2>&1
tail -1 /tmp/ks.log
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/ks/*/*_kernel_stats.csv")[0])):
    print("%-48s calls %5s total %8.3f ms avg %8.3f ms  %5s%%" % (r["Name"][:48], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
