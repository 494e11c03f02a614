#!/bin/bash
# development helper (GPU box): per-kernel time split of the wavefront organisation (what share is traversal, what is shading)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export CORONA_MI_MODE=wave
rm -rf /tmp/ws
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws -- python3 $R/tools/wf_probe.py > /tmp/ws.log 2>&1
tail -2 /tmp/ws.log
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/ws/*/*_kernel_stats.csv")[0])):
    print("%-40s calls %5s total %8.3f ms avg %8.3f ms  %5s%%" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Percentage"]))
PY
