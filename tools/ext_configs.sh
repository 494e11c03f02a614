#!/bin/bash
# GPU box: one compact line per non-headline configuration (extended kernels, large trees): kernel, kernel time, Msamples/s.
#   tools/ext_configs.sh [configs...]
cd "${GRAFT_REPO_ROOT:-.}"
for c in ${@:-cfg3 cfg4 media media_ptdl fog fog_ptdl cam_mb mb fine large large_ptdl}; do
  python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
w = d['work_rate_vs_hbm']['live_work_per_sample']
print('%-11s %8.1f Msamples/s  kernel %7.2f ms  %s  rays %.3f nodes %.2f prims %.2f per sample' % ('$c', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'].split(' (')[0], w['rays'], w['node_visits'], w['prim_tests']))
" || echo "$c FAILED"
done
