#!/bin/bash
# development helper (GPU box): bench lines of the extended kernels (pt and ptdl), shipped library against variant libraries
#   bash tests/dev/mediaab2.sh [variant.so ...]
cd "${GRAFT_REPO_ROOT:-.}"
for lib in "" "$@"; do
  for c in media media_ptdl fog fog_ptdl cam_mb mb; do
    if [ -n "$lib" ]; then export CORONA_MI_LIB=$PWD/$lib; else unset CORONA_MI_LIB; fi
    python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -n 1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('${lib:-default}', '$c', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')
except Exception as e:
    print('${lib:-default}', '$c', 'FAILED', e)"
  done
done
