#!/bin/bash
# development helper (GPU box): parity tests, then the bench line without the CPU baseline
cd "${GRAFT_REPO_ROOT:-.}"
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('Msamples/s %.1f  ms/frame %.2f  kernel_ms %.2f  frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))
"
