#!/bin/bash
# GPU box: every BASELINE.json configuration on one MI355X (3 steps each), one line per configuration -- the table in DESIGN.md section 4
#   tools/all_configs.sh        the scalar kernels;   tools/all_configs.sh hero   four wavelengths per path (bench.py --wavelengths 4: paths per second)
if [ "$1" = hero ]; then
  for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
    python3 bench.py --config $c --wavelengths 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%s  hero  %8.1f Mpaths/s  %8.2f ms/step  image mean %s' % ('$c', d['value'], d['ms_per_step'], ' '.join('%.4f' % x for x in d['image']['mean_xyz'])))
"
  done
  exit 0
fi
for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
w = d['work_rate_vs_hbm']
print('%s  %8.1f Msamples/s  %8.2f ms/step  kernel %.2f ms  B/sample %.0f (%s)  work rate / HBM peak %.3f  rays/sample %.3f node visits %.2f prim tests %.2f' % ('$c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'],
      w['algorithmic_bytes_per_sample'], w['work_counts'].split()[0], w['rate_over_hbm_peak'], w['live_work_per_sample']['rays'], w['live_work_per_sample']['node_visits'], w['live_work_per_sample']['prim_tests']))
"
done
