for c in cfg1 cfg2 cfg3 cfg4 cfg5; do
  python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%s  %8.1f Msamples/s  %8.2f ms/step  B/sample %.0f  frac %.3f  rays/sample %.3f' % ('$c', d['value'], d['ms_per_step'], r['algorithmic_bytes_per_sample'], r['frac'], r['live_work_per_sample']['rays']))
"
done
