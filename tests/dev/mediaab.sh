# development helper (GPU box): bench lines of the extended / motion-blur pt kernels, shipped library against a variant
#   bash tests/dev/mediaab.sh [variant.so]
for lib in "" $1; do
  for c in media fog cam_mb mb; do
    if [ -n "$lib" ]; then export CORONA_MI_LIB=$PWD/$lib; else unset CORONA_MI_LIB; fi
    python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('${lib:-default}', '$c', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')"
  done
done
