for p in 2097152 8388608 33554432; do
  echo "pool $p"; CORONA_MI_MODE=wave CORONA_MI_POOL=$p bash tools/perf.sh | tail -1
done
