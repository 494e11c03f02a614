#!/bin/bash
# development helper (GPU box): parity + bench line for every kernel-variant library csrc/libcorona_mi*.so, both organisations
for f in corona-13_amd/csrc/libcorona_mi*.so; do
  for m in mega wave; do
    echo "variant $f $m"; CORONA_MI_MODE=$m CORONA_MI_LIB=$PWD/$f bash tools/perf.sh | tail -2
  done
done
