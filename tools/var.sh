#!/bin/bash
# development helper (GPU box): parity + bench line for every kernel-variant library csrc/libcorona_mi*.so
for f in corona-13_amd/csrc/libcorona_mi_*.so; do
  echo "variant $f"; CORONA_MI_LIB=$PWD/$f bash tools/perf.sh | tail -2
done
