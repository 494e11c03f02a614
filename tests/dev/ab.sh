#!/bin/bash
# development helper (GPU box): parity + timing of every kernel variant under corona-13_amd/csrc/variants/ (tools/variants.sh),
# bracketed by the shipped library.  tests/dev/ab.sh [tags...] > gpurun_out/ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
V=corona-13_amd/csrc/variants
tags=${@:-$(ls $V | sed 's/libcorona_mi_//; s/.so//')}
python3 tests/dev/ab_one.py
for t in $tags; do
  extra=""; case $t in *loops*) extra="--loops";; esac
  CORONA_MI_LIB=$PWD/$V/libcorona_mi_$t.so timeout 300 python3 tests/dev/ab_one.py $extra || echo "$t FAILED"
done
python3 tests/dev/ab_one.py
