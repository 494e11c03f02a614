#!/bin/bash
# development helper (GPU box): the shipped library and the variants against the round-2 library on one box
V=corona-13_amd/csrc/variants
mkdir -p gpurun_out
{
CORONA_MI_LIB=$V/libcorona_mi_r02.so python3 tests/dev/ab_r02.py
bash tests/dev/ab.sh $(ls $V | sed 's/libcorona_mi_//; s/.so//' | grep -v r02)
CORONA_MI_LIB=$V/libcorona_mi_r02.so python3 tests/dev/ab_r02.py
python3 tests/dev/ab_r02.py
python3 tests/dev/ab_r02.py --exact
} > gpurun_out/ab_bis.txt 2>&1
cat gpurun_out/ab_bis.txt
