#!/bin/bash
# development helper (build container): compile a kernel variant for an A/B run on the GPU box.
#   tools/variants.sh <tag> [extra hipcc flags...]   ->  corona-13_amd/csrc/variants/libcorona_mi_<tag>.so
# -DMI_DEV_FAST: only the plain tree-in-LDS kernels (pt / ptdl, exact and FAST rounds, with and without RECORD / COUNT) -- 12 s per
# variant on 8 cores. MI_DEV_FAST_FLAG= (empty) builds everything. The variants travel to the GPU box with the snapshot;
# tests/dev/ab.sh runs parity + timing for each. Prints the registers / spills of the variant's kernels (tools/kstat.py).
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p corona-13_amd/csrc/variants
rm -rf /tmp/mi_variants/$tag          # objects of an earlier variant of the same name were built with other flags
make -s -C corona-13_amd -j8 MI_DEFS="${MI_DEV_FAST_FLAG--DMI_DEV_FAST} $*" BUILD=/tmp/mi_variants/$tag MI_LIB=csrc/variants/libcorona_mi_$tag.so csrc/variants/libcorona_mi_$tag.so
echo "built corona-13_amd/csrc/variants/libcorona_mi_$tag.so ($*)"
python3 tools/kstat.py corona-13_amd/csrc/variants/libcorona_mi_$tag.so mi_path_kernel | grep -v "kernel<true"
