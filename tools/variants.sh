#!/bin/bash
# development helper (build container): compile kernel variants for an A/B run on the GPU box.
#   tools/variants.sh <tag> [extra hipcc flags...]   ->  corona-13_amd/csrc/variants/libcorona_mi_<tag>.so
# -DMI_DEV_FAST: only the plain tree-in-LDS kernels (pt / ptdl, with and without RECORD) -- 20 s per variant.
# The variants travel to the GPU box with the snapshot; tests/dev/ab.sh runs parity + timing for each.
set -e
cd "$(dirname "$0")/../corona-13_amd"
tag=$1; shift
mkdir -p csrc/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -fno-slp-vectorize \
  -mllvm -enable-post-misched=0 -Wall -Wno-unused-function -I../include -Ihost -Icsrc ${MI_DEV_FAST_FLAG:--DMI_DEV_FAST} "$@" \
  -shared csrc/mi_abi.hip -o csrc/variants/libcorona_mi_$tag.so
echo "built csrc/variants/libcorona_mi_$tag.so ($*)"
