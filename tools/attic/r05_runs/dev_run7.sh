cd $GRAFT_REPO_ROOT
V=$PWD/corona-13_amd/csrc/variants
mkdir -p gpurun_out/r5g
bash tools/profile_large.sh r5g large > gpurun_out/r5g/large.out 2>&1
bash tools/profile_large.sh r5g huge > gpurun_out/r5g/huge.out 2>&1
bash tools/profile_large.sh r5g fine > gpurun_out/r5g/fine.out 2>&1
{
echo "== tools/build_scale.py (counting kernels, 8 spp), current library"; timeout 600 python3 tools/build_scale.py 2 8 16
echo "== the same, round-4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so timeout 600 python3 tools/build_scale.py 2 8 16
echo "== bench lines huge: current / r4"; bash tools/ext_configs.sh huge; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh huge
} > gpurun_out/r5g/scale.txt 2>&1
tail -3 gpurun_out/r5g/large.out | cut -c1-1500; cat gpurun_out/r5g/scale.txt
