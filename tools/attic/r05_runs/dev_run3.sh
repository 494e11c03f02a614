cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
V=$PWD/corona-13_amd/csrc/variants
{
echo "== r4 library (same box)"
CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh cfg3 media media_ptdl fog_ptdl cam_mb mb fine large
echo "== current library"
bash tools/ext_configs.sh cfg3 media media_ptdl fog_ptdl cam_mb mb fine large large_ptdl
echo "== mb: all of the tree in LDS, no pools (CORONA_MI_MB_ALL_LDS=1)"
CORONA_MI_MB_ALL_LDS=1 bash tools/ext_configs.sh mb
echo "== fine / large: the pools' share of LDS against staged nodes"
for pool in 0 8192 16384 24576 32768 40960; do
  echo "-- CORONA_MI_NODES_POOL=$pool"
  CORONA_MI_NODES_POOL=$pool bash tools/ext_configs.sh fine large
done
echo "-- nothing staged (CORONA_MI_NODES_TOP=0), pools 24 KB / 48 KB"
CORONA_MI_NODES_TOP=0 bash tools/ext_configs.sh fine large
CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
} > gpurun_out/r5c/ext.txt 2>&1
(timeout 900 python -m pytest tests -m gpu -x -q -k "moving or mb or blur or large" 2>&1 | tail -5) > gpurun_out/r5c/tests.txt
cat gpurun_out/r5c/ext.txt gpurun_out/r5c/tests.txt
