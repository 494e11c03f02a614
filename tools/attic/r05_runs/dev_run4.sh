cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
V=$PWD/corona-13_amd/csrc/variants
{
echo "== r4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh fine large mb
echo "== current (hybrid)"; bash tools/ext_configs.sh fine large mb
echo "== no hybrid code, nothing staged, pools 48 KB"; CORONA_MI_LIB=$V/libcorona_mi_nohyb.so CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
echo "== ... and scene by value (no lazy loads)"; CORONA_MI_LIB=$V/libcorona_mi_nohybnolazy.so CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
echo "== r4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh fine large
} > gpurun_out/r5d/ext.txt 2>&1
cat gpurun_out/r5d/ext.txt
