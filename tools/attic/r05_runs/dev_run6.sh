cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
V=$PWD/corona-13_amd/csrc/variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r5f/tests.txt
{
echo "== r4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh fine large mb
echo "== current (LDS records for the top in breadth-first order, field-major HBM in depth-first order)"; bash tools/ext_configs.sh fine large large_ptdl mb cfg3
echo "== caller's numbering kept (CORONA_MI_NODE_ORDER=keep), nothing staged, pools 48 KB"; CORONA_MI_NODE_ORDER=keep CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
echo "== nothing staged, pools 48 KB"; CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
for pool in 8192 16384 32768 40960; do echo "-- CORONA_MI_NODES_POOL=$pool"; CORONA_MI_NODES_POOL=$pool bash tools/ext_configs.sh fine large; done
echo "== r4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh fine large
} > gpurun_out/r5f/ext.txt 2>&1
cat gpurun_out/r5f/tests.txt gpurun_out/r5f/ext.txt
