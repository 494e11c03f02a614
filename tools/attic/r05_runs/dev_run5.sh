cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
V=$PWD/corona-13_amd/csrc/variants
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/r5e/tests.txt
{
echo "== r4 library"; CORONA_MI_LIB=$V/libcorona_mi_r4full.so bash tools/ext_configs.sh fine large mb
echo "== current (LDS records + field-major HBM, hybrid)"; bash tools/ext_configs.sh fine large large_ptdl mb
echo "== nothing staged, pools 48 KB"; CORONA_MI_NODES_TOP=0 CORONA_MI_NODES_POOL=49152 bash tools/ext_configs.sh fine large
for pool in 16384 32768; do echo "-- CORONA_MI_NODES_POOL=$pool"; CORONA_MI_NODES_POOL=$pool bash tools/ext_configs.sh fine large; done
} > gpurun_out/r5e/ext.txt 2>&1
(bash tests/dev/ab_env.sh aoslazy v2 aoslazy v2 2>&1) > gpurun_out/r5e/ab.txt
cat gpurun_out/r5e/tests.txt gpurun_out/r5e/ext.txt; cut -c1-420 gpurun_out/r5e/ab.txt
