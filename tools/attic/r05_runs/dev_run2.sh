cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
{
echo "== extended configs, full library (lazy scene + 128-byte node records)"
bash tools/ext_configs.sh
echo "== fine / large: the pools' share of LDS against staged nodes"
for pool in 0 16384 24576 32768 49152; do for top in "" 0; do
  echo "-- CORONA_MI_NODES_POOL=$pool CORONA_MI_NODES_TOP=$top"
  CORONA_MI_NODES_POOL=$pool CORONA_MI_NODES_TOP=$top bash tools/ext_configs.sh fine large
done; done
} > gpurun_out/r5b/ext.txt 2>&1
cat gpurun_out/r5b/ext.txt
