cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
mkdir -p gpurun_out/$TAG
bash tools/profile.sh $TAG > gpurun_out/$TAG/profile.out 2>&1
bash tools/all_configs.sh > gpurun_out/$TAG/all_configs.txt 2>&1
bash tools/ext_configs.sh > gpurun_out/$TAG/ext_configs.txt 2>&1
python3 tests/regression_report.py --out gpurun_out/$TAG/regression > gpurun_out/$TAG/regression.txt 2>&1
rm -rf gpurun_out/$TAG/regression/*/*.png gpurun_out/$TAG/regression/report.html
tail -2 gpurun_out/$TAG/profile.out | cut -c1-3000; cat gpurun_out/$TAG/all_configs.txt gpurun_out/$TAG/regression.txt
