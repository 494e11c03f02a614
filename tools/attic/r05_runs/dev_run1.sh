cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r5a/tests.txt
(bash tests/dev/ab_env.sh base lazy aos aoslazy base aoslazy 2>&1) > gpurun_out/r5a/ab.txt
(rocprofv3 -L 2>&1 | grep -i -E "SQ_INSTS_VALU|SQ_INSTS_|SQ_ACTIVE|SQ_VALU|SQ_THREAD" | head -80) > gpurun_out/r5a/counters.txt
(MI_SCALE_SAH="2" timeout 600 python3 tools/build_scale.py 1 2 8 16 2>&1) > gpurun_out/r5a/scale.txt
tail -3 gpurun_out/r5a/tests.txt; cat gpurun_out/r5a/ab.txt; cat gpurun_out/r5a/scale.txt
