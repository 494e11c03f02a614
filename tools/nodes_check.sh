python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
CORONA_MI_NODES=global python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
CORONA_MI_NODES=global CORONA_MI_MODE=wave python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/perf.sh | tail -1
CORONA_MI_NODES=global bash tools/perf.sh | tail -1
