#!/usr/bin/env python3
"""What the oracle achieves against the reference's path dumps (tests/golden/paths_*.npz), case by case: the numbers
tests/test_oracle_golden.py's thresholds are derived from (measured value + a margin, not loose constants).
  python3 tests/golden/measure_oracle_vs_reference.py > tests/golden/oracle_vs_reference_measured.json
CPU only (oracle + fixtures)."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import test_oracle_golden as T          # noqa: E402

out = {}
for name, sampler, scene_path, etol in T.CASES + T.METAL_REFERENCE_CASES:
    m = T.measure_case(name, sampler, scene_path)
    if m is not None:
        out[name] = m
json.dump(out, sys.stdout, indent=1, sort_keys=True)
print()
