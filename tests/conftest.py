import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (host library + oracle: gcc only; HIP library: hipcc cross-compiles without a GPU)
    if not (REPO / "corona-13_amd" / "host" / "libcorona_host.so").exists() or \
       not (REPO / "corona-13_amd" / "csrc" / "libcorona_mi.so").exists():
        subprocess.check_call(["make", "-C", str(REPO / "corona-13_amd")])
    if not (REPO / "oracle" / "liboracle.so").exists():
        subprocess.check_call(["make", "-C", str(REPO / "oracle"), "liboracle.so"])
