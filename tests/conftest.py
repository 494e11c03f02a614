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
    built = [REPO / "corona-13_amd" / "host" / "libcorona_host.so", REPO / "corona-13_amd" / "csrc" / "libcorona_mi.so",
             REPO / "corona-13_amd" / "host" / "corona-mi", REPO / "corona-13_amd" / "host" / "pfmdiff-mi"]
    if not all(p.exists() for p in built):
        subprocess.check_call(["make", "-C", str(REPO / "corona-13_amd")])
    if not (REPO / "oracle" / "liboracle.so").exists():
        subprocess.check_call(["make", "-C", str(REPO / "oracle"), "liboracle.so"])
    # generated scene data: the finer backdrop of scenes/0054_fine (deterministic, checked by hash)
    fine = REPO / "scenes" / "geo" / "plane_fine.geo"
    if not fine.exists():
        subprocess.check_call([sys.executable, str(REPO / "tools" / "make_geo.py"), "subdivide",
                               str(REPO / "scenes" / "geo" / "plane.geo"), str(fine), "2"])
    import hashlib
    digest = hashlib.sha256(fine.read_bytes()).hexdigest()
    assert digest == "5ab97b31c13baef90daac0af6367b1fae8e764d591f5da9e895a698a19d6410f", "scenes/geo/plane_fine.geo differs from the file the goldens were made with"
    # ... and the 262 144-quad backdrop of scenes/0064_large (27 MB, 0.7 s to generate): a tree of which only the top fits LDS
    large = REPO / "scenes" / "geo" / "plane_k8.geo"
    if not large.exists():
        subprocess.check_call([sys.executable, str(REPO / "tools" / "make_geo.py"), "subdivide",
                               str(REPO / "scenes" / "geo" / "plane.geo"), str(large), "8"])
    digest = hashlib.sha256(large.read_bytes()).hexdigest()
    assert digest == "42233bc5affc7449029a09de780bc65b7e7450ebd8c214d4c6d9fc6590a1c166", "scenes/geo/plane_k8.geo differs from the file the tests were written with"
