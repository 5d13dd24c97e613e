import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is a build artefact (git-ignored): (re)build it before any test imports it.  make is
    # idempotent and tracks the headers, so an edited .hip / .h can never be tested against a stale library.
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "super_primitive_amd", "csrc", "libsp_hip.so")
    if shutil.which("hipcc") or not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.dirname(lib), "-j8"], check=True)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def unpack_masks(g):
    H, W, N = (int(v) for v in g["in_HWN"])
    return np.unpackbits(g["in_masks"], axis=-1, count=W).astype(bool).reshape(N, H, W)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
