import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _libsnerf_built():
    """Host-side tests (struct layouts, hash-grid layout arithmetic, ABI surface) load the library without a GPU: build it when it is
    not there yet (hipcc cross-compiles gfx950 on the CPU-only container), whatever order the test files run in."""
    from soccernerfs_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" and a.shape != () else a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
