import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wsss-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) libwsscam.so and the C oracle once per session."""
    import __graft_entry__ as ge

    ge.build_lib()
    ge.build_oracle()
    return ge


@pytest.fixture(scope="session")
def ctx(built):
    from wsscam import _lib

    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "resnet50_cam.npz"), allow_pickle=False)
