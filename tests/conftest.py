import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hostsim")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_wav():
    import numpy as np
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "vdl2_model_16b_1050kHz.wav"), dtype=np.uint8)
    return raw


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
