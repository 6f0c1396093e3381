import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name)))


def rel_err(a, b):
    """max|a-b| / max|b|  (the per-tensor metric used for every fp32 parity gate)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.abs(b).max()
    num = np.abs(a - b).max() if a.size else 0.0
    return num / den if den > 0 else num


@pytest.fixture(scope="session")
def golden_dir():
    return GOLD
