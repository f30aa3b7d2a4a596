import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors lifted from the reference's own tests (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN_DIR, "reference_vectors.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_tables():
    with open(os.path.join(GOLDEN_DIR, "reference_tables.json")) as fh:
        return json.load(fh)["tables"]


@pytest.fixture(scope="session", autouse=True)
def _parity_report():
    """XRS_PARITY_REPORT=<path>: write the per-op / per-config error table the GPU tests recorded (tests/parity_log.py)."""
    yield
    from tests import parity_log
    parity_log.flush()
