import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
DATA = GOLDEN / "data"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(GOLDEN / "golden.npz")


@pytest.fixture(scope="session")
def manifest():
    return json.loads((GOLDEN / "manifest.json").read_text())


@pytest.fixture(scope="session")
def oracle():
    from oracle import caf_oracle
    return caf_oracle


@pytest.fixture(scope="session")
def coracle(oracle):
    return oracle.COracle()
