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
    # A test that blocks inside a HIP call can never be interrupted from Python: with pytest-timeout's "thread" method a
    # watchdog thread dumps every thread's stack (so the log says WHERE it hung) and ends the run instead of letting the
    # whole suite sit until an outer limit kills it without a trace.  15 minutes per test: the first `import torch` on a
    # fresh box alone can take two.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900
        config.option.timeout_method = "thread"


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
