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
    _install_abort_trace()


def _install_abort_trace():
    """tests/cpp/abort_trace.c: if a native library under the tests calls abort(), print the native stack before
    faulthandler's Python stacks.  Installed after pytest's own faulthandler set-up, so this
    handler runs first and then hands over to it.  Missing helper (not built): nothing happens."""
    import ctypes
    so = ROOT / "tests" / "cpp" / "abort_trace.so"
    if so.exists():
        try:
            ctypes.CDLL(str(so)).abort_trace_install()
        except OSError:
            pass


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
