"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/caf_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import caf_cookoff_amd
    return caf_cookoff_amd.load()


def declared_symbols():
    text = (ROOT / "include" / "caf_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(caf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/caf_hip.h but not exported"


def test_python_binding_covers_header(lib):
    from caf_cookoff_amd import _lib
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(declared_symbols())


def test_abi_version_and_struct_layout(lib):
    from caf_cookoff_amd import CafPeak
    assert lib.caf_abi_version() == 5
    assert ctypes.sizeof(CafPeak) == 32


def test_no_gpu_fails_loudly(lib):
    """Without a device every context creation must fail with CAF_ERR_NO_DEVICE
    (the product path never falls back to the CPU).  Skipped on a GPU box."""
    import caf_cookoff_amd as caf
    if lib.caf_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(caf.CafError) as ei:
        caf.Engine(0)
    assert ei.value.code == 5
    assert "no CPU fallback" in str(ei.value)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under caf_cookoff_amd/ or include/ may use it."""
    for p in list((ROOT / "caf_cookoff_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in {".py", ".hip", ".hpp", ".h", ".cpp", ""} and p.name != "libcaf_hip.so":
            try:
                t = p.read_text()
            except UnicodeDecodeError:
                continue
            assert "oracle" not in t.lower() or p.name == "Makefile", f"{p} mentions the oracle"


def test_header_is_clean_c11():
    """include/caf_hip.h is what a C, cgo or bindgen host includes: it must compile as strict C (DESIGN.md section 10 has claimed
    this check since round 4; VERDICT r05 weak #13 found no test doing it) -- and as C++ too."""
    import subprocess
    hdr = str(ROOT / "include" / "caf_hip.h")
    for cmd in (["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr


def test_timeout_status_and_setter_are_part_of_abi_5(lib):
    """ABI 5: CAF_ERR_TIMEOUT and caf_multi_surface_set_timeout (argument checks need no GPU)."""
    from caf_cookoff_amd import _lib
    hdr = (ROOT / "include" / "caf_hip.h").read_text()
    assert re.search(r"CAF_ERR_TIMEOUT = 8\b", hdr) and _lib.CAF_ERR_TIMEOUT == 8
    assert lib.caf_multi_surface_set_timeout(None, 1.0) == _lib.CAF_ERR_BAD_ARG
    assert b"caf_multi_surface_set_timeout" in lib.caf_last_error_string()
