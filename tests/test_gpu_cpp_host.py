"""Builds and runs the C++ host-side mirror of the reference's callers on the GPU:
tests/cpp/test_kats.cpp (== caf_rust/tests/test.rs) and examples/caf_main.cpp
(== caf_rust/src/main.rs:10-32, whose output format is checked)."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def built():
    # Only the C++ hosts are (re)built here, against the libcaf_hip.so that is already in the tree.  Never run the
    # library's own Makefile from a test process: this process has libcaf_hip.so mapped, and a rebuild (file times can
    # shift when the tree is copied to another box) would rewrite the mapped file under the running code.
    assert (ROOT / "caf_cookoff_amd" / "libcaf_hip.so").exists(), "build the HIP library first (__graft_entry__.build())"
    subprocess.run(["make", "-C", str(ROOT / "tests" / "cpp")], check=True, capture_output=True)
    return ROOT / "tests" / "cpp"


def test_cpp_kats(built):
    r = subprocess.run([str(built / "test_kats"), str(ROOT / "tests" / "golden" / "data")], capture_output=True,
                       text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("... ok") == 17 and "test result: ok. 0 failed" in r.stdout


def test_cpp_main_demo(built):
    r = subprocess.run([str(built / "caf_main")], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    # main.rs:29-31 on the chirp_0 pair with the 0.5 Hz grid -> 69.0 Hz, 202 samples, 202/48 ms
    assert r.stdout.splitlines() == ["Frequency offset: 69.0Hz", "Time offset: 202 samples (4.208ms)"]


def test_cpp_bench_loop(built):
    """examples/caf_bench.cpp == benches/caf_bench.rs:150-168 from a compiled host: the literal loop, the peaks-only call and the
    loop as ONE caf_multi_surface_run_batch call per 64 pairs (RCCL join with this box's one GPU); it checks its own answers
    ((69.0, 202) and tau = 202 + b % 64 for the delayed copies) and prints one JSON line."""
    import json
    r = subprocess.run([str(built / "caf_bench"), str(ROOT / "tests" / "golden" / "data"), "64", "5", "rccl"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["batch"] == 64 and line["gpus"] >= 1 and line["peak_join"] == "rccl" and line["call_timeout_s"] == 60
    assert 0 < line["batch_resident_ms_per_call"] <= line["batch_with_upload_ms_per_call"] * 1.5
    assert line["batch_resident_surfaces_per_s"] > 20 * 1e3 / line["literal_loop_ms_per_surface"]   # the batched call is the point
