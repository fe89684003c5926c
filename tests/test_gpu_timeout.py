"""caf_multi_surface_set_timeout / caf_multi_stream_set_timeout (ABI 5): a multi-device call must not wait for ever for a device
that does not answer.
The reference's join panics on a dead worker (`rx.recv().unwrap()`, mod.rs:452-457); a GPU worker can stay silent instead.
The MEASUREMENT build can make one: caf_debug_multi_stall(worker, ms, at) puts a kernel that sleeps for `ms` milliseconds
(it ends by itself) on that worker's stream ahead of its row launch (at = 0), ahead of its part of the RCCL join (at = 1), or on
its first slot stream ahead of the first replay of a caf_multi_stream_run (at = 2).
Two workers on the one GPU of this box (device ids 0, 0)."""
import ctypes
import time

import numpy as np
import pytest

from gpu_common import FS

pytestmark = [pytest.mark.gpu]


def _measure_lib():
    import caf_cookoff_amd as caf
    lib = caf.load(caf.MEASURE_LIB_PATH)
    lib.caf_debug_multi_stall.restype = ctypes.c_int
    lib.caf_debug_multi_stall.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return lib


def _batch(B):
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    nd, hs, lags, _ = make_batch(B, 4096, FS, seed0=9100)
    return caf.bench_shifts(), nd, hs, np.asarray(lags)


def _destroy(ms):
    """caf_multi_surface_destroy with its status and duration (MultiSurface.close() drops the status)"""
    t0 = time.perf_counter()
    rc = ms.lib.caf_multi_surface_destroy(ms._h)
    ms._h = None
    return rc, time.perf_counter() - t0


def test_a_silent_worker_ends_the_call_with_timeout_then_state_and_destroy_does_not_wait(meng):
    """Worker 1's stream sleeps for 5 s; the deadline is 1.5 s.  The batch call returns CAF_ERR_TIMEOUT within the bound and
    names worker 1 and its device; every later run of the object is CAF_ERR_STATE; caf_multi_surface_destroy gives the device
    its 2 s to drain, then leaves the workers of that device behind and says so -- it does not wait for the sleeper."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = _measure_lib()
    fr, nd, hs, lags = _batch(8)
    ms = caf.MultiSurface([0, 0], 4096, fr, FS, surface_on_device=True, lib=caf.MEASURE_LIB_PATH)
    _, _, want = ms.run_batch(nd, hs, want_rows=False)
    assert np.array_equal(want["idx"], lags)
    # a deadline changes nothing about a healthy call: polled waits, same bits
    ms.set_timeout(1.5)
    for _ in range(3):
        _, _, got = ms.run_batch(batch=8, want_rows=False)
        assert got.tobytes() == want.tobytes()
    _, _, _, pk1 = ms.run(nd[0], hs[0], want_surface=False)
    assert int(pk1["idx"]) == lags[0]
    _, _, got = ms.run_batch(nd, hs, want_rows=False)          # (the single-surface run replaced nothing of the batch state)
    assert got.tobytes() == want.tobytes()
    assert lib.caf_debug_multi_stall(1, 5000, 0) == 0
    t_stall = time.perf_counter()
    with pytest.raises(caf.CafError) as ei:
        ms.run_batch(batch=8, want_rows=False)
    took = time.perf_counter() - t_stall
    assert ei.value.code == _lib.CAF_ERR_TIMEOUT, str(ei.value)
    assert "worker 1 (device 0" in str(ei.value) and "did not finish" in str(ei.value), str(ei.value)
    assert 1.4 <= took < 1.5 + 2.5, took
    for call in (lambda: ms.run_batch(batch=8, want_rows=False), lambda: ms.run(nd[0], hs[0], want_surface=False),
                 lambda: ms.run_batch(nd, hs, want_rows=False)):
        with pytest.raises(caf.CafError) as e2:
            call()
        assert e2.value.code == _lib.CAF_ERR_STATE and "timeout" in str(e2.value)
    rc, took_destroy = _destroy(ms)
    assert rc == _lib.CAF_ERR_TIMEOUT and took_destroy < 3.5, (rc, took_destroy)    # 2 s of grace, not the sleeper's 5 s
    assert b"left behind" in lib.caf_last_error_string()
    # the sleeper ends by itself; let the device drain before anything else uses it
    time.sleep(max(0.0, 5.5 - (time.perf_counter() - t_stall)))
    meng.synchronize()
    import torch
    torch.cuda.synchronize()


def test_a_short_stall_times_out_but_destroy_releases_everything(meng):
    """Same with a 2.2 s sleeper and a 1 s deadline on the single-surface call: CAF_ERR_TIMEOUT, and by the time destroy's
    grace has passed the device has drained: everything is released, status CAF_OK.  A fresh object works afterwards."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = _measure_lib()
    fr, nd, hs, lags = _batch(2)
    ms = caf.MultiSurface([0, 0], 4096, fr, FS, lib=caf.MEASURE_LIB_PATH)
    ms.set_timeout(1.0)
    _, _, _, pk = ms.run(nd[0], hs[0], want_surface=False)
    assert int(pk["idx"]) == lags[0]
    assert lib.caf_debug_multi_stall(0, 2200, 0) == 0             # worker 0 = the calling thread this time
    t0 = time.perf_counter()
    with pytest.raises(caf.CafError) as ei:
        ms.run(nd[1], hs[1], want_surface=False)
    took = time.perf_counter() - t0
    assert ei.value.code == _lib.CAF_ERR_TIMEOUT and "worker 0 (device 0" in str(ei.value), str(ei.value)
    assert 0.9 <= took < 2.2, took
    rc, took_destroy = _destroy(ms)
    assert rc == _lib.CAF_OK and took_destroy < 2.5, (rc, took_destroy)
    ms2 = caf.MultiSurface([0, 0], 4096, fr, FS, lib=caf.MEASURE_LIB_PATH)
    ms2.set_timeout(30.0)
    _, _, _, pk = ms2.run(nd[1], hs[1], want_surface=False)
    assert int(pk["idx"]) == lags[1]
    ms2.close()


def test_a_silent_peer_in_the_rccl_join_times_out(meng):
    """The in-library RCCL join (one rank on this one-GPU box): the sleeper sits ahead of the worker's part of the peak
    exchange; the wait for the exchange is a poll against the same deadline -> CAF_ERR_TIMEOUT naming the worker, the object
    unusable, destroy (the device has drained by then) releases everything including the communicator."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = _measure_lib()
    fr, nd, hs, lags = _batch(4)
    ms = caf.MultiSurface([0], 4096, fr, FS, rccl=True, surface_on_device=True, lib=caf.MEASURE_LIB_PATH)
    ms.set_timeout(1.0)
    _, _, want = ms.run_batch(nd, hs, want_rows=False)
    assert np.array_equal(want["idx"], lags)
    assert lib.caf_debug_multi_stall(0, 2200, 1) == 0
    t0 = time.perf_counter()
    with pytest.raises(caf.CafError) as ei:
        ms.run_batch(batch=4, want_rows=False)
    took = time.perf_counter() - t0
    assert ei.value.code == _lib.CAF_ERR_TIMEOUT and "RCCL peak reduction: worker 0" in str(ei.value), str(ei.value)
    assert 0.9 <= took < 2.2, took
    with pytest.raises(caf.CafError) as e2:
        ms.run_batch(batch=4, want_rows=False)
    assert e2.value.code == _lib.CAF_ERR_STATE
    rc, took_destroy = _destroy(ms)
    assert rc == _lib.CAF_OK and took_destroy < 3.0, (rc, took_destroy)


def test_a_silent_device_under_the_surface_parallel_streams_times_out(meng):
    """caf_multi_stream_run (whole surfaces round-robin, two workers on the one GPU): worker 1's first slot stream sleeps
    2.5 s, the deadline is 1 s -> CAF_ERR_TIMEOUT naming worker 1, later runs CAF_ERR_STATE; destroy: the device has drained
    inside the grace period, everything is released.  Healthy runs under a deadline return the bits of runs without one."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = _measure_lib()
    fr, nd, hs, lags = _batch(37)
    ms = caf.MultiStream([0, 0], 4096, fr, FS, nslots=2, lib=caf.MEASURE_LIB_PATH)
    want, _, _ = ms.run(nd, hs)
    assert np.array_equal(want["idx"], lags)
    ms.set_timeout(1.0)
    got, _, _ = ms.run(nd, hs)
    assert got.tobytes() == want.tobytes()
    assert lib.caf_debug_multi_stall(1, 2500, 2) == 0
    t0 = time.perf_counter()
    with pytest.raises(caf.CafError) as ei:
        ms.run(nd, hs)
    took = time.perf_counter() - t0
    assert ei.value.code == _lib.CAF_ERR_TIMEOUT and "worker 1 (device 0)" in str(ei.value), str(ei.value)
    assert 0.9 <= took < 2.4, took
    with pytest.raises(caf.CafError) as e2:
        ms.run(nd, hs)
    assert e2.value.code == _lib.CAF_ERR_STATE
    t0 = time.perf_counter()
    rc = ms.lib.caf_multi_stream_destroy(ms._h)
    ms._h = None
    assert rc == _lib.CAF_OK and time.perf_counter() - t0 < 3.0, rc
    ms2 = caf.MultiStream([0, 0], 4096, fr, FS, nslots=2)
    again, _, _ = ms2.run(nd, hs)
    assert again.tobytes() == want.tobytes()
    ms2.close()


def test_set_timeout_argument_checks_and_zero_means_none(eng):
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    fr, nd, hs, lags = _batch(2)
    ms = caf.MultiSurface([0, 0], 4096, fr, FS)
    for bad in (-1.0, float("nan"), 1e9):
        with pytest.raises(caf.CafError) as ei:
            ms.set_timeout(bad)
        assert ei.value.code == _lib.CAF_ERR_BAD_ARG
    ms.set_timeout(5.0)
    ms.set_timeout(0.0)   # back to plain blocking waits
    _, _, pk = ms.run_batch(nd, hs, want_rows=False)
    assert np.array_equal(pk["idx"], lags)
    ms.close()
