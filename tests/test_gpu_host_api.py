"""The literal drop-in calls with host pointers (caf_surface_c128 / _c64): plan cache, in-place surfaces in caf_host_alloc /
caf_host_register memory, the timing binary, a short soak.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import ctypes
import json
import subprocess

import numpy as np
import pytest

from gpu_common import PAGE, _guard_rows, FS, ROOT, TOL32, TOL64, _mmap_array, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


# ------------------------------------------------------ host-pointer API: cached plans --
def test_host_api_alternating_shapes_through_the_plan_cache(eng, oracle):
    """main.rs:25-26 / tests/test.rs:25-26 call caf_surface with whatever (n, freq list) they like; the context
    keeps the four most recently used plans.  SIX different (n, freq list, dtype) combinations (more than the
    cache holds) in an interleaved order, three rounds: every call is checked against the oracle and a repeated
    call returns the same bits whether its plan was still cached or had been evicted and rebuilt."""
    rng = np.random.default_rng(31)
    shapes = [(4096, np.arange(-20.0, 20.0, 0.5), "c128"), (1024, np.arange(0.0, 50.0, 1.0), "c128"),
              (4096, np.arange(-5.0, 5.0, 0.25), "c128"), (4096, np.arange(-20.0, 20.0, 0.5), "c64"),
              (64, np.array([0.0, 10.0, 20.0]), "c128"), (8192, np.arange(10.0, 14.0, 0.5), "c64")]
    cases = []
    for n, fr, dt in shapes:
        x, y = _planted(rng, n, FS, float(fr[len(fr) // 3]), int(rng.integers(1, n // 4)),
                        np.complex128 if dt == "c128" else np.complex64)
        osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
        cases.append((n, fr, dt, x, y, osurf, oidx, oracle.np_find_peak(fr, oidx, oval)))
    first = {}
    order = [0, 1, 0, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 5, 4, 3, 2, 1, 0]
    for k in order:
        n, fr, dt, x, y, osurf, oidx, opk = cases[k]
        surf, ridx, rval, peak = eng.surface_arrays(x, y, fr, FS, dtype=dt)
        tol = (TOL64 if dt == "c128" else TOL32) * osurf.max()
        assert np.max(np.abs(surf - osurf)) <= tol, f"shape {k}"
        assert (peak.freq, int(peak.idx)) == opk, f"shape {k}"
        if k in first:
            assert np.array_equal(first[k][0], surf) and np.array_equal(first[k][1], ridx) and np.array_equal(first[k][2], rval)
        else:
            first[k] = (surf.copy(), ridx.copy(), rval.copy())
        # peaks-only call of the same shape: same rows without the surface
        _, ridx2, rval2, peak2 = eng.surface_arrays(x, y, fr, FS, want_surface=False, dtype=dt)
        assert np.array_equal(ridx, ridx2) and np.array_equal(rval, rval2) and (peak2.freq, peak2.idx) == (peak.freq, peak.idx)


@pytest.mark.parametrize("dtype,n", [("c128", 4096), ("c64", 4096), ("c128", 1024), ("c128", 64)])
def test_host_surface_in_place_equals_copied(dtype, n, eng, oracle):
    """A surface written in place by the row kernel (caf_host_alloc / caf_host_register memory) holds the same
    bits as one copied back from the device slab (pageable destination), for the one-launch n = 4096 path, a
    chain path and the generic path; a destination that only partly lies in registered memory takes the copy."""
    rng = np.random.default_rng(5)
    fr = np.arange(-10.0, 10.0, 0.5)
    cdt, rdt = (np.complex128, np.float64) if dtype == "c128" else (np.complex64, np.float32)
    x, y = _planted(rng, n, FS, 3.0, n // 8, cdt)
    F, L = len(fr), 2 * n
    ref, ridx0, rval0, pk0 = eng.surface_arrays(x, y, fr, FS, dtype=dtype)
    osurf, _, _ = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
    assert np.max(np.abs(ref - osurf)) <= (TOL64 if dtype == "c128" else TOL32) * osurf.max()
    pinned = eng.host_empty((F, L), rdt)
    pinned[:] = -1.0
    out, ridx, rval, pk = eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=pinned)
    assert out is pinned and np.array_equal(pinned, ref) and np.array_equal(ridx, ridx0) and pk.idx == pk0.idx
    # a sub-range of a larger pinned arena
    arena = eng.host_empty((3 * F, L), rdt)
    arena[:] = -1.0
    mid = arena[F:2 * F]
    eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=mid)
    assert np.array_equal(mid, ref) and (arena[:F] == -1.0).all() and (arena[2 * F:] == -1.0).all()
    # caller-owned memory, registered once: whole pages only, so the fences either side are whole pages of rows
    g = _guard_rows(L * np.dtype(rdt).itemsize)
    own = _mmap_array((F + 2 * g, L), rdt, -1.0)
    assert own[g:F + g].nbytes % PAGE == 0 and own[g:F + g].ctypes.data % PAGE == 0
    eng.host_register(own[g:F + g])
    eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=own[g:F + g])
    assert np.array_equal(own[g:F + g], ref) and (own[:g] == -1.0).all() and (own[F + g:] == -1.0).all()
    # half inside, half outside the registration: falls back to the copy, same bits
    own[:] = -1.0
    eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=own[g + 1:F + g + 1])
    assert np.array_equal(own[g + 1:F + g + 1], ref)
    eng.host_unregister(own[g:F + g])
    del pinned, arena, mid


def test_host_memory_api_errors(eng):
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = eng.lib
    p = ctypes.c_void_p()
    assert lib.caf_host_alloc(eng._h, 0, ctypes.byref(p)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_host_alloc(None, 64, ctypes.byref(p)) == _lib.CAF_ERR_BAD_ARG
    buf = _mmap_array((4096,), np.float64)
    assert lib.caf_host_unregister(eng._h, ctypes.c_void_p(buf.ctypes.data)) == _lib.CAF_ERR_BAD_ARG
    assert b"caf_host_register" in lib.caf_last_error_string()
    eng.host_register(buf)
    with pytest.raises(caf.CafError) as ei:
        eng.host_register(buf)
    assert ei.value.code == _lib.CAF_ERR_STATE
    assert lib.caf_host_free(eng._h, ctypes.c_void_p(buf.ctypes.data)) == _lib.CAF_ERR_BAD_ARG  # not from caf_host_alloc
    eng.host_unregister(buf)
    assert lib.caf_host_free(eng._h, None) == _lib.CAF_OK
    a = eng.host_empty((8,), np.float64)
    a[:] = 3.0
    assert a.sum() == 24.0
    # a context destroyed with live host memory frees it itself
    e2 = caf.Engine(0)
    q = ctypes.c_void_p()
    assert e2.lib.caf_host_alloc(e2._h, 1 << 20, ctypes.byref(q)) == 0 and q.value
    e2.host_register(buf)
    e2.close()


def test_host_api_timing_binary_runs_and_reports(eng):
    """tests/cpp/host_api_time (what bench.py's extra.host_api runs): builds, checks its own results (peak,
    surface bits equal across the three destinations) and reports sane numbers.  Loose bounds only: the exact
    figures belong to the bench line."""
    # (tests/cpp only -- never the library's own Makefile from a process that has libcaf_hip.so mapped)
    subprocess.run(["make", "-C", str(ROOT / "tests" / "cpp")], check=True, capture_output=True)
    r = subprocess.run([str(ROOT / "tests" / "cpp" / "host_api_time"), "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout)
    print(j)
    assert 5.0 < j["peaks_only_us"] < 500.0 and 0.2 < j["with_surface_ms"] < 5.0 and j["apply_shift_4096_us"] < 500.0
    assert j["with_surface_in_place_ms"] <= j["with_surface_ms"] * 1.5


def test_short_host_api_soak():
    """tools/host_api_soak.py, short form: 4000 polled host-pointer calls per dtype, every result bit-equal to round 0."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("host_api_soak", ROOT / "tools" / "host_api_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(4000, "c128", log=False) == 0
    assert mod.run(4000, "c64", log=False) == 0


def test_engine_close_refuses_under_live_host_memory():
    """Engine.host_empty arrays keep their Engine alive, and close() refuses while one of them exists (the context owns the
    pinned memory under the array: closing would turn every later access into a use-after-free)."""
    import gc
    import caf_cookoff_amd as caf
    e = caf.Engine(0)
    a = e.host_empty((4, 8), np.float64)
    view = a[1:3]
    with pytest.raises(RuntimeError, match="still alive"):
        e.close()
    del a
    with pytest.raises(RuntimeError):   # a view keeps the buffer alive too
        e.close()
    view[:] = 1.0                       # ... and the memory is still there
    del view
    gc.collect()
    e.close()
