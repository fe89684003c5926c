"""What the ABI refuses and what it accepts like the reference: status codes, lifetime rules, NaN / inf inputs, exact ties,
fs == 0 (mod.rs:54-56), whole-page registration.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from gpu_common import FS

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


# ------------------------------------------------------------- C-ABI error paths --
def test_capi_error_codes(eng):
    """Status codes instead of panics (xcor_rustfft.rs:54-55 asserts, mod.rs unwraps):
    every failure returns a code and leaves a message in caf_last_error_string()."""
    import ctypes
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    lib = eng.lib
    fr = np.array([0.0, 1.0, 2.0])
    dp = fr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    h = ctypes.c_void_p()
    # bad shard range / dtype / length
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, _lib.CAF_C128, 2, 1, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, _lib.CAF_C128, 0, 4, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, 7, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4095, dp, 3, FS, _lib.CAF_C128, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_LENGTH
    assert lib.caf_plan_create(eng._h, 0, dp, 3, FS, _lib.CAF_C128, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_LENGTH
    assert b"power of two" in lib.caf_last_error_string()
    assert h.value is None
    # (fs == 0 is NOT an error: the reference accepts it, mod.rs:54-56 -- test_fs_zero_follows_the_reference below)
    # NULL arguments to the device entry point
    plan = eng.plan(4096, fr, FS)
    assert lib.caf_surface_dev(plan._h, None, None, 1, None, None, None, None) == _lib.CAF_ERR_BAD_ARG
    # an empty shard is legal: no rows, peak = "no row"
    empty = eng.plan(4096, fr, FS, row_begin=3, row_end=3)
    assert empty.rows == 0
    empty.close()
    # streaming: bad slot / bad slot count
    with pytest.raises(caf.CafError):
        caf.Stream(plan, batch=1, nslots=1)
    st = caf.Stream(plan, batch=1, nslots=2, want_surface=False)
    assert lib.caf_stream_submit(st._h, 5) == _lib.CAF_ERR_BAD_ARG
    assert st.surface_ptr(0) == 0
    st.close()
    plan.close()
    # views: bad view id, n == 0
    buf = np.zeros((1, 16))
    assert lib.caf_surface_view(eng._h, _lib.CAF_C128, buf.ctypes.data, 1, 8, 9, buf.ctypes.data) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_surface_view(eng._h, _lib.CAF_C128, buf.ctypes.data, 1, 0, 1, buf.ctypes.data) == _lib.CAF_ERR_LENGTH
    # context for a device that does not exist
    h2 = ctypes.c_void_p()
    assert lib.caf_ctx_create(999, ctypes.byref(h2)) == _lib.CAF_ERR_NO_DEVICE


# ------------------------------------------------------------- (c) NaN / Inf inputs --
@pytest.mark.parametrize("n,dtype", [(4096, "c128"), (4096, "c64"), (64, "c128"), (32768, "c64")])
def test_nan_inputs_never_win(n, dtype, eng, oracle, coracle):
    """mod.rs:143-151: `if mag > max` is false for a NaN, so a NaN never becomes the row maximum; a
    NaN sample reaches every bin of the DFT, so every row is (idx 0, val 0.0) and find_peak returns
    its initial (0.0, 0) (mod.rs:32-35).  Checked on the fused, generic and tiled paths, NaN in the
    needle and NaN in the haystack, and against both oracles on the small case."""
    rng = np.random.default_rng(3)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    b = np.roll(a, 5)
    fr = np.array([-3.0, 0.0, 2.5])
    for which in ("needle", "haystack"):
        x, y = a.copy(), b.copy()
        (x if which == "needle" else y)[n // 3] = complex(np.nan, 1.0)
        surf, ridx, rval, peak = eng.surface_arrays(x, y, fr, FS, dtype=dtype)
        assert np.isnan(surf).all(), f"{which}: a NaN sample reaches every lag"
        assert not ridx.any() and not rval.any()
        assert (peak.freq, peak.idx, peak.val, peak.row) == (0.0, 0, 0.0, -1)
        if n == 64:
            osurf, oidx, oval = oracle.np_caf_surface(x, y, fr, FS)
            assert np.isnan(osurf).all() and not oidx.any() and not oval.any()
            csurf, cidx, cval = coracle.caf_surface(x, y, fr, FS, hoist=False, nthreads=1)
            assert not cidx.any() and not cval.any() and coracle.find_peak(fr, cidx, cval) == (0.0, 0)


def test_inf_input_matches_oracle_semantics(eng, oracle):
    """An infinite sample: the mixer turns (inf, 0)*(c, s) into (inf|nan, inf|nan) (mod.rs:57), the
    transforms then mix +inf and -inf into NaN.  Which bins end up inf and which NaN depends on the
    FFT's operation order (rustfft's is not pinned), so the checked contract is the reference's
    comparison rule itself: the reported row peak is the FIRST lag holding the row's largest
    non-NaN value if that is > 0, else (0, 0.0); and find_peak takes the first strictly greater row."""
    rng = np.random.default_rng(4)
    n = 4096
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = np.roll(a, 9)
    a[17] = complex(np.inf, 0.0)
    fr = np.array([0.0, 1.0, -7.5])
    surf, ridx, rval, peak = eng.surface_arrays(a, b, fr, FS)
    for r in range(len(fr)):
        cand = np.where(np.isnan(surf[r]), -np.inf, surf[r])
        k = int(np.argmax(cand))
        if cand[k] > 0.0:
            assert (int(ridx[r]), rval[r]) == (k, cand[k])
        else:
            assert (int(ridx[r]), rval[r]) == (0, 0.0)
    bf, bi = oracle.np_find_peak(fr, ridx, rval)
    assert (peak.freq, int(peak.idx)) == (bf, bi)


def test_find_peak_ignores_nan_rows(eng):
    """caf_find_peak on caller-held rows containing NaN / inf peaks (mod.rs:36: NaN > x is false)."""
    from caf_cookoff_amd import CafSurfaceRow
    rows = [CafSurfaceRow(1.0, None, 10, float("nan")), CafSurfaceRow(2.0, None, 20, 5.0),
            CafSurfaceRow(3.0, None, 30, float("nan")), CafSurfaceRow(4.0, None, 40, float("inf")),
            CafSurfaceRow(5.0, None, 50, float("inf"))]
    assert eng.find_peak(rows) == (4.0, 40)        # first +inf row; the later equal one does not replace it
    assert eng.find_peak(rows[:3]) == (2.0, 20)
    assert eng.find_peak([rows[0], rows[2]]) == (0.0, 0)


def test_lifetime_rules(eng):
    """caf_plan_destroy refuses while a caf_stream of the plan is alive (its graphs hold the plan's
    buffers); the Python wrappers close streams before plans and plans before the context."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    fr = np.array([0.0, 1.0])
    plan = eng.plan(4096, fr, FS)
    st = caf.Stream(plan, batch=1, nslots=2, want_surface=False)
    assert eng.lib.caf_plan_destroy(plan._h) == _lib.CAF_ERR_STATE
    assert b"caf_stream" in eng.lib.caf_last_error_string()
    a, b = st.buffers(0)
    a[:] = 1.0
    b[:] = 1.0
    st.submit(0)
    peaks, _, _ = st.wait(0, want_rows=False)
    assert int(peaks[0]["idx"]) == 0 and peaks[0]["freq"] == 0.0
    plan.close()            # closes the stream first
    assert st._h is None and plan._h is None
    # a second engine: closing it closes its plans
    e2 = caf.Engine(0)
    p2 = e2.plan(64, fr, FS)
    e2.close()
    assert p2._h is None


# ------------------------------------------------------ in-row exact ties (mod.rs:148-151) --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [4096, 1024, 64])
def test_in_row_exact_tie_lowest_lag_wins(dtype, n, eng):
    """mod.rs:148-151 scans a row with a strict '>': among bit-equal maxima the LOWEST lag wins.  The wave and
    workgroup reductions of the HIP path (wave_arg_reduce_maxmin, arg_merge) must do the same.  A delta needle
    makes row 0 a copy of the haystack's magnitudes, two equal spikes make two (nearly) equal maxima; trials
    with random spike positions / amplitudes are run until bit-equal pairs have been seen -- the test FAILS if
    none of 96 trials produced one, so the tie branch cannot go untested silently."""
    rng = np.random.default_rng(77)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    ties = 0
    for trial in range(96):
        d = np.zeros(n, dtype=cdt)
        d[0] = 1.0
        h = np.zeros(n, dtype=cdt)
        a, b = sorted(int(v) for v in rng.choice(n, size=2, replace=False))
        amp = float(rng.choice([1.0, 2.0, 0.5, 3.0, 1.5]))
        ph = np.exp(1j * rng.uniform(0, 2 * np.pi)) if trial % 3 == 2 else 1.0
        h[a] = amp * ph
        h[b] = amp * ph
        if trial % 4 == 3:                      # hi half of the lag axis too: negative lags via a shifted delta
            d[:] = 0
            d[n // 2] = 1.0
        surf, ridx, rval, _ = eng.surface_arrays(d, h, np.array([0.0]), FS, dtype=dtype)
        row = surf[0]
        mx = row.max()
        where = np.flatnonzero(row == mx)
        assert int(ridx[0]) == int(where[0]) and rval[0] == mx          # first lag attaining the maximum, always
        if len(where) >= 2:
            ties += 1
            assert int(ridx[0]) == int(where.min())
    assert ties >= 3, f"only {ties} bit-equal in-row ties in 96 trials: the tie branch was not exercised"


# ------------------------------------------------------ fs == 0: accepted like the reference accepts it --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 64, 1024, 4096, 32768])
def test_fs_zero_follows_the_reference(n, dtype, eng, oracle, coracle):
    """mod.rs:54-56 with fs = 0: dt = inf, NaN phasors, every lag of every row NaN, rows keep (idx 0, val 0.0)
    (mod.rs:143-151) and find_peak returns (0.0, 0) (mod.rs:32-41) -- through every kernel family (lane-group rows, chain
    R = 2 / 4, the tuned one-launch surface, and the batched device path), equal to what the oracle computes."""
    import torch
    import warnings
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    y = np.roll(x, 1 if n > 1 else 0)
    fr = np.array([0.0, 12.5, -3.0, 1e6, 0.0])
    surf, ridx, rval, pk = eng.surface_arrays(x, y, fr, 0, dtype=dtype)
    assert np.isnan(surf).all() and not ridx.any() and not rval.any()
    assert (pk.val, pk.freq, pk.idx, pk.row) == (0.0, 0.0, 0, -1)
    if n <= 4096:      # the oracle on the same input (f64): same NaN pattern, same records, same answer
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, 0)
        csurf, cidx, cval = coracle.caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, 0, want_surface=True)
        assert np.array_equal(np.isnan(osurf), np.isnan(surf)) and np.array_equal(np.isnan(csurf), np.isnan(surf))
        assert np.array_equal(oidx, ridx) and np.array_equal(cidx, ridx) and not oval.any() and not cval.any()
        assert oracle.np_find_peak(fr, oidx, oval) == coracle.find_peak(fr, cidx, cval) == (pk.freq, pk.idx) == (0.0, 0)
    # device-pointer path, batch of 3, rows [1, 4)
    plan = eng.plan(n, fr, 0, dtype=dtype, row_begin=1, row_end=4)
    tdt = torch.float64 if dtype == "c128" else torch.float32
    dn = torch.from_numpy(np.stack([x, y, x])).cuda()
    ds = torch.zeros((3, 3, 2 * n), dtype=tdt, device="cuda")
    di = torch.full((3, 3), 7, dtype=torch.int64, device="cuda")
    dv = torch.full((3, 3), 7, dtype=tdt, device="cuda")
    dp = torch.zeros((3, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(dn.data_ptr(), dn.data_ptr(), 3, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    eng.synchronize()
    assert bool(torch.isnan(ds).all()) and not bool(di.any()) and not bool(dv.any())
    pkd = dp.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    assert not pkd["val"].any() and not pkd["freq"].any() and not pkd["idx"].any() and (pkd["row"] == -1).all()
    plan.close()


@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_apply_freq_shift_with_fs_zero(dtype, eng, oracle, coracle):
    """mod.rs:57-61 with NaN phasors: sample 0 is multiplied by the initial 1 + 0j and stays, every later sample is NaN + NaN j
    (also for f = 0, where the phase is 0 * inf = NaN); n = 0 and n = 1 included."""
    rng = np.random.default_rng(4)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    x = (rng.standard_normal(300) + 1j * rng.standard_normal(300)).astype(cdt)
    for f in (5.0, 0.0, -1e9):
        out = eng.apply_freq_shift(x, f, 0)
        assert out[0] == x[0] and np.isnan(out[1:].real).all() and np.isnan(out[1:].imag).all()
        ref = coracle.apply_freq_shift(x.astype(np.complex128), f, 0)
        assert ref[0] == x[0] and np.array_equal(np.isnan(ref.view(np.float64)), np.isnan(out.astype(np.complex128).view(np.float64)))
    assert eng.apply_freq_shift(x[:1], 5.0, 0)[0] == x[0] and len(eng.apply_freq_shift(x[:0], 5.0, 0)) == 0
    # and a finite phase still leaves sample 0 bit-exact, the rest as before
    got = eng.apply_freq_shift(x, 12.5, FS)
    want = oracle.np_apply_freq_shift(x.astype(np.complex128), 12.5, FS)
    assert got[0] == x[0] and np.max(np.abs(got - want)) <= (1e-12 if dtype == "c128" else 2e-6)


@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_apply_freq_shift_non_finite_sample_zero_follows_the_reference(dtype, eng, coracle):
    """mod.rs:57-60 multiplies sample 0 by the recurrence's initial 1 + 0j with a FULL complex multiply: an inf or NaN component
    spreads exactly as (a*1 - b*0) + (a*0 + b*1) j says -- (inf + 1j) becomes inf + NaN j -- it is not passed through
    (the round-5 kernel returned in[0] unchanged).  Finite samples are untouched by the same arithmetic."""
    cdt = np.complex128 if dtype == "c128" else np.complex64
    rng = np.random.default_rng(12)
    for z0 in (complex(np.inf, 1.0), complex(2.0, -np.inf), complex(np.nan, 3.0), complex(-np.inf, np.inf), complex(-0.0, 0.0)):
        x = (rng.standard_normal(64) + 1j * rng.standard_normal(64)).astype(cdt)
        x[0] = z0
        for fs in (FS, 0):
            got = eng.apply_freq_shift(x, 7.5, fs).astype(np.complex128)
            with np.errstate(invalid="ignore"):
                want = coracle.apply_freq_shift(x.astype(np.complex128), 7.5, fs)
            g0, w0 = got[:1].view(np.float64), want[:1].view(np.float64)
            assert np.array_equal(np.isnan(g0), np.isnan(w0)) and np.array_equal(g0[~np.isnan(g0)], w0[~np.isnan(w0)]), (z0, fs, got[0], want[0])
            if fs:
                assert np.max(np.abs(got[1:] - want[1:])) <= (1e-12 if dtype == "c128" else 2e-6)


def test_multi_surface_with_fs_zero(eng):
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(8)
    x = rng.standard_normal(4096) + 1j * rng.standard_normal(4096)
    fr = np.arange(6.0)
    ms = caf.MultiSurface([0, 0], 4096, fr, 0)
    surf, ridx, rval, pk = ms.run(x, x)
    assert np.isnan(surf).all() and not ridx.any() and not rval.any() and (pk["val"], pk["freq"], int(pk["idx"]), int(pk["row"])) == (0.0, 0.0, 0, -1)
    ridx, rval, peaks = ms.run_batch(np.stack([x, x]), np.stack([x, x]))
    assert not ridx.any() and not rval.any() and (peaks["row"] == -1).all() and not peaks["val"].any()
    ms.close()


# ------------------------------------------------------ caf_host_register: whole pages only --
def test_host_register_takes_whole_pages_only(eng):
    """include/caf_hip.h: a registered range consists of whole pages, so that it can never share a page with unrelated heap
    objects (the reference's rows are callee-owned Vecs, mod.rs:156-161: the arena exists only on this side of the ABI).  A heap
    block, a ragged size and a ragged start are refused with CAF_ERR_BAD_ARG before the runtime sees them; the same rule holds
    for the multi-device registration."""
    import ctypes
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    from gpu_common import PAGE, _mmap_array
    lib = eng.lib
    heap = np.zeros(5000, dtype=np.float64)                        # malloc'ed: 16-byte aligned at best, 40 000 bytes
    assert lib.caf_host_register(eng._h, ctypes.c_void_p(heap.ctypes.data), heap.nbytes) == _lib.CAF_ERR_BAD_ARG
    assert b"whole number" in lib.caf_last_error_string()
    buf = _mmap_array((4 * PAGE // 8,), np.float64)                # four pages, page-aligned
    assert buf.ctypes.data % PAGE == 0
    assert lib.caf_host_register(eng._h, ctypes.c_void_p(buf.ctypes.data), 3 * PAGE + 8) == _lib.CAF_ERR_BAD_ARG      # ragged size
    assert lib.caf_host_register(eng._h, ctypes.c_void_p(buf.ctypes.data + 64), 2 * PAGE) == _lib.CAF_ERR_BAD_ARG     # ragged start
    with pytest.raises(caf.CafError):
        eng.host_register(buf[1:])
    eng.host_register(buf[: 2 * PAGE // 8])                        # whole pages: accepted
    eng.host_unregister(buf[: 2 * PAGE // 8])
    assert not eng._registered
    ms = caf.MultiSurface([0, 0], 64, np.array([0.0, 1.0]), FS)
    assert ms.lib.caf_multi_surface_host_register(ms._h, ctypes.c_void_p(heap.ctypes.data), heap.nbytes) == _lib.CAF_ERR_BAD_ARG
    assert ms.lib.caf_multi_surface_host_register(ms._h, ctypes.c_void_p(buf.ctypes.data), 2 * PAGE) == _lib.CAF_OK
    assert ms.lib.caf_multi_surface_host_unregister(ms._h, ctypes.c_void_p(buf.ctypes.data)) == _lib.CAF_OK
    ms.close()
