"""GPU parity tests added in round 3: the rebuilt host-pointer entry points (cached plans, pinned staging,
in-place surfaces), the surface-parallel multi-device driver, the in-row tie rule made unconditional and a
seeded fuzz of every kernel path against the ORACLE (not against the HIP path itself).  Every call goes
through the C ABI (libcaf_hip.so)."""
import ctypes
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

FS = 48000
TOL64 = 1e-6
TOL32 = 1e-3


@pytest.fixture(scope="module")
def eng():
    import caf_cookoff_amd as caf
    assert caf.LIB_PATH.exists(), "HIP extension missing: the product path must not run without it"
    e = caf.Engine(0)
    yield e
    e.close()


def _mmap_array(shape, dtype, fill=0.0):
    """A caller-owned buffer for caf_host_register that is its own anonymous mapping: page-aligned, and returned to the
    kernel (munmap, which tears down every GPU mapping of the range) when the array dies -- not a heap block that goes back
    to malloc and is handed out again, still mapped, as somebody else's copy destination."""
    import mmap
    count = int(np.prod(shape))
    m = mmap.mmap(-1, max(count * np.dtype(dtype).itemsize, 1))
    a = np.frombuffer(m, dtype=dtype, count=count).reshape(shape)
    a[...] = fill
    return a


def _planted(rng, n, fs, f, lag, cdt=np.complex128):
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n) if n >= 8 else \
        (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    y = np.roll(x, lag) * np.exp(2j * np.pi * f * np.arange(n) / fs)
    y[:lag] = 0
    y = y + 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(cdt), y.astype(cdt)


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
    # caller-owned memory, registered once
    own = _mmap_array((F + 2, L), rdt, -1.0)
    eng.host_register(own[1:F + 1])
    eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=own[1:F + 1])
    assert np.array_equal(own[1:F + 1], ref) and (own[0] == -1.0).all() and (own[F + 1] == -1.0).all()
    # half inside, half outside the registration: falls back to the copy, same bits
    own[:] = -1.0
    eng.surface_arrays(x, y, fr, FS, dtype=dtype, out=own[2:F + 2])
    assert np.array_equal(own[2:F + 2], ref)
    eng.host_unregister(own[1:F + 1])
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


# ------------------------------------------------------ surface-parallel multi-device driver --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_multi_stream_two_contexts_on_one_gpu(dtype, eng, oracle):
    """caf_multi_stream_*: whole surfaces round-robin over workers, one host thread each (here: TWO contexts
    on device 0, the closest a one-GPU box gets to two devices).  37 pairs (odd, ragged), results in input
    order, every (tau, f) and every row peak against the oracle; equal to a single caf_stream run of the same form
    (eight surfaces per replay) bit for bit."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    cdt = np.complex128 if dtype == "c128" else np.complex64
    fr = caf.bench_shifts()[::8]
    nd, hs, lags, fos = make_batch(37, 4096, FS, seed0=900, dtype=cdt)
    ms = caf.MultiStream([0, 0], 4096, fr, FS, dtype=dtype, nslots=2)
    assert ms.ndev == 2
    peaks, ridx, rval = ms.run(nd, hs, want_rows=True)
    plan = eng.plan(4096, fr, FS, dtype=dtype)
    # (the workers stream eight surfaces per replay; a one-surface-per-replay stream runs the one-launch kernel, whose
    #  haystack spectrum differs from k_seq_prepare's in the last bit)
    st = caf.Stream(plan, batch=8, nslots=2, want_surface=False)
    p1, i1, v1 = st.run(nd, hs, want_rows=True)
    st.close()
    plan.close()
    assert np.array_equal(peaks, p1) and np.array_equal(ridx, i1) and np.array_equal(rval, v1)
    for k in range(37):
        _, oidx, oval = oracle.np_caf_surface(nd[k].astype(np.complex128), hs[k].astype(np.complex128), fr, FS, want_surface=False)
        of, oi = oracle.np_find_peak(fr, oidx, oval)
        assert (peaks[k]["freq"], int(peaks[k]["idx"])) == (of, oi) and int(peaks[k]["idx"]) == lags[k]
        tol = (TOL64 if dtype == "c128" else TOL32) * oval.max()
        assert np.max(np.abs(rval[k].astype(np.float64) - oval)) <= tol
    # a second run on the same object, fewer pairs than workers, and an empty run
    p2, _, _ = ms.run(nd[:1], hs[:1])
    assert int(p2[0]["idx"]) == lags[0]
    p3, _, _ = ms.run(nd[:0], hs[:0])
    assert len(p3) == 0
    ms.close()


def test_multi_stream_error_propagation(eng):
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    fr = np.array([0.0, 1.0])
    with pytest.raises(caf.CafError) as ei:
        caf.MultiStream([0, 999], 4096, fr, FS)          # second worker's device does not exist
    assert ei.value.code == _lib.CAF_ERR_NO_DEVICE
    with pytest.raises(caf.CafError):
        caf.MultiStream([], 4096, fr, FS)
    with pytest.raises(caf.CafError) as ei:
        caf.MultiStream([0], 4095, fr, FS)
    assert ei.value.code == _lib.CAF_ERR_LENGTH


# ------------------------------------------------------ short inputs: lane-group rows, n = 1 ... 512 --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [8, 64, 512])
def test_small_rows_many_groups_vs_oracle(n, dtype, eng, oracle):
    """k_small_rows with more rows than resident workgroups' worth of one pass and a ragged tail: 3 surfaces x 1031
    rows (a prime; 512 / (n / 8) rows per workgroup), every value of the surfaces against the numpy oracle."""
    import torch
    rng = np.random.default_rng(7000 + n)
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    tol = TOL64 if dtype == "c128" else TOL32
    fs, F, B = 48000, 1031, 3
    fr = np.linspace(-400.0, 400.0, F)
    nd = (rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n))).astype(cdt)
    hs = (rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n))).astype(cdt)
    plan = eng.plan(n, fr, fs, dtype=dtype)
    assert plan.kernel_name.startswith("caf::k_small_rows<")
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.full((B, F, 2 * n), -1.0, dtype=tdt, device="cuda")
    di = torch.zeros((B, F), dtype=torch.int64, device="cuda")
    dv = torch.zeros((B, F), dtype=tdt, device="cuda")
    dp = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    got, gi, gv = ds.cpu().numpy(), di.cpu().numpy(), dv.cpu().numpy()
    for b in range(B):
        ob, oi, ov = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr, fs)
        assert np.max(np.abs(got[b] - ob)) <= tol * ob.max()
        assert np.array_equal(gi[b], np.argmax(got[b], axis=1))                   # first maximum of its own row
        assert np.array_equal(gv[b], got[b][np.arange(F), gi[b]])
    plan.close()


def test_small_rows_without_phasor_table_bit_equal(eng):
    """k_small_rows takes w^tl and w^TPR from a per-plan table up to 256 MiB and runs the two f64 sincos itself beyond
    (same function, same arguments): 270 000 rows at n = 512 (peaks only) are past the limit; 64 of those rows through
    a plan that has the table must give the same bits."""
    import torch
    rng = np.random.default_rng(99)
    n, fs, F = 512, 48000, 270000
    fr = np.linspace(-2000.0, 2000.0, F)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    dn, dh = torch.from_numpy(x[None]).cuda(), torch.from_numpy(y[None]).cuda()
    dp = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
    big = eng.plan(n, fr, fs, dtype="c64")
    bi = torch.zeros((1, F), dtype=torch.int64, device="cuda")
    bv = torch.zeros((1, F), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    big.surface_dev(dn.data_ptr(), dh.data_ptr(), 1, None, bi.data_ptr(), bv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    big.close()
    sel = np.arange(131000, 131064)
    small = eng.plan(n, fr[sel], fs, dtype="c64")
    si = torch.zeros((1, 64), dtype=torch.int64, device="cuda")
    sv = torch.zeros((1, 64), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    small.surface_dev(dn.data_ptr(), dh.data_ptr(), 1, None, si.data_ptr(), sv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    small.close()
    assert np.array_equal(bi.cpu().numpy()[0, sel], si.cpu().numpy()[0])
    assert np.array_equal(bv.cpu().numpy()[0, sel], sv.cpu().numpy()[0])
    assert float(sv.max()) > 0


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512])
def test_small_path_vs_oracle(n, dtype, eng, oracle):
    """kernels_small.hpp ("any power of two", xcor_rustfft.rs:2): every n below the chain kernels' range runs as
    lane-group rows in ONE launch; surface vs the numpy oracle (1e-6 / 1e-3 of the maximum), row argmax where the
    oracle's row has a clear winner, global peak exact; ragged row counts (rows per workgroup = 512 / (n / 8) from n = 8
    on, 256 / max(1, n / 8) below, does not divide them), several surfaces per launch through the device API."""
    import torch
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(1000 + n)
    cdt, rdt, tdt = (np.complex128, np.float64, torch.float64) if dtype == "c128" else (np.complex64, np.float32, torch.float32)
    tol = TOL64 if dtype == "c128" else TOL32
    fs = 48000
    F = 37
    fr = np.linspace(-300.0, 300.0, F)
    fr[5] = 120.0
    lag = 0 if n == 1 else int(rng.integers(0, max(1, n // 2)))
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    y = np.roll(x, lag) * np.exp(2j * np.pi * 120.0 * np.arange(n) / fs)
    y[:lag] = 0
    x, y = x.astype(cdt), y.astype(cdt)
    plan = eng.plan(n, fr, fs, dtype=dtype)
    assert plan.path == "small" and plan.kernel_name.startswith(("caf::k_small<", "caf::k_small_rows<"))
    plan.close()
    surf, ridx, rval, peak = eng.surface_arrays(x, y, fr, fs, dtype=dtype)
    osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, fs)
    assert surf.shape == (F, 2 * n) and np.max(np.abs(surf - osurf)) <= tol * osurf.max()
    part = np.sort(osurf, axis=1)
    clear = (part[:, -1] - (part[:, -2] if 2 * n > 1 else 0)) > 10 * tol * osurf.max()
    assert np.array_equal(ridx[clear], oidx[clear])
    assert np.array_equal(rval, surf[np.arange(F), ridx.astype(np.int64)])      # the row record points at its own maximum
    assert np.array_equal(ridx, np.argmax(surf, axis=1).astype(np.uint64))      # ... the FIRST one (np.argmax: first max)
    # batch of 5 through the device API, shard [3, 30)
    B = 5
    nd = np.stack([x * (1 + 0.1 * b) for b in range(B)]).astype(cdt)
    hs = np.stack([np.roll(y, b) for b in range(B)]).astype(cdt)
    plan = eng.plan(n, fr, fs, dtype=dtype, row_begin=3, row_end=30)
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.full((B, 27, 2 * n), -1.0, dtype=tdt, device="cuda")
    di = torch.zeros((B, 27), dtype=torch.int64, device="cuda")
    dv = torch.zeros((B, 27), dtype=tdt, device="cuda")
    dp = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # the tensors above were filled on torch's stream; the plan launches on the engine's own
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    for b in range(B):
        ob, oi, ov = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr[3:30], fs)
        got = ds[b].cpu().numpy()
        assert np.max(np.abs(got - ob)) <= tol * max(ob.max(), 1e-300)
        pk = dp[b].cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0]
        best = int(np.argmax(got.max(axis=1)))
        assert int(pk["row"]) == 3 + best and int(pk["idx"]) == int(np.argmax(got[best]))
    plan.close()


# ------------------------------------------------------ seeded fuzz against the ORACLE --
def _fuzz_cases():
    rng = np.random.default_rng(20261004)
    sizes = [1, 2, 8, 32, 128, 512, 1024, 2048, 4096, 4096, 4096, 8192, 16384]
    cases = []
    for i in range(40):
        n = int(sizes[i % len(sizes)])
        dtype = "c128" if (i // 2) % 2 == 0 else "c64"
        budget = 1 << 21                                     # rows * L per case: keeps the numpy oracle fast
        nfreq = int(min(rng.integers(1, 701), max(1, budget // (2 * n))))
        fs = int(rng.choice([8000, 44100, 48000, 250000, 1000000]))
        batch = int(rng.integers(1, 6)) if nfreq * 2 * n * 5 <= budget * 2 else 1
        lo = int(rng.integers(0, nfreq))
        hi = int(rng.integers(lo + 1, nfreq + 1))
        if i % 3 == 0:
            lo, hi = 0, nfreq
        cases.append((i, n, dtype, nfreq, fs, batch, lo, hi))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c[0]}-n{c[1]}-{c[2]}-F{c[3]}-b{c[5]}-{c[6]}:{c[7]}")
def test_fuzz_every_path_vs_oracle(case, eng, oracle):
    """40 seeded random cases over every kernel path (small, chain, tuned n = 4096), both dtypes, nfreq 1...700,
    five sample rates, batches of 1...5 surfaces, random row shards -- each checked against the numpy ORACLE
    (np_caf_surface / np_find_peak), never against the HIP path itself: surface within 1e-6 (complex128) / 1e-3
    (complex64) of its maximum, row argmax equal wherever the oracle's best and second-best lag differ by more than
    the error bar, shard peak (row, lag) exact when the oracle's winning row leads by more than the error bar."""
    import torch
    import caf_cookoff_amd as caf
    i, n, dtype, nfreq, fs, batch, lo, hi = case
    rng = np.random.default_rng(555 + i)
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.sort(rng.uniform(-0.01 * fs, 0.01 * fs, nfreq))
    if i % 4 == 1:
        rng.shuffle(fr)                                      # any order, the list is the row order (mod.rs:135)
    nd = np.empty((batch, n), dtype=cdt)
    hs = np.empty((batch, n), dtype=cdt)
    for b in range(batch):
        f_true = float(fr[int(rng.integers(lo, hi))])
        lag = int(rng.integers(0, max(1, n // 4)))
        x, y = _planted(rng, n, fs, f_true, lag, cdt)
        nd[b], hs[b] = x, y
    plan = eng.plan(n, fr, fs, dtype=dtype, row_begin=lo, row_end=hi)
    rows = hi - lo
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.empty((batch, rows, 2 * n), dtype=tdt, device="cuda")
    di = torch.zeros((batch, rows), dtype=torch.int64, device="cuda")
    dv = torch.zeros((batch, rows), dtype=tdt, device="cuda")
    dp = torch.zeros((batch, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # the tensors above were filled on torch's stream; the plan launches on the engine's own
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), batch, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    for b in range(batch):
        osurf, oidx, oval = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr[lo:hi], fs)
        mx = osurf.max()
        got, gi, gv = ds[b].cpu().numpy(), di[b].cpu().numpy(), dv[b].cpu().numpy()
        assert np.max(np.abs(got - osurf)) <= tol * mx, f"case {i} surface {b}"
        if 2 * n > 1:
            part = np.partition(osurf, -2, axis=1)
            clear = (part[:, -1] - part[:, -2]) > 4 * tol * mx
            assert np.array_equal(gi[clear], oidx[clear].astype(np.int64)), f"case {i} surface {b}: row argmax"
        assert np.max(np.abs(gv.astype(np.float64) - oval)) <= tol * mx
        pk = dp[b].cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0]
        order = np.sort(oval)
        if len(order) == 1 or order[-1] - order[-2] > 4 * tol * mx:
            of, oi = oracle.np_find_peak(fr[lo:hi], oidx, oval)
            assert (pk["freq"], int(pk["idx"])) == (of, oi), f"case {i} surface {b}: shard peak"
            assert int(pk["row"]) == lo + int(np.argmax(oval))
    plan.close()


# ------------------------------------------------------ configs[3] with 32 points per thread (measured and rejected) --
def test_r32_variant_matches_oracle_and_product_kernel(eng, oracle, monkeypatch):
    """kernels_r32.hpp (16384 = 32 x 32 x 16, 512 threads, two LDS exchanges per transform) lives
    in the MEASUREMENT library (CAF_R32=1): parity-green against the oracle, row argmax and peak equal to the product
    chain kernel's on a multi-row launch that wraps the persistent grid (300 rows > 256 workgroups), and
    bit-identical from run to run.  It runs at the product kernel's speed, not faster -- DESIGN.md section 5."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    monkeypatch.setenv("CAF_R32", "1")
    meng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    n = 32768
    s0, s1, lag, fo = make_pair(n=n, seed=5, lag=777, foffset=-31.5, dtype=np.complex64)
    fr = np.concatenate([np.array([-40.0, -32.0, -31.5, -31.0, 0.0, 31.5, 977.25]), np.linspace(-60.0, 60.0, 293)])
    plan = meng.plan(n, fr, FS, dtype="c64")
    assert plan.kernel_name == "caf::k_r32_rows<float>"
    dn, dh = torch.from_numpy(s0[None]).cuda(), torch.from_numpy(s1[None]).cuda()
    outs = []
    for rep in range(2):
        ds = torch.empty((1, len(fr), 2 * n), dtype=torch.float32, device="cuda")
        di = torch.zeros((1, len(fr)), dtype=torch.int64, device="cuda")
        dv = torch.zeros((1, len(fr)), dtype=torch.float32, device="cuda")
        dp = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()  # the tensors above were filled on torch's stream; the plan launches on the engine's own
        plan.surface_dev(dn.data_ptr(), dh.data_ptr(), 1, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
        meng.synchronize()
        outs.append((ds[0].cpu().numpy(), di[0].cpu().numpy(), dv[0].cpu().numpy(), dp.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]))
    plan.close()
    meng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    surf, ridx, rval, pk = outs[0]
    rows = [0, 1, 2, 3, 6, 150, 255, 256, 257, 299]           # rows on both sides of the grid wrap
    osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr[rows], FS)
    mx = max(osurf.max(), float(surf.max()))
    assert np.max(np.abs(surf[rows] - osurf)) <= TOL32 * mx
    _, ridx0, rval0, pk0 = eng.surface_arrays(s0, s1, fr, FS, want_surface=False, dtype="c64")
    part = np.partition(surf.astype(np.float64), -2, axis=1)
    clear = (part[:, -1] - part[:, -2]) > 1e-4 * mx
    assert np.array_equal(ridx[clear], ridx0[clear].astype(np.int64)) and (pk["freq"], int(pk["idx"])) == (pk0.freq, pk0.idx) == (-31.5, lag)


def test_stream_run_stats_account_for_the_host_thread(eng):
    """caf_stream_run_stats: where the host thread spent the last caf_stream_run (fill / launch / wait / collect);
    the four parts are non-negative and add up to no more than the call's wall time."""
    import time
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    nd, hs, lags, _ = make_batch(32, 4096, FS, seed0=4100)
    plan = eng.plan(4096, caf.bench_shifts(), FS)
    st = caf.Stream(plan, batch=1, nslots=3, want_surface=False)
    st.run(nd[:4], hs[:4])
    t0 = time.perf_counter()
    peaks, _, _ = st.run(nd, hs)
    wall = time.perf_counter() - t0
    s = st.run_stats()
    assert all(v >= 0.0 for v in s.values()) and 0.0 < sum(s.values()) <= wall
    assert [int(p["idx"]) for p in peaks] == list(lags)
    st.close()
    plan.close()


def test_short_host_api_soak():
    """tools/host_api_soak.py, short form: 4000 polled host-pointer calls per dtype, every result bit-equal to round 0."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("host_api_soak", ROOT / "tools" / "host_api_soak.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(4000, "c128", log=False) == 0
    assert mod.run(4000, "c64", log=False) == 0


@pytest.mark.parametrize("n,dtype,split", [(64, "c128", False), (512, "c64", True), (8, "c128", True)])
def test_small_path_streaming_slots(n, dtype, split, eng, oracle):
    """caf_stream_* over a lane-group plan (n <= 512): every slot owns its haystack spectra, so slots (and the
    branches of a split replay) run concurrently; 23 pairs (ragged against batch 4) against the oracle."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.linspace(-200.0, 200.0, 21)
    nd = np.empty((23, n), dtype=cdt)
    hs = np.empty((23, n), dtype=cdt)
    for k in range(23):
        nd[k], hs[k] = _planted(rng, n, FS, float(fr[k % 21]), k % max(1, n // 2), cdt)
    plan = eng.plan(n, fr, FS, dtype=dtype)
    assert plan.path == "small"
    st = caf.Stream(plan, batch=4, nslots=3, want_surface=False, split=split)
    peaks, ridx, rval = st.run(nd, hs, want_rows=True)
    st.close()
    plan.close()
    for k in range(23):
        _, oidx, oval = oracle.np_caf_surface(nd[k].astype(np.complex128), hs[k].astype(np.complex128), fr, FS, want_surface=False)
        assert np.max(np.abs(rval[k].astype(np.float64) - oval)) <= tol * oval.max()
        order = np.sort(oval)
        if order[-1] - order[-2] > 4 * tol * oval.max():
            assert int(peaks[k]["row"]) == int(np.argmax(oval))


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_xcor_every_size_vs_oracle(n, dtype, eng, oracle):
    """Xcor::run (xcor_rustfft.rs:51-78) for every power of two: one launch up to n = 16384 (kernels_xcor.hpp; complex128:
    8192), radix-2 passes beyond; against the numpy restatement, plus the defining property out[k] = sum a[m+k] conj(b[m])
    on a shifted copy (peak at the shift) and Xcor's length assertions."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(7000 + n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    got = caf.Xcor(n, eng).run(a, b)
    want = oracle.np_xcor(a.astype(np.complex128), b.astype(np.complex128))
    assert got.dtype == cdt and got.shape == (n,)
    tol = (1e-10 if dtype == "c128" else 2e-4) * max(1.0, np.max(np.abs(want)))
    assert np.max(np.abs(got - want)) <= tol
    if n >= 4:
        sh = n // 3
        got2 = eng.xcor(np.roll(b, sh), b)      # a[m] = b[m - sh]  ->  peak at k = sh
        assert int(np.argmax(np.abs(got2))) == sh
    # a second call with other data through the same cached tables
    got3 = eng.xcor(b, a)
    assert np.max(np.abs(got3 - oracle.np_xcor(b.astype(np.complex128), a.astype(np.complex128)))) <= tol


def test_views_in_place_on_pinned_memory(eng):
    """caf_surface_view reads / writes memory of caf_host_alloc in place (no staging copy) and returns the same bits as
    through ordinary memory; a source that straddles the edge of a registered range still works (copy cut at the edge)."""
    rng = np.random.default_rng(3)
    surf = rng.random((12, 256))
    for view in ("go", "python"):
        ref = eng.surface_view(surf, view)
        psrc = eng.host_empty(surf.shape, np.float64)
        psrc[:] = surf
        assert np.array_equal(eng.surface_view(psrc, view), ref)
        big = _mmap_array((14, 256), np.float64)
        big[1:13] = surf
        eng.host_register(big[:6])                      # the first six rows only: big[1:13] straddles the edge
        assert np.array_equal(eng.surface_view(big[1:13], view), ref)
        eng.host_unregister(big[:6])
        del psrc


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 64, 128, 1024, 2048, 8192])
def test_generic_radix16_passes_vs_oracle(n, dtype, oracle, monkeypatch):
    """The path of every shape no LDS-resident kernel covers (n > 131072 / 65536 in the product): mixed-radix Stockham
    passes over HBM -- radix 16 while at least 16 points remain, then 8 / 4 / 2 (k_fft_pass).  Forced here for small
    and medium sizes through the measurement library (CAF_SMALL=0, CAF_CHAIN=0) so that every pass combination
    (L = 2 ... 16384: remainders 2, 4, 8 and none) is checked against the oracle."""
    import caf_cookoff_amd as caf
    monkeypatch.setenv("CAF_SMALL", "0")
    monkeypatch.setenv("CAF_CHAIN", "0")
    meng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    rng = np.random.default_rng(900 + n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.array([-50.0, 0.0, 12.5, 50.0, 333.0])
    lag = 0 if n < 4 else n // 4
    x, y = _planted(rng, n, FS, 12.5, lag, cdt)
    plan = meng.plan(n, fr, FS, dtype=dtype)
    assert plan.path == "generic" and "k_fft_pass" in plan.kernel_name
    plan.close()
    surf, ridx, rval, pk = meng.surface_arrays(x, y, fr, FS, dtype=dtype)
    osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
    assert np.max(np.abs(surf - osurf)) <= tol * osurf.max()
    if n >= 16:
        assert (pk.freq, int(pk.idx)) == oracle.np_find_peak(fr, oidx, oval) == (12.5, lag)
    meng.close()
