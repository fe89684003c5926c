"""caf_stream_*: back-to-back surfaces from host memory through pinned slots and captured hipGraphs (BASELINE configs[4]).
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
from pathlib import Path

import numpy as np
import pytest

from conftest import DATA
from gpu_common import FS, TOL32, TOL64, _pair, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


# ------------------------------------------------------------------ streaming --
def test_streaming_double_buffer(eng, oracle, golden, manifest):
    """BASELINE configs[4] mechanics: pinned double-buffered H2D + one hipGraph per slot;
    the ten reference pairs cycled through two slots give the bench-grid answers, and a
    replay of the same slot with new data gives new results (graphs are not stale)."""
    import caf_cookoff_amd as caf
    fr = oracle.bench_shifts()
    plan = eng.plan(4096, fr, FS)
    st = caf.Stream(plan, batch=2, nslots=2, want_surface=True)
    try:
        pairs = [_pair(oracle, k) for k in range(10)]
        expect = []
        for nd, hs in pairs:
            _, oi, ov = oracle.np_caf_surface(nd, hs, fr, FS, want_surface=False)
            expect.append(oracle.np_find_peak(fr, oi, ov) + (oi, ov))
        got = [None] * 10
        order = [(0, (0, 1)), (1, (2, 3)), (0, (4, 5)), (1, (6, 7)), (0, (8, 9))]
        pending = []
        for slot, ks in order:
            if len(pending) == 2:                     # slot is busy: retire its previous submit first
                ps, pks = pending.pop(0)
                peaks, ridx, rval = st.wait(ps)
                for j, k in enumerate(pks):
                    got[k] = (float(peaks[j]["freq"]), int(peaks[j]["idx"]), ridx[j].copy(), rval[j].copy())
            a, b = st.buffers(slot)
            for j, k in enumerate(ks):
                a[j], b[j] = pairs[k]
            st.submit(slot)
            pending.append((slot, ks))
        for ps, pks in pending:
            peaks, ridx, rval = st.wait(ps)
            for j, k in enumerate(pks):
                got[k] = (float(peaks[j]["freq"]), int(peaks[j]["idx"]), ridx[j].copy(), rval[j].copy())
        for k in range(10):
            ef, ei, oi, ov = expect[k]
            assert (got[k][0], got[k][1]) == (ef, ei), f"chirp_{k}"
            assert np.array_equal(got[k][2], oi) and np.max(np.abs(got[k][3] - ov)) <= TOL64 * ov.max()
        assert (got[0][0], got[0][1]) == (manifest["bench"]["0"]["best_freq"], manifest["bench"]["0"]["best_idx"])
        assert st.surface_ptr(0) != 0
    finally:
        st.close()
        plan.close()


def test_short_stream_soak():
    """tools/stream_sweep.py soak, 2048 surfaces x 6 rounds x 5 streaming forms x 2 dtypes: every row peak and record of
    every round equals round 0's bit for bit.  (What it caught in round 2: row words written to the pinned
    result buffers with plain stores reached the host after the sequence word of the launch, 1-3 surfaces in
    10^5; they are system-scope stores now.)"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "stream_sweep.py"), "soak", "2048", "6"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "SOAK ok" in r.stdout, r.stdout + r.stderr


def test_chain_streaming_slots(eng, oracle):
    """Streaming through a chain plan: every slot owns its spectrum (and radix-4 scratch) buffers, so
    two slots in flight do not disturb each other (n = 32768 complex64, R = 4)."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    n = 32768
    fr = np.array([-5.0, 0.0, 5.0, 10.0])
    pairs = [make_pair(n=n, seed=500 + k, lag=20 + 11 * k, foffset=[5.0, -5.0, 10.0, 0.0][k], dtype=np.complex64) for k in range(4)]
    plan = eng.plan(n, fr, FS, dtype="c64")
    st = caf.Stream(plan, batch=1, nslots=2, want_surface=True)
    try:
        got = {}
        for rnd in range(2):
            for slot in range(2):
                k = 2 * rnd + slot
                a, b = st.buffers(slot)
                a[0], b[0] = pairs[k][0], pairs[k][1]
                st.submit(slot)
            for slot in range(2):
                peaks, _, _ = st.wait(slot, want_rows=False)
                got[2 * rnd + slot] = (float(peaks[0]["freq"]), int(peaks[0]["idx"]))
        for k in range(4):
            assert got[k] == (pairs[k][3], pairs[k][2]), f"pair {k}"
    finally:
        st.close()
        plan.close()


@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_streaming_split_mode_parity(dtype, eng, oracle):
    """CAF_STREAM_SPLIT: four independent single-surface node chains per graph replay (parallel
    branches, each with its own stage-in, spectrum buffer, row kernel and find_peak).  The ten
    reference pairs cycled through two such slots give the bench-grid answers of the oracle; the row
    peaks agree with the batched mode's (indices exactly, values to rounding)."""
    import caf_cookoff_amd as caf
    fr = oracle.bench_shifts()
    pairs = [oracle.load_pair(DATA, f"chirp_{k}_raw.c64", oracle.KATS[k][1]) for k in range(10)]
    expect = []
    for nd, hs in pairs:
        _, oi, ov = oracle.np_caf_surface(nd, hs, fr, FS, want_surface=False)
        expect.append(oracle.np_find_peak(fr, oi, ov) + (oi, ov))
    plan = eng.plan(4096, fr, FS, dtype=dtype)
    results = {}
    for split in (True, False):
        st = caf.Stream(plan, batch=4, nslots=2, want_surface=False, split=split)
        got = {}
        order = [(0, (0, 1, 2, 3)), (1, (4, 5, 6, 7)), (0, (8, 9, 0, 1)), (1, (2, 3, 4, 5))]
        pending = []
        for slot, ks in order:
            if len(pending) == 2:
                ps, pks = pending.pop(0)
                peaks, ridx, rval = st.wait(ps)
                for j, k in enumerate(pks):
                    got[k] = (float(peaks[j]["freq"]), int(peaks[j]["idx"]), ridx[j].copy(), rval[j].copy())
            a, b = st.buffers(slot)
            for j, k in enumerate(ks):
                a[j], b[j] = pairs[k]
            st.submit(slot)
            pending.append((slot, ks))
        for ps, pks in pending:
            peaks, ridx, rval = st.wait(ps)
            for j, k in enumerate(pks):
                got[k] = (float(peaks[j]["freq"]), int(peaks[j]["idx"]), ridx[j].copy(), rval[j].copy())
        st.close()
        results[split] = got
    plan.close()
    tol = TOL64 if dtype == "c128" else TOL32
    for k in range(10):
        ef, ei, oi, ov = expect[k]
        f, i, ri, rv = results[True][k]
        assert i == ei, f"chirp_{k}"
        if dtype == "c128":
            assert f == ef and np.array_equal(ri, oi)
        assert np.max(np.abs(rv.astype(np.float64) - ov)) <= tol * ov.max()
        # split chains of n = 4096 plans are the one-launch surface kernel: same functions as the batched
        # row kernel but a separate instantiation (the compiler may contract a*b+c differently), so the
        # values agree to rounding, the indices exactly
        fb, ib, rib, rvb = results[False][k]
        assert (f, i) == (fb, ib) and np.array_equal(ri, rib)
        assert np.max(np.abs(rv.astype(np.float64) - rvb.astype(np.float64))) <= (1e-13 if dtype == "c128" else 1e-5) * ov.max()


@pytest.mark.parametrize("dtype,nrows,form", [("c128", 400, "one"), ("c64", 400, "one"), ("c128", 1, "one"), ("c128", 1300, "one"),
                                              ("c64", 37, "one"), ("c128", 400, "two"), ("c64", 400, "two"), ("c128", 1300, "two")])
def test_stream_single_launch_surface(dtype, nrows, form, eng, oracle):
    """Single-surface streaming chains of the n = 4096 path are ONE launch (k_seq_surface: needle staging,
    haystack spectrum, rows and find_peak as ordered-ticket roles of one grid).  Against the three-node form
    {k_seq_prepare, row kernel, k_peak} and the oracle: complex128 argmax indices and caf_peak records
    equal the three-node form's exactly and the values to 1e-13 of the peak (same functions, separate
    instantiation: the compiler contracts a*b+c differently in places); complex64 runs k_seq_rows'
    arithmetic instead of k_duo_rows' and is held to the oracle tolerance.  Twelve replays over two slots check that the launch
    re-arms its own counters; 1300 rows exceed the resident workgroup slots (later tickets start as earlier
    ones retire), 1 row and 37 rows are the small ends.  form "two": the same kernel behind a k_seq_prepare
    node (staging + spectrum), i.e. rows + find_peak only -- what slots with more than two surfaces in
    flight use."""
    import torch
    import caf_cookoff_amd as caf
    fr = np.linspace(-100.0, 100.0, nrows, endpoint=False) if nrows > 1 else np.array([12.5])
    pairs = [oracle.load_pair(DATA, f"chirp_{k}_raw.c64", oracle.KATS[k][1]) for k in range(6)]
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    plan = eng.plan(4096, fr, FS, dtype=dtype)
    res = {}
    for three in (False, True):
        st = caf.Stream(plan, batch=1, nslots=2, want_surface=True, three_kernels=three, one_kernel=form == "one" and not three,
                        two_kernels=form == "two" and not three)
        out = []
        pending = []
        for step in range(12):
            slot = step % 2
            if len(pending) == 2:
                ps, pk = pending.pop(0)
                peaks, ridx, rval = st.wait(ps)
                out.append((pk, peaks[0].copy(), ridx[0].copy(), rval[0].copy()))
            a, b = st.buffers(slot)
            a[0], b[0] = (x.astype(cdt) for x in pairs[step % 6])
            st.submit(slot)
            pending.append((slot, step % 6))
        for ps, pk in pending:
            peaks, ridx, rval = st.wait(ps)
            out.append((pk, peaks[0].copy(), ridx[0].copy(), rval[0].copy()))
        # the surface of the last replay of slot 1 (pair 5), read through a borrowed device view

        class _Dev:
            __cuda_array_interface__ = {"shape": (nrows, 8192), "typestr": "<f8" if dtype == "c128" else "<f4",
                                        "data": (st.surface_ptr(1), False), "version": 2}
        res[three] = (out, torch.as_tensor(_Dev(), device="cuda").cpu().numpy().astype(np.float64))
        st.close()
    plan.close()
    one, ref = res[False], res[True]
    assert len(one[0]) == 12
    for (k, pk, ri, rv), (k2, pk2, ri2, rv2) in zip(one[0], ref[0]):
        assert k == k2
        nd, hs = pairs[k]
        _, oi, ov = oracle.np_caf_surface(nd, hs, fr, FS, want_surface=False)
        ef, ei = oracle.np_find_peak(fr, oi, ov)
        assert int(pk["idx"]) == ei and (dtype == "c64" or float(pk["freq"]) == ef)
        assert np.max(np.abs(rv.astype(np.float64) - ov)) <= tol * ov.max()
        if dtype == "c128":
            assert np.array_equal(ri, oi) and np.array_equal(ri, ri2)
            assert (pk["freq"], pk["idx"], pk["row"]) == (pk2["freq"], pk2["idx"], pk2["row"])
            assert np.max(np.abs(rv - rv2)) <= 1e-13 * ov.max()
    osurf, _, _ = oracle.np_caf_surface(*pairs[5], fr, FS)
    assert np.max(np.abs(one[1] - osurf)) <= tol * osurf.max()
    if dtype == "c128":
        print(f"one-launch vs three-kernel surface: max|d|/max = {np.max(np.abs(one[1] - ref[1])) / osurf.max():.2e}")
        assert np.max(np.abs(one[1] - ref[1])) <= 1e-13 * osurf.max()


@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_stream_single_launch_edge_inputs(dtype, eng):
    """The one-launch surface kernel on the inputs the reference's comparison rule is sensitive to (mod.rs:
    32-35, 143-151): all-zero pair -> every row (0, 0.0), peak (0.0, 0) with row -1; NaN in the needle / in
    the haystack -> the same; an exact two-row tie -> the first row wins; then an ordinary pair on the same
    slots (the counters re-arm after every one of these).  Each answer equals the batched API's."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(11)
    n = 4096
    cdt = np.complex128 if dtype == "c128" else np.complex64
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    b = np.roll(a, 21)
    z = np.zeros(n, dtype=cdt)
    an, bn = a.copy(), b.copy()
    an[n // 3] = complex(np.nan, 1.0)
    bn[7] = complex(1.0, np.nan)
    fr = np.array([-2.5, 0.0, 0.0, 4.0])  # rows 1 and 2 tie exactly
    cases = [(z, z), (an, b), (a, bn), (a, b), (b, a)]
    plan = eng.plan(n, fr, FS, dtype=dtype)
    st = caf.Stream(plan, batch=1, nslots=2, want_surface=False)
    try:
        nd = np.stack([c[0] for c in cases])
        hs = np.stack([c[1] for c in cases])
        peaks, ridx, rval = st.run(nd, hs, want_rows=True)
    finally:
        st.close()
        plan.close()
    for k, (x, y) in enumerate(cases):
        _, ri, rv, pk = eng.surface_arrays(x, y, fr, FS, dtype=dtype, want_surface=False)
        assert (float(peaks[k]["freq"]), int(peaks[k]["idx"]), int(peaks[k]["row"])) == (pk.freq, pk.idx, pk.row), k
        assert np.array_equal(ridx[k], ri), k
        assert np.allclose(rval[k], rv, rtol=1e-12 if dtype == "c128" else 1e-4, atol=0.0), k
    for k in (0, 1, 2):
        assert (float(peaks[k]["freq"]), int(peaks[k]["idx"]), float(peaks[k]["val"]), int(peaks[k]["row"])) == (0.0, 0, 0.0, -1)
        assert not ridx[k].any() and not rval[k].any()
    assert int(peaks[3]["row"]) == 1 and int(peaks[3]["idx"]) == 21 and float(peaks[3]["freq"]) == 0.0


@pytest.mark.parametrize("dtype,batch,nslots,split", [("c128", 1, 2, False), ("c128", 1, 3, False), ("c128", 4, 2, True),
                                                      ("c128", 4, 2, False), ("c64", 1, 2, False), ("c64", 3, 2, True)])
def test_stream_run_native_loop(dtype, batch, nslots, split, eng, oracle):
    """caf_stream_run: the whole streaming loop in one native call.  23 pairs (the ten reference pairs,
    cycled; 23 is ragged for every batch used here) come back in input order with the oracle's (tau, f) and
    row peaks; two consecutive runs on the same stream agree bit for bit (slots are re-armed by the launches
    themselves, completion is read from the pinned sequence words)."""
    import caf_cookoff_amd as caf
    fr = oracle.bench_shifts()
    pairs = [oracle.load_pair(DATA, f"chirp_{k}_raw.c64", oracle.KATS[k][1]) for k in range(10)]
    expect = []
    for nd, hs in pairs:
        _, oi, ov = oracle.np_caf_surface(nd, hs, fr, FS, want_surface=False)
        expect.append(oracle.np_find_peak(fr, oi, ov) + (oi, ov))
    count = 23
    nd = np.stack([pairs[k % 10][0] for k in range(count)])
    hs = np.stack([pairs[k % 10][1] for k in range(count)])
    plan = eng.plan(4096, fr, FS, dtype=dtype)
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=False, split=split)
    tol = TOL64 if dtype == "c128" else TOL32
    try:
        peaks, ridx, rval = st.run(nd, hs, want_rows=True)
        peaks2, ridx2, rval2 = st.run(nd, hs, want_rows=True)
        assert np.array_equal(peaks, peaks2) and np.array_equal(ridx, ridx2) and np.array_equal(rval, rval2)
        for k in range(count):
            ef, ei, oi, ov = expect[k % 10]
            assert int(peaks[k]["idx"]) == ei, f"pair {k}"
            if dtype == "c128":
                assert float(peaks[k]["freq"]) == ef and np.array_equal(ridx[k], oi)
            assert np.max(np.abs(rval[k].astype(np.float64) - ov)) <= tol * ov.max()
        # an empty run is a no-op; the step-by-step API still works on the same stream afterwards
        p0, _, _ = st.run(nd[:0], hs[:0])
        assert len(p0) == 0
        a, b = st.buffers(0)
        a[:], b[:] = 0, 0
        a[0], b[0] = pairs[3]
        st.submit(0)
        pk, _, _ = st.wait(0, want_rows=False)
        assert int(pk[0]["idx"]) == expect[3][1]
    finally:
        st.close()
        plan.close()


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



@pytest.mark.parametrize("dtype,n,batch,nslots", [("c128", 4096, 8, 4), ("c128", 4096, 1, 2), ("c64", 4096, 3, 2), ("c64", 2048, 2, 2),
                                                  ("c128", 64, 5, 3), ("c128", 4096, 20, 2), ("c64", 4096, 32, 2), ("c128", 4096, 16, 3)])
def test_stream_memcpy_nodes_form_equals_mapped_form(dtype, n, batch, nslots, eng, oracle):
    """CAF_STREAM_MEMCPY_NODES (BASELINE configs[4] to the letter: hipMemcpyAsync nodes carry the inputs in and the results out)
    returns the same bits as the default form (kernels read / write the mapped pinned buffers) for the same batched chain, and
    the planted peaks; a ragged last replay included."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    cdt = np.complex128 if dtype == "c128" else np.complex64
    fr = np.linspace(-100.0, 100.0, 80, endpoint=False)   # (make_batch plants its offsets within +-100 Hz)
    count = 3 * batch * nslots + 1
    nd, hs, lags, _ = make_batch(count, n, FS, seed0=9100, dtype=cdt)
    plan = eng.plan(n, fr, FS, dtype=dtype)
    res = {}
    for mc in (False, True):
        st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, memcpy_nodes=mc, two_kernels=(batch == 1 and not mc))
        res[mc] = st.run(nd, hs, want_rows=True)
        st.close()
    (p0, i0, v0), (p1, i1, v1) = res[False], res[True]
    assert np.array_equal(p1["idx"], np.asarray(lags)) and np.array_equal(i0, i1)
    if batch > 1:   # the same batched kernels either way: bit-equal; batch 1 of n = 4096 compares the one-launch surface with the batched kernels
        assert p0.tobytes() == p1.tobytes() and np.array_equal(v0, v1)
    else:
        assert np.array_equal(p0["idx"], p1["idx"]) and np.array_equal(p0["row"], p1["row"]) and np.allclose(v0, v1, rtol=1e-12, atol=0)
    if batch >= 16:
        # replays of 16 surfaces or more launch their rows on 32 workgroup slots fewer (api/stream.inc): the same bits as the
        # device-pointer batch of the same pairs on the full grid, row records and peaks
        import torch
        tdt = torch.float64 if dtype == "c128" else torch.float32
        dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
        di = torch.empty((count, len(fr)), dtype=torch.int64, device="cuda")
        dv = torch.empty((count, len(fr)), dtype=tdt, device="cuda")
        dp = torch.empty((count, 4), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        plan.surface_dev(dn.data_ptr(), dh.data_ptr(), count, None, di.data_ptr(), dv.data_ptr(), dp.data_ptr())
        eng.synchronize()
        assert np.array_equal(i0.astype(np.int64), di.cpu().numpy()) and np.array_equal(v0, dv.cpu().numpy())
        assert p0.tobytes() == dp.cpu().numpy().tobytes()
    if (dtype, n, batch) == ("c128", 4096, 1):
        # the flag forces the {copy, spectrum, rows, find_peak, copy} chain: asking for a single-launch form as well is refused,
        # not silently ignored; the older three-node chain and split chains go with it
        from caf_cookoff_amd import _lib
        for kw in (dict(one_kernel=True), dict(two_kernels=True)):
            with pytest.raises(caf.CafError) as ei:
                caf.Stream(plan, batch=1, nslots=2, memcpy_nodes=True, **kw)
            assert ei.value.code == _lib.CAF_ERR_BAD_ARG and "CAF_STREAM_MEMCPY_NODES excludes" in str(ei.value)
        caf.Stream(plan, batch=4, nslots=2, memcpy_nodes=True, split=True).close()
    plan.close()
