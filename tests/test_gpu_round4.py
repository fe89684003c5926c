"""GPU parity tests added in round 4: Doppler-row shards of one surface behind the C ABI (caf_multi_surface_*: host
join and in-process RCCL join), surface-parallel streams that keep their surfaces, red zones around every buffer a
kernel writes, a seeded fuzz of the long chain kernels (R = 4 / 8 / 16) against the ORACLE and an oracle comparison at
bench.py's own launch shape.  Every call goes through the C ABI (libcaf_hip.so)."""
import numpy as np
import pytest

from conftest import DATA

pytestmark = pytest.mark.gpu

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


def _planted(rng, n, fs, f, lag, cdt=np.complex128):
    """haystack = needle delayed by `lag` (lag < 0: advanced, the peak lands at 2n + lag) and shifted by f."""
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n) if n >= 8 else \
        (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    y = np.roll(x, lag) * np.exp(2j * np.pi * f * np.arange(n) / fs)
    if lag >= 0:
        y[:lag] = 0
    else:
        y[lag:] = 0
    y = y + 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(cdt), y.astype(cdt)


def _dev_view(ptr, shape, typestr):
    import torch

    class _Dev:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Dev(), device="cuda")


# ------------------------------------------------------ row shards of ONE surface behind the C ABI --
@pytest.mark.parametrize("workers", [2, 3])
@pytest.mark.parametrize("kat", [0, 2, 4])
def test_multi_surface_kats_bit_equal_to_unsharded(kat, workers, eng, oracle):
    """caf_multi_surface_run over [0, 0] / [0, 0, 0] (contexts on one GPU) == the unsharded caf_surface_c128 bit for bit
    (surface, every row record, the peak) and exact on the reference's KAT 0 / 2 (tightest row margin) / 4."""
    import caf_cookoff_amd as caf
    _, hf, (s, e, st), exp = oracle.KATS[kat]
    nd, hs = oracle.load_pair(DATA, f"chirp_{kat}_raw.c64", hf)
    fr = oracle.gen_float_shifts(s, e, st)
    surf0, ridx0, rval0, pk0 = eng.surface_arrays(nd, hs, fr, FS)
    ms = caf.MultiSurface([0] * workers, len(nd), fr, FS)
    assert ms.ndev == workers
    los = [ms.worker_info(w)[1:3] for w in range(workers)]
    assert los == [caf.shard_range(len(fr), w, workers) for w in range(workers)]
    pinned = ms.host_empty((len(fr), 2 * len(nd)), np.float64)        # every worker writes its rows in place
    for out in (None, pinned):
        surf, ridx, rval, pk = ms.run(nd, hs, out=out)
        assert np.array_equal(surf, surf0) and np.array_equal(ridx, ridx0) and np.array_equal(rval, rval0)
        assert (pk["freq"], int(pk["idx"])) == tuple(exp) == (pk0.freq, pk0.idx)
        assert int(pk["row"]) == pk0.row and pk["val"] == pk0.val
    stats, shard = ms.run_stats()
    assert stats["shards_s"] > 0 and len(shard) == workers
    best = caf.multi_surface_reduce(shard)
    assert (best["freq"], int(best["idx"]), int(best["row"])) == (pk["freq"], int(pk["idx"]), int(pk["row"]))
    # peaks only
    _, ridx, rval, pk = ms.run(nd, hs, want_surface=False)
    assert np.array_equal(ridx, ridx0) and (pk["freq"], int(pk["idx"])) == tuple(exp)
    with pytest.raises(RuntimeError, match="still alive"):   # the object owns the pinned arena under `pinned`
        ms.close()
    del pinned, surf, out
    ms.close()


@pytest.mark.parametrize("dtype,n,nfreq", [("c64", 4096, 400), ("c128", 1024, 37), ("c64", 2048, 11), ("c128", 64, 23),
                                          ("c64", 32768, 9), ("c128", 16384, 5), ("c128", 4096, 2), ("c64", 512, 1)])
def test_multi_surface_every_path_vs_oracle_and_unsharded(dtype, n, nfreq, eng, oracle):
    """Every kernel family behind the sharded call (tuned n = 4096 incl. the one-launch form, chain R = 2 / 4, lane-group
    rows), both dtypes, ragged shards and more workers than rows (empty shards): equal to the unsharded call bit for bit,
    and within tolerance of the ORACLE with the oracle's global peak."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(n + nfreq)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.linspace(-90.0, 90.0, nfreq) if nfreq > 1 else np.array([-33.0])   # (a regular grid: no two rows closer than f32 resolves)
    x, y = _planted(rng, n, FS, float(fr[nfreq // 2]), min(37, n // 4), cdt)
    surf0, ridx0, rval0, pk0 = eng.surface_arrays(x, y, fr, FS, dtype=dtype)
    osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
    of, oi = oracle.np_find_peak(fr, oidx, oval)
    for workers in (2, 3):
        ms = caf.MultiSurface([0] * workers, n, fr, FS, dtype=dtype)
        surf, ridx, rval, pk = ms.run(x, y)
        assert np.array_equal(surf, surf0) and np.array_equal(ridx, ridx0) and np.array_equal(rval, rval0)
        assert (pk["freq"], int(pk["idx"]), int(pk["row"])) == (pk0.freq, pk0.idx, pk0.row) and (pk["freq"], int(pk["idx"])) == (of, oi)
        assert np.max(np.abs(surf - osurf)) <= tol * osurf.max()
        ms.close()


def test_multi_surface_tie_and_no_peak_semantics(eng):
    """find_peak over the joined rows (mod.rs:31-42): equal maxima in different shards -> the lowest global row; an
    all-zero surface -> (0.0, 0), row -1.  Through the kernels (identical rows give identical peaks) and through the
    reduction rule by itself."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(7)
    x, y = _planted(rng, 4096, FS, 25.0, 100)
    fr = np.array([25.0, -40.0, 10.0, 25.0, 3.0, 25.0])      # rows 0, 3, 5 are the same row: exactly equal peaks
    ms = caf.MultiSurface([0, 0, 0], 4096, fr, FS)            # shards [0,2) [2,4) [4,6): one winner candidate in each
    _, ridx, rval, pk = ms.run(x, y, want_surface=False)
    assert rval[0] == rval[3] == rval[5] and int(pk["row"]) == 0 and pk["freq"] == 25.0 and int(pk["idx"]) == 100
    _, shard = ms.run_stats()
    assert [int(s["row"]) for s in shard] == [0, 3, 5]
    z = np.zeros(4096, dtype=np.complex128)
    _, ridx, rval, pk = ms.run(z, z, want_surface=False)
    assert (pk["val"], pk["freq"], int(pk["idx"]), int(pk["row"])) == (0.0, 0.0, 0, -1) and not rval.any()
    ms.close()


def test_multi_surface_rccl_join_single_rank(eng, oracle):
    """CAF_MULTI_REDUCE_RCCL with ONE rank on this one-GPU box: librccl is dlopen()ed, ncclCommInitAll + the two grouped
    all-reduces run on the worker's stream, and the result equals the host join.  Repeated device ids are refused (one
    RCCL rank per GPU)."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    _, hf, (s, e, st), exp = oracle.KATS[2]
    nd, hs = oracle.load_pair(DATA, "chirp_2_raw.c64", hf)
    fr = oracle.gen_float_shifts(s, e, st)
    ms = caf.MultiSurface([0], len(nd), fr, FS, rccl=True)
    for _ in range(3):
        surf, ridx, rval, pk = ms.run(nd, hs)
        assert (pk["freq"], int(pk["idx"])) == tuple(exp)
        assert pk["val"] == rval[int(pk["row"])] == surf[int(pk["row"]), int(pk["idx"])]
    stats, _ = ms.run_stats()
    assert stats["reduce_s"] > 0
    z = np.zeros(len(nd), dtype=np.complex128)
    _, _, _, pk = ms.run(z, z, want_surface=False)
    assert (pk["val"], pk["freq"], int(pk["idx"]), int(pk["row"])) == (0.0, 0.0, 0, -1)
    ms.close()
    # configs[3]'s kernel family (chain, complex64) through the RCCL join
    rng = np.random.default_rng(3)
    f3 = np.arange(24) * 0.05 - 0.6
    x, y = _planted(rng, 32768, FS, float(f3[17]), 211, np.complex64)
    ms = caf.MultiSurface([0], 32768, f3, FS, dtype="c64", rccl=True)
    _, _, _, pk = ms.run(x, y, want_surface=False)
    assert int(pk["idx"]) == 211 and int(pk["row"]) == 17
    ms.close()
    with pytest.raises(caf.CafError) as ei:
        caf.MultiSurface([0, 0], 4096, fr, FS, rccl=True)
    assert ei.value.code == _lib.CAF_ERR_BAD_ARG and "distinct devices" in str(ei.value)


def test_multi_surface_errors(eng):
    import caf_cookoff_amd as caf
    from caf_cookoff_amd import _lib
    fr = np.array([0.0, 1.0, 2.0])
    with pytest.raises(caf.CafError) as ei:
        caf.MultiSurface([0, 999], 4096, fr, FS)
    assert ei.value.code == _lib.CAF_ERR_NO_DEVICE
    with pytest.raises(caf.CafError) as ei:
        caf.MultiSurface([0], 4095, fr, FS)
    assert ei.value.code == _lib.CAF_ERR_LENGTH
    with pytest.raises(caf.CafError):
        caf.MultiSurface([], 4096, fr, FS)
    ms = caf.MultiSurface([0, 0], 64, fr, FS)
    with pytest.raises(AssertionError):
        ms.run(np.zeros(64, dtype=np.complex128), np.zeros(32, dtype=np.complex128))
    # an empty freq list: no rows, (0.0, 0)
    ms0 = caf.MultiSurface([0, 0], 64, np.array([]), FS)
    surf, ridx, rval, pk = ms0.run(np.ones(64, dtype=np.complex128), np.ones(64, dtype=np.complex128))
    assert surf.shape == (0, 128) and len(ridx) == 0 and (pk["freq"], int(pk["idx"]), int(pk["row"])) == (0.0, 0, -1)
    ms0.close()
    ms.close()


# ------------------------------------------------------ surface-parallel streams keep their surfaces --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_multi_stream_surfaces_vs_oracle(dtype, eng, oracle):
    """caf_multi_stream_create(want_surface = 1): 37 pairs over two contexts on GPU 0; every pair's surface is found with
    caf_multi_stream_locate / caf_multi_stream_surface and compared with the ORACLE (three replays per worker on three
    slots: all resident); a run that wraps the slots reports the early pairs as no longer resident."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    fr = caf.bench_shifts()[::16]   # 25 rows
    nd, hs, lags, _ = make_batch(37, 4096, FS, seed0=4400, dtype=cdt)
    ms = caf.MultiStream([0, 0], 4096, fr, FS, dtype=dtype, nslots=3, want_surface=True)
    peaks, ridx, rval = ms.run(nd, hs, want_rows=True)
    ts = "<f8" if dtype == "c128" else "<f4"
    slabs = {}
    for k in range(37):
        w, slot, idx, resident = ms.locate(37, k)
        assert w == k % 2 and resident
        if (w, slot) not in slabs:
            assert ms.surface_ptr(w, slot) != 0
            slabs[(w, slot)] = _dev_view(ms.surface_ptr(w, slot), (8, len(fr), 8192), ts).cpu().numpy()
        got = slabs[(w, slot)][idx]
        osurf, oidx, oval = oracle.np_caf_surface(nd[k].astype(np.complex128), hs[k].astype(np.complex128), fr, FS)
        assert np.max(np.abs(got - osurf)) <= tol * osurf.max(), f"pair {k}"
        assert np.array_equal(got[np.arange(len(fr)), ridx[k].astype(np.int64)], rval[k])   # the row records index this surface
        assert int(peaks[k]["idx"]) == lags[k] == oracle.np_find_peak(fr, oidx, oval)[1]
    assert len(slabs) == 6
    # 80 pairs: 40 per worker = five replays on three slots -> the first two replays' surfaces are gone
    assert [ms.locate(80, k)[3] for k in (0, 1, 31, 32, 79)] == [False, False, False, True, True]
    ms.close()
    ms0 = caf.MultiStream([0, 0], 4096, fr, FS, dtype=dtype, nslots=2)
    assert ms0.surface_ptr(0, 0) == 0
    ms0.close()


# ------------------------------------------------------ red zones around everything a kernel writes --
class _Fenced:
    """A caller-owned device buffer with 4 KiB of 0xA5 either side (torch memory, pointer offset by one page)."""
    PAGE = 4096

    def __init__(self, shape, dtype):
        import torch
        self.shape, self.dtype = tuple(shape), dtype
        self.nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        self.raw = torch.full((self.nbytes + 2 * self.PAGE,), 0xA5, dtype=torch.uint8, device="cuda")

    @property
    def ptr(self):
        return self.raw.data_ptr() + self.PAGE

    def tensor(self):
        return self.raw[self.PAGE:self.PAGE + self.nbytes].view(self.dtype).reshape(self.shape)

    def intact(self):
        return bool((self.raw[:self.PAGE] == 0xA5).all()) and bool((self.raw[self.PAGE + self.nbytes:] == 0xA5).all())


@pytest.fixture
def guards():
    import caf_cookoff_amd as caf
    caf.debug_guard_bands(4096)
    yield caf
    caf.debug_guard_bands(0)


def _check_guards(caf, min_allocs=1):
    checked, bad = caf.debug_check_guards()
    assert bad == 0 and checked >= min_allocs
    return checked


_DEV_CASES = [
    # (n, dtype, nfreq, lo, hi, batch): every kernel family; ragged rows, odd shards, batches beyond the resident set
    (16, "c128", 5, 0, 5, 3), (16, "c64", 7, 2, 7, 70), (64, "c64", 33, 1, 30, 9), (256, "c128", 19, 0, 19, 41),
    (512, "c64", 3, 0, 3, 700), (8, "c128", 2, 1, 2, 1),
    (1024, "c128", 13, 3, 11, 5), (1024, "c64", 29, 0, 29, 40), (2048, "c64", 7, 0, 7, 3), (8192, "c128", 9, 2, 9, 2),
    (16384, "c64", 5, 0, 5, 2),
    (4096, "c128", 401, 0, 401, 6), (4096, "c64", 401, 7, 398, 7), (4096, "c128", 3, 0, 3, 1), (4096, "c64", 1, 0, 1, 5),
    (16384, "c128", 5, 1, 4, 2),                                   # R = 4 complex128
    (32768, "c64", 11, 2, 9, 3), (32768, "c64", 1300, 0, 1300, 1),    # R = 4 complex64 (configs[3]); more rows than resident workgroups
    (32768, "c128", 5, 0, 5, 1),                                   # R = 8 complex128
    (65536, "c64", 7, 1, 6, 2),                                    # R = 8 complex64
    (65536, "c128", 3, 0, 3, 1),                                   # R = 16 complex128
    (131072, "c64", 5, 0, 5, 1),                                   # R = 16 complex64
    (262144, "c64", 2, 0, 2, 1), (131072, "c128", 2, 0, 2, 1),     # generic radix-16 passes over HBM
]


@pytest.mark.parametrize("case", _DEV_CASES, ids=lambda c: f"n{c[0]}-{c[1]}-F{c[2]}-{c[3]}:{c[4]}-b{c[5]}")
def test_red_zones_device_api(case, guards):
    """caf_surface_dev with every buffer fenced: the caller-owned surface / row_idx / row_val / caf_peak arrays sit between
    two pages of 0xA5 (torch memory, offset pointers), and every allocation the library makes for this context (tables,
    phasors, spectra, slabs, ticket words) has red zones of its own (caf_debug_guard_bands).  After the launches every
    fence is intact, and the results are the unfenced ones (peak row and lag of the plant)."""
    import torch
    caf = guards
    n, dtype, nfreq, lo, hi, batch = case
    rng = np.random.default_rng(n * 7 + nfreq)
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    fr = np.linspace(-80.0, 80.0, nfreq) if nfreq > 1 else np.array([12.5])
    rows = hi - lo
    lag = min(5, n // 4)
    x, y = _planted(rng, n, FS, float(fr[lo + rows // 2]), lag, cdt)
    e = caf.Engine(0)
    try:
        plan = e.plan(n, fr, FS, dtype=dtype, row_begin=lo, row_end=hi)
        dn = torch.from_numpy(np.tile(x, (batch, 1))).cuda()
        dh = torch.from_numpy(np.tile(y, (batch, 1))).cuda()
        fs_, fi, fv, fp = (_Fenced((batch, rows, 2 * n), tdt), _Fenced((batch, rows), torch.int64), _Fenced((batch, rows), tdt),
                           _Fenced((batch, 4), torch.float64))
        torch.cuda.synchronize()
        for with_surface in (True, False):
            plan.surface_dev(dn.data_ptr(), dh.data_ptr(), batch, fs_.ptr if with_surface else None, fi.ptr, fv.ptr, fp.ptr)
            e.synchronize()
            assert all(f.intact() for f in (fs_, fi, fv, fp)), "a kernel wrote outside a caller-owned buffer"
            _check_guards(caf)
        pk = fp.tensor().cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
        gv, gi = fv.tensor().cpu().numpy(), fi.tensor().cpu().numpy()
        best = np.argmax(gv, axis=1)                                   # first maximum == first strictly-greater row
        assert np.array_equal(pk["row"], lo + best) and np.array_equal(pk["idx"].astype(np.int64), gi[np.arange(batch), best])
        if n >= 1024:                                                  # (shorter inputs cannot tell these rows apart)
            assert (pk["row"] == lo + rows // 2).all() and (pk["idx"] == lag).all()
        assert torch.equal(fv.tensor()[0], fs_.tensor()[0].max(dim=1).values)
        plan.close()
    finally:
        e.close()


_HOST_CASES = [(16, "c128", 5), (64, "c64", 33), (512, "c128", 7), (1024, "c64", 13), (2048, "c128", 3), (8192, "c64", 5),
               (4096, "c128", 401), (4096, "c64", 37), (4096, "c128", 1), (16384, "c128", 3), (32768, "c64", 6)]


@pytest.mark.parametrize("n,dtype,nfreq", _HOST_CASES)
def test_red_zones_host_api(n, dtype, nfreq, guards, oracle):
    """The host-pointer calls under red zones: pinned staging, the device slab + copy path, the surface written IN PLACE
    into caf_host_alloc memory (itself fenced: a store past the last row would land in its red zone), xcor,
    apply_freq_shift, find_peak and the two views.  Results against the ORACLE."""
    caf = guards
    rng = np.random.default_rng(n + 3 * nfreq)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    rdt = np.float64 if dtype == "c128" else np.float32
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.linspace(-80.0, 80.0, nfreq) if nfreq > 1 else np.array([12.5])
    x, y = _planted(rng, n, FS, float(fr[nfreq // 2]), min(9, n // 4), cdt)
    osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
    def body(e):
        pinned = e.host_empty((nfreq, 2 * n), rdt)
        for out in (None, pinned):
            surf, ridx, rval, pk = e.surface_arrays(x, y, fr, FS, dtype=dtype, out=out)
            assert np.max(np.abs(surf - osurf)) <= tol * osurf.max()
            assert (pk.freq, pk.idx) == oracle.np_find_peak(fr, oidx, oval)
            _check_guards(caf)
        _, _, _, pk = e.surface_arrays(x, y, fr, FS, dtype=dtype, want_surface=False)
        assert (pk.freq, pk.idx) == oracle.np_find_peak(fr, oidx, oval)
        got = e.xcor(x, y)
        want = oracle.np_xcor(x.astype(np.complex128), y.astype(np.complex128))
        assert np.max(np.abs(got - want)) <= (1e-9 if dtype == "c128" else 2e-3) * np.max(np.abs(want))
        sh = e.apply_freq_shift(x, 12.5, FS)
        assert np.max(np.abs(sh - oracle.np_apply_freq_shift_fast(x.astype(np.complex128), 12.5, FS))) <= (1e-11 if dtype == "c128" else 1e-5)   # (phase argument error grows with the sample index)
        rows = e.caf_surface(x, y, fr, FS, want_surface=False, dtype=dtype)
        assert e.find_peak(rows) == oracle.np_find_peak(fr, oidx, oval)
        if n >= 2:
            for view in ("go", "python"):
                v = e.surface_view(surf, view)
                assert v.shape == (nfreq, 2 * n if view == "go" else n)
            vp = e.host_empty((nfreq, 2 * n), rdt)
            vp[:] = surf
            assert np.array_equal(e.surface_view(vp, "go"), e.surface_view(surf, "go"))
        _check_guards(caf, min_allocs=3)

    e = caf.Engine(0)
    try:
        body(e)          # (its pinned arrays die with the call: the engine refuses to close under live host_empty() memory)
    finally:
        e.close()


@pytest.mark.parametrize("n,dtype,batch,nslots,split", [(4096, "c128", 1, 2, False), (4096, "c64", 1, 4, False),
                                                       (4096, "c128", 8, 4, False), (4096, "c128", 4, 2, True),
                                                       (1024, "c64", 3, 2, False), (64, "c128", 5, 2, True),
                                                       (32768, "c64", 1, 2, False)])
def test_red_zones_streaming(n, dtype, batch, nslots, split, guards):
    """caf_stream slots under red zones (pinned inputs and results, per-slot device buffers, spectra, slabs, sequence words,
    the surface slabs), in every graph form: batched, split, one-launch, two-node.  19 pairs (ragged last replay); the
    answers are the plants."""
    caf = guards
    rng = np.random.default_rng(n + batch)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    fr = np.linspace(-40.0, 40.0, 21)
    count = 19
    nd = np.empty((count, n), dtype=cdt)
    hs = np.empty((count, n), dtype=cdt)
    lags = []
    for k in range(count):
        lags.append(int(rng.integers(1, n // 4)))
        nd[k], hs[k] = _planted(rng, n, FS, float(fr[k % 21]), lags[-1], cdt)
    e = caf.Engine(0)
    try:
        plan = e.plan(n, fr, FS, dtype=dtype)
        st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, split=split)
        for _ in range(2):
            peaks, ridx, rval = st.run(nd, hs, want_rows=True)
            assert [int(p["idx"]) for p in peaks] == lags
            if n >= 1024:   # (shorter inputs cannot tell rows 4 Hz apart)
                assert [int(p["row"]) for p in peaks] == [k % 21 for k in range(count)]
            _check_guards(caf, min_allocs=5 * nslots)
        st.close()
        plan.close()
    finally:
        e.close()


def test_red_zones_multi_objects(guards, oracle):
    """caf_multi_stream_* (with surfaces) and caf_multi_surface_* (host join, in-place arena, RCCL join with one rank) under
    red zones: their internally created contexts inherit the process-wide setting."""
    caf = guards
    from caf_cookoff_amd.synth import make_batch
    fr = caf.bench_shifts()[::10]
    nd, hs, lags, _ = make_batch(21, 4096, FS, seed0=77)
    ms = caf.MultiStream([0, 0], 4096, fr, FS, nslots=2, want_surface=True)
    peaks, _, _ = ms.run(nd, hs)
    assert [int(p["idx"]) for p in peaks] == lags
    before = _check_guards(caf, min_allocs=20)
    ms.close()
    for rccl, devs in ((False, [0, 0, 0]), (True, [0])):
        mf = caf.MultiSurface(devs, 4096, fr, FS, rccl=rccl)
        arena = mf.host_empty((len(fr), 8192), np.float64)
        surf, ridx, rval, pk = mf.run(nd[0], hs[0], out=arena)
        assert int(pk["idx"]) == lags[0]
        osurf, _, _ = oracle.np_caf_surface(nd[0], hs[0], fr, FS)
        assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
        _check_guards(caf, min_allocs=8)
        del arena, surf
        mf.close()
    assert before > 0


def test_red_zone_checker_sees_a_stray_store(guards):
    """The checker itself: a store one element past a library allocation (done here on purpose, from the test, into the
    pinned red zone behind a caf_host_alloc buffer) is reported with the allocation's size and the offset."""
    caf = guards
    import ctypes
    e = caf.Engine(0)
    try:
        buf = e.host_empty((16,), np.float64)
        assert caf.debug_check_guards()[1] == 0
        ctypes.c_double.from_address(buf.ctypes.data + 16 * 8).value = 1.0      # one past the end
        with pytest.raises(caf.CafError) as ei:
            caf.debug_check_guards()
        assert "128 bytes" in str(ei.value) and "offset 128" in str(ei.value)
        ctypes.memset(buf.ctypes.data + 16 * 8, 0xA5, 8)                       # repair, so that later checks pass
        assert caf.debug_check_guards()[1] == 0
        del buf
    finally:
        e.close()


# ------------------------------------------------------ the long chains (R = 4 / 8 / 16) against the ORACLE --
def _big_fuzz_cases():
    rng = np.random.default_rng(20261005)
    shapes = [(32768, "c64"), (16384, "c128"), (32768, "c128"), (65536, "c64"), (65536, "c128"), (131072, "c64")]   # R = 4, 4, 8, 8, 16, 16
    primes = [1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 40]
    cases = []
    for i in range(18):
        n, dtype = shapes[i % len(shapes)]
        nfreq = int(primes[int(rng.integers(0, len(primes)))])
        lo = int(rng.integers(0, nfreq))
        hi = int(rng.integers(lo + 1, nfreq + 1))
        if i % 3 == 0:
            lo, hi = 0, nfreq
        batch = int(rng.integers(1, 4))
        cases.append((i, n, dtype, nfreq, lo, hi, batch, i % 2 == 1, i % 4 >= 2))
    return cases


@pytest.mark.parametrize("case", _big_fuzz_cases(), ids=lambda c: f"{c[0]}-n{c[1]}-{c[2]}-F{c[3]}-{c[4]}:{c[5]}-b{c[6]}")
def test_fuzz_long_chains_vs_oracle(case, eng, oracle):
    """18 seeded cases over the R = 4 / 8 / 16 chain kernels (n = 16384 ... 131072, both dtypes): nfreq 1 ... 40 incl.
    primes, random row shards, batches of 1 ... 3, shuffled freq lists, negative-lag plants (peak index >= n) -- each checked
    against the numpy ORACLE: up to 12 sampled rows of every surface within 1e-6 / 1e-3 of the maximum, EVERY row peak,
    and the shard peak (row, lag) exact."""
    import torch
    import caf_cookoff_amd as caf
    i, n, dtype, nfreq, lo, hi, batch, shuffled, negative = case
    rng = np.random.default_rng(9000 + i)
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    tol = TOL64 if dtype == "c128" else TOL32
    step = 120.0 / nfreq                                     # jittered grid: rows at least 0.4 * step (>= 1.2 Hz) apart
    fr = (np.arange(nfreq) - nfreq / 2) * step + rng.uniform(-0.3, 0.3, nfreq) * step
    if shuffled:
        rng.shuffle(fr)
    rows = hi - lo
    nd = np.empty((batch, n), dtype=cdt)
    hs = np.empty((batch, n), dtype=cdt)
    want = []
    for b in range(batch):
        r_true = int(rng.integers(lo, hi))
        lag = int(rng.integers(1, n // 4)) * (-1 if negative else 1)
        nd[b], hs[b] = _planted(rng, n, FS, float(fr[r_true]), lag, cdt)
        want.append((r_true, lag % (2 * n)))
    plan = eng.plan(n, fr, FS, dtype=dtype, row_begin=lo, row_end=hi)
    assert plan.path == "chain" and any(f", {R}, " in plan.kernel_name for R in (4, 8, 16)), plan.kernel_name
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.empty((batch, rows, 2 * n), dtype=tdt, device="cuda")
    di = torch.zeros((batch, rows), dtype=torch.int64, device="cuda")
    dv = torch.zeros((batch, rows), dtype=tdt, device="cuda")
    dp = torch.zeros((batch, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), batch, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    for b in range(batch):
        osurf, oidx, oval = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr[lo:hi], FS)
        mx = osurf.max()
        gi, gv = di[b].cpu().numpy(), dv[b].cpu().numpy().astype(np.float64)
        sample = np.unique(np.concatenate([[0, rows - 1, want[b][0] - lo], rng.integers(0, rows, 9)]))
        got = ds[b][torch.from_numpy(sample).cuda()].cpu().numpy()
        assert np.max(np.abs(got - osurf[sample])) <= tol * mx, f"case {i} surface {b}"
        assert np.max(np.abs(gv - oval)) <= tol * mx                                        # every row peak value
        part = np.partition(osurf, -2, axis=1)
        clear = (part[:, -1] - part[:, -2]) > 4 * tol * mx
        assert np.array_equal(gi[clear], oidx[clear].astype(np.int64)), f"case {i} surface {b}: row argmax"
        assert clear[want[b][0] - lo] and int(gi[want[b][0] - lo]) == want[b][1]            # the plant, negative lags included
        pk = dp[b].cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0]
        of, oi = oracle.np_find_peak(fr[lo:hi], oidx, oval)
        assert (pk["freq"], int(pk["idx"]), int(pk["row"])) == (of, oi, want[b][0]), f"case {i} surface {b}: shard peak"
    plan.close()


# ------------------------------------------------------ bench.py's own launch shape against the ORACLE --
@pytest.mark.parametrize("dtype", ["c128", "c64"])
def test_bench_launch_shape_vs_oracle(dtype, eng, oracle, coracle):
    """The headline launch exactly as bench.py issues it (256 distinct pairs x 400 rows, one caf_surface_dev call: the
    row-ticket path, 102 400 rows over 512 / 768 resident workgroups), complex128 (configs[1], k_seq_rows) and complex64
    (configs[2], k_duo_rows): ALL 256 global peaks against find_peak of the C ORACLE's row peaks (complex64: the lag exact,
    the row within one 0.5 Hz step -- neighbouring rows of the 0.5 Hz grid differ by less than f32 resolves on some pairs),
    every row peak value, and 16 sampled surfaces x 4 sampled rows against the numpy ORACLE within 1e-6 / 1e-3 of the
    maximum."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    B, F, n = 256, 400, 4096
    c128 = dtype == "c128"
    tol = TOL64 if c128 else TOL32
    tdt = torch.float64 if c128 else torch.float32
    fr = caf.bench_shifts()
    nd, hs, lags, fos = make_batch(B, n, FS, seed0=1000, dtype=np.complex128 if c128 else np.complex64)
    plan = eng.plan(n, fr, FS, dtype=dtype)
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.empty((B, F, 2 * n), dtype=tdt, device="cuda")
    di = torch.zeros((B, F), dtype=torch.int64, device="cuda")
    dv = torch.zeros((B, F), dtype=tdt, device="cuda")
    dp = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    pk = dp.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    gi, gv = di.cpu().numpy(), dv.cpu().numpy()
    rng = np.random.default_rng(1)
    sampled = set(int(b) for b in rng.choice(B, 16, replace=False))
    for b in range(B):
        x, y = nd[b].astype(np.complex128), hs[b].astype(np.complex128)
        _, oidx, oval = coracle.caf_surface(x, y, fr, FS, want_surface=False, hoist=True, nthreads=8)
        of, oi = oracle.np_find_peak(fr, oidx, oval)
        assert int(pk["idx"][b]) == oi == lags[b], f"surface {b}"
        if c128:
            assert pk["freq"][b] == of, f"surface {b}"
        else:
            assert abs(pk["freq"][b] - of) <= 0.5 + 1e-9, f"surface {b}"
        assert np.max(np.abs(gv[b].astype(np.float64) - oval)) <= tol * oval.max()
        if b in sampled:
            rsel = np.unique(np.concatenate([[int(pk["row"][b])], rng.integers(0, F, 3)]))
            osurf, _, _ = oracle.np_caf_surface(x, y, fr[rsel], FS)
            got = ds[b][torch.from_numpy(rsel).cuda()].cpu().numpy()
            assert np.max(np.abs(got - osurf)) <= tol * oval.max(), f"surface {b}"
            assert np.array_equal(gi[b][rsel], np.argmax(got, axis=1))
    plan.close()


# ------------------------------------------------------ one wave per row (measured and rejected; measurement library) --
def test_wave_row_kernel_variant_vs_oracle(oracle, monkeypatch):
    """k_wave_rows<float> (measure/kernels_wave4096.hpp, CAF_ROW_KERNEL=4): the structural attempt of round 4 stays in the
    measurement library only -- and stays correct: 400 x 8192 complex64 surfaces against the ORACLE within 1e-3 of the
    maximum, peaks exact, ragged shard, batch beyond the resident waves; complex128 plans ignore the switch."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    if not caf.MEASURE_LIB_PATH.exists():
        pytest.skip("measurement library not built")
    monkeypatch.setenv("CAF_ROW_KERNEL", "4")
    caf.debug_guard_bands(4096, lib=caf.MEASURE_LIB_PATH)   # (the measurement library has its own allocation registry)
    eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    try:
        fr = caf.bench_shifts()
        B = 5
        nd, hs, lags, _ = make_batch(B, 4096, FS, seed0=6100, dtype=np.complex64)
        for lo, hi in ((0, 400), (7, 390)):
            plan = eng.plan(4096, fr, FS, dtype="c64", row_begin=lo, row_end=hi)
            assert plan.kernel_name == "caf::k_wave_rows<float>"
            rows = hi - lo
            dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
            ds = torch.empty((B, rows, 8192), dtype=torch.float32, device="cuda")
            di = torch.zeros((B, rows), dtype=torch.int64, device="cuda")
            dv = torch.zeros((B, rows), dtype=torch.float32, device="cuda")
            dp = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
            torch.cuda.synchronize()
            pk = dp.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
            for b in range(B):
                osurf, oidx, oval = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr[lo:hi], FS)
                assert np.max(np.abs(ds[b].cpu().numpy() - osurf)) <= TOL32 * osurf.max()
                of, oi = oracle.np_find_peak(fr[lo:hi], oidx, oval)
                assert (pk["freq"][b], int(pk["idx"][b])) == (of, oi) and oi == lags[b]
            plan.close()
        p128 = eng.plan(4096, fr, FS, dtype="c128")
        assert "k_seq_rows<double" in p128.kernel_name
        p128.close()
        checked, bad = caf.debug_check_guards(lib=caf.MEASURE_LIB_PATH)   # tables, phasors, spectra, ticket word: fences intact
        assert bad == 0 and checked >= 4
    finally:
        eng.close()
        caf.debug_guard_bands(0, lib=caf.MEASURE_LIB_PATH)


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
