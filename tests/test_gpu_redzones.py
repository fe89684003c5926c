"""Red zones (caf_debug_guard_bands) and fenced caller buffers around everything a kernel writes, over every kernel family.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from gpu_common import FS, TOL32, TOL64, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


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
