"""GPU parity tests: every call goes through the C ABI (libcaf_hip.so) and is
compared with the CPU oracle / the committed golden vectors.

Bars (BASELINE.json north_star):
  * argmax (freq, idx): exact equality with the reference's known answers
  * f64 surface: |d| <= 1e-6 * max(surface)     (per-element relative is meaningless:
    the surface has exact zeros and entries 1e-20 below the peak)
  * f32 surface: |d| <= 1e-3 * max(surface)
"""
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
    cu, name = e.device_info()
    print(f"device: {name}, {cu} CUs")
    yield e
    e.close()


@pytest.fixture(scope="module")
def meng():
    """Engine on the MEASUREMENT build (libcaf_hip_measure.so): rejected kernel variants and the
    CAF_* environment switches live only there."""
    import caf_cookoff_amd as caf
    assert caf.MEASURE_LIB_PATH.exists(), "measurement library missing (make -C caf_cookoff_amd/csrc)"
    e = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    yield e
    e.close()


def _pair(oracle, k):
    return oracle.load_pair(DATA, f"chirp_{k}_raw.c64", oracle.KATS[k][1])


# ----------------------------------------------------------------- KATs ------
def _kats():
    from oracle import caf_oracle as O
    return O.KATS


@pytest.mark.parametrize("kat", _kats(), ids=lambda k: f"chirp{k[0]}")
def test_reference_kats_fused_f64(kat, eng, oracle, golden):
    """caf_rust/tests/test.rs:14-316 through the fused n=4096 kernel."""
    import caf_cookoff_amd as caf
    k, hf, (s, e, st), exp = kat
    nd, hs = caf.load_files(DATA / f"chirp_{k}_raw.c64", DATA / hf)
    fr = caf.gen_float_shifts(s, e, st)
    surface = eng.caf_surface(nd, hs, fr, FS, want_surface=False)
    freq, idx = eng.find_peak(surface)
    assert freq == exp[0] and idx == exp[1]  # assert_eq! semantics: exact
    ridx = np.array([r.xcor_peak_idx for r in surface], dtype=np.uint64)
    rval = np.array([r.xcor_peak_val for r in surface])
    g = golden[f"kat{k}_row_val"]
    assert np.array_equal(ridx, golden[f"kat{k}_row_idx"])
    assert np.max(np.abs(rval - g)) <= TOL64 * g.max()


def test_bench_config_f64_golden(eng, oracle, golden, manifest):
    """BASELINE configs[1]: 400x8192 c128, argmax equality + surface parity."""
    fr = oracle.bench_shifts()
    for k in ("0", "4"):
        m = manifest["bench"][k]
        nd, hs = oracle.load_pair(DATA, m["needle"], m["haystack"])
        surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
        assert (peak.freq, peak.idx) == (m["best_freq"], m["best_idx"])
        assert peak.val == rval[int(peak.row)] and fr[int(peak.row)] == peak.freq
        assert np.array_equal(ridx, golden[f"bench{k}_row_idx"])
        tol = TOL64 * m["surface_max"]
        assert np.max(np.abs(rval - golden[f"bench{k}_row_val"])) <= tol
        assert np.max(np.abs(surf[manifest["full_rows"]] - golden[f"bench{k}_rows"])) <= tol
        assert np.max(np.abs(surf.reshape(-1)[::manifest["stride"]] - golden[f"bench{k}_strided"])) <= tol
        # row peaks are consistent with the stored surface
        assert np.array_equal(surf.argmax(axis=1).astype(np.uint64), ridx)
        assert np.array_equal(surf.max(axis=1), rval)
        err = np.max(np.abs(surf[manifest["full_rows"]] - golden[f"bench{k}_rows"])) / m["surface_max"]
        print(f"chirp_{k} bench: max|d|/max = {err:.3e}")


def test_full_surface_vs_c_oracle(eng, oracle, coracle):
    """Whole 400x8192 surface against the C restatement (own FFT)."""
    fr = oracle.bench_shifts()
    nd, hs = _pair(oracle, 9)
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    osurf, oidx, oval = coracle.caf_surface(nd, hs, fr, FS, hoist=True, nthreads=8)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
    assert np.array_equal(ridx, oidx)
    assert (peak.freq, peak.idx) == coracle.find_peak(fr, oidx, oval)


def test_fused_vs_generic_path_agree(eng, oracle):
    """The same rows through both kernel paths (generic path forced via a plan on
    half-length inputs is a different problem, so compare on n=4096 by calling the
    generic kernels through xcor + apply_freq_shift)."""
    nd, hs = _pair(oracle, 4)
    fr = np.array([82.9, -13.0])
    surf, ridx, rval, _ = eng.surface_arrays(nd, hs, fr, FS)
    z = np.zeros(4096, dtype=np.complex128)
    for r, f in enumerate(fr):
        shifted = eng.apply_freq_shift(np.concatenate([nd, z]), f, FS)   # mod.rs:130,138
        c = eng.xcor(np.concatenate([hs, z]), shifted)                   # mod.rs:139
        mag = c.real ** 2 + c.imag ** 2
        assert np.max(np.abs(mag - surf[r])) <= TOL64 * mag.max()
        assert int(np.argmax(mag)) == int(ridx[r])


# ------------------------------------------------------------- generic path --
@pytest.mark.parametrize("n", [1, 2, 8, 64, 512, 1024, 2048, 8192, 16384])
def test_generic_sizes_vs_oracle(n, eng, oracle):
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    lag = n // 3
    b = np.roll(a, lag) * np.exp(2j * np.pi * 37.5 * np.arange(n) / FS) + 0.01 * rng.standard_normal(n)
    fr = np.array([-75.0, 0.0, 37.5, 75.0, 112.5])
    surf, ridx, rval, peak = eng.surface_arrays(a, b, fr, FS)
    osurf, oidx, oval = oracle.np_caf_surface(a, b, fr, FS)
    assert surf.shape == (5, 2 * n)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
    assert np.array_equal(ridx, oidx)
    assert (peak.freq, peak.idx) == oracle.np_find_peak(fr, oidx, oval)


@pytest.mark.parametrize("n", [8, 64, 4096])
def test_apply_freq_shift_golden(n, eng, golden):
    """mod.rs:46-65 vectors (recurrence) vs direct-phasor kernel: <= 1e-12 absolute
    on O(1) data (the recurrence itself drifts ~2e-14 from the exact phasor)."""
    a = golden[f"vec{n}_a"]
    for tag, f in (("77p77", 77.77), ("m12p5", -12.5)):
        out = eng.apply_freq_shift(a, f, FS)
        g = golden[f"vec{n}_shift_{tag}"]
        assert out[0] == a[0]  # sample 0 is multiplied by 1+0j (mod.rs:57-59)
        assert np.max(np.abs(out - g)) <= 1e-12 * max(1.0, np.max(np.abs(g)))


@pytest.mark.parametrize("n", [8, 64, 4096])
def test_xcor_golden(n, eng, golden):
    import caf_cookoff_amd as caf
    a, b = golden[f"vec{n}_a"], golden[f"vec{n}_b"]
    x = caf.Xcor(n, eng)
    out = x.clone().run(a, b)
    g = golden[f"vec{n}_xcor"]
    assert np.max(np.abs(out - g)) <= 1e-12 * np.max(np.abs(g))


def test_xcor_size_independent_properties(eng):
    """Full-size properties: circular-shift covariance and conjugate symmetry
    xcor(a,b)[k] == conj(xcor(b,a)[-k]); linearity in a."""
    rng = np.random.default_rng(5)
    n = 8192
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xab = eng.xcor(a, b)
    scale = np.max(np.abs(xab))
    assert np.max(np.abs(eng.xcor(np.roll(a, -17), b) - np.roll(xab, -17))) <= 1e-12 * scale
    xba = eng.xcor(b, a)
    assert np.max(np.abs(xab - np.conj(np.roll(xba[::-1], 1)))) <= 1e-12 * scale
    assert np.max(np.abs(eng.xcor(a + 2.5 * c, b) - (xab + 2.5 * eng.xcor(c, b)))) <= 1e-11 * scale


# ---------------------------------------------------------------- edge cases --
def test_edge_cases(eng, oracle):
    import caf_cookoff_amd as caf
    # all-zero inputs: every row (0, 0.0), find_peak == (0.0, 0)  (mod.rs:32-35,143)
    for n in (8, 4096):
        z = np.zeros(n, dtype=np.complex128)
        fr = np.array([5.0, 6.0, 7.0])
        surf, ridx, rval, peak = eng.surface_arrays(z, z, fr, FS)
        assert not surf.any() and not ridx.any() and not rval.any()
        assert (peak.freq, peak.idx, peak.val, peak.row) == (0.0, 0, 0.0, -1)
        rows = eng.caf_surface(z, z, fr, FS)
        assert eng.find_peak(rows) == (0.0, 0)
    # empty frequency list: empty surface, peak (0.0, 0)
    a = np.ones(4096, dtype=np.complex128)
    surf, ridx, rval, peak = eng.surface_arrays(a, a, np.array([]), FS)
    assert surf.shape == (0, 8192) and len(ridx) == 0 and (peak.freq, peak.idx) == (0.0, 0)
    assert eng.find_peak([]) == (0.0, 0)
    # length mismatch asserts like xcor_rustfft.rs:54-55
    with pytest.raises(AssertionError):
        eng.caf_surface(a, a[:2048], [0.0], FS)
    # non power of two -> CAF_ERR_LENGTH
    with pytest.raises(caf.CafError) as ei:
        eng.caf_surface(a[:12], a[:12], [0.0], FS)
    assert ei.value.code == 2
    with pytest.raises(caf.CafError):
        eng.xcor(a[:12], a[:12])
    # exact ties between rows: the FIRST row wins (mod.rs:36 strict '>')
    nd, hs = _pair(oracle, 1)
    fr = np.array([36.0, 36.0, 35.0, 36.0])
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    assert rval[0] == rval[1] == rval[3] and peak.row == 0
    # a delta needle: row peak index is the delay, ties inside a row pick the first lag
    d = np.zeros(4096, dtype=np.complex128)
    d[0] = 1.0
    h = np.zeros(4096, dtype=np.complex128)
    h[100] = 2.0
    h[300] = 2.0  # two (nearly) equal peaks: the row argmax is the FIRST lag that attains the row maximum
    surf, ridx, rval, peak = eng.surface_arrays(d, h, np.array([0.0]), FS)
    first_max = int(np.flatnonzero(surf[0] == surf[0].max())[0])          # mod.rs:148-151: strict '>' scan
    assert int(ridx[0]) == first_max and rval[0] == surf[0, first_max] and abs(rval[0] - 4.0) < 1e-12
    if surf[0, 100] == surf[0, 300]:                                      # bitwise tie: the lower lag must win
        assert int(ridx[0]) == 100
    assert {first_max} <= {100, 300}


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_freq_lists_and_sample_rates(seed, eng, oracle):
    """Seeded random cases of the n = 4096 row kernel against the numpy oracle: irregular
    frequency lists (any order, repeated and large |f| values), other sample rates, random
    complex Gaussian inputs with a planted delay + Doppler; complex128 bar 1e-6 of max, argmax
    of every row equal wherever the oracle's row has a clear winner."""
    rng = np.random.default_rng(seed)
    n = 4096
    fs = int(rng.choice([8000, 48000, 1000000]))
    nf = int(rng.integers(1, 40))
    fr = np.concatenate([rng.uniform(-0.01 * fs, 0.01 * fs, nf), [0.0, -0.25 * fs, 0.01 * fs / 3]])
    rng.shuffle(fr)
    lag = int(rng.integers(0, 300))
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n)
    y = np.roll(x, lag) * np.exp(2j * np.pi * fr[1] * np.arange(n) / fs)
    y[:lag] = 0
    y += 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    surf, ridx, rval, peak = eng.surface_arrays(x, y, fr, fs)
    osurf, oidx, oval = oracle.np_caf_surface(x, y, fr, fs)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
    # rows whose best and second-best lags differ by more than the error bar must agree exactly
    part = np.partition(osurf, -2, axis=1)
    clear = (part[:, -1] - part[:, -2]) > 1e-9 * osurf.max()
    assert clear.any() and np.array_equal(ridx[clear], oidx[clear])
    of, oi = oracle.np_find_peak(fr, oidx, oval)
    assert (peak.freq, int(peak.idx)) == (of, oi) == (fr[1], lag)


def test_negative_lag_and_wraparound(eng, oracle):
    """index >= n means negative lag (circular): needle delayed w.r.t. haystack."""
    rng = np.random.default_rng(11)
    n = 4096
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a[-64:] = 0
    nd = np.roll(a, 40)     # needle is the delayed one -> lag -40 -> index 8192-40
    surf, ridx, rval, peak = eng.surface_arrays(nd, a, np.array([0.0]), FS)
    assert int(ridx[0]) == 8192 - 40
    osurf, oidx, _ = oracle.np_caf_surface(nd, a, np.array([0.0]), FS)
    assert int(oidx[0]) == 8192 - 40 and np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()


# ------------------------------------------------------------------- c64 ------
def test_c64_fused_bench_config(eng, oracle, golden, manifest):
    """BASELINE configs[2]: complex64 / f32 surface, tolerance 1e-3 of max; chirp_4
    (row margin 9.2e-4) must also reproduce the row, chirp_0 (margin 6.8e-6) tau."""
    fr = oracle.bench_shifts()
    for k in ("4", "0"):
        m = manifest["bench"][k]
        nd, hs = oracle.load_pair(DATA, m["needle"], m["haystack"])
        surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS, dtype="c64")
        assert surf.dtype == np.float32
        tol = TOL32 * m["surface_max"]
        assert np.max(np.abs(surf[manifest["full_rows"]].astype(np.float64) - golden[f"bench{k}_rows"])) <= tol
        assert np.max(np.abs(rval.astype(np.float64) - golden[f"bench{k}_row_val"])) <= tol
        assert peak.idx == m["best_idx"]
        assert abs(peak.val - m["peak"]) <= tol
        err = np.max(np.abs(surf[manifest["full_rows"]].astype(np.float64) - golden[f"bench{k}_rows"])) / m["surface_max"]
        print(f"c64 chirp_{k}: max|d|/max = {err:.3e}, peak row {peak.row} ({peak.freq} Hz)")
        if k == "4":
            assert peak.freq == m["best_freq"]


@pytest.mark.parametrize("n", [8, 256, 2048])
def test_c64_generic_sizes(n, eng, oracle):
    rng = np.random.default_rng(100 + n)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    b = (np.roll(a, n // 4) * np.exp(2j * np.pi * 50.0 * np.arange(n) / FS)).astype(np.complex64)
    fr = np.array([0.0, 50.0, 100.0])
    surf, ridx, rval, peak = eng.surface_arrays(a, b, fr, FS, dtype="c64")
    osurf, oidx, oval = oracle.np_caf_surface(a.astype(np.complex128), b.astype(np.complex128), fr, FS)
    assert np.max(np.abs(surf - osurf)) <= TOL32 * osurf.max()
    assert int(ridx[1]) == int(oidx[1]) and peak.row == 1


# ------------------------------------------------------- device-resident path --
def test_plan_batch_and_shards(eng, oracle, golden, manifest):
    """caf_surface_dev: a batch of two different pairs, and the 2-way row shard
    used for multi-GPU, reproduce the single-call result (torch only as allocator)."""
    import torch
    import caf_cookoff_amd as caf
    fr = oracle.bench_shifts()
    pairs = [_pair(oracle, 0), _pair(oracle, 4)]
    nd = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    hs = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        full = eng.plan(4096, fr, FS)
        assert full.path == "fused4096" and full.rows == 400
        surf = torch.empty((2, 400, 8192), dtype=torch.float64, device="cuda")
        ridx = torch.empty((2, 400), dtype=torch.int64, device="cuda")
        rval = torch.empty((2, 400), dtype=torch.float64, device="cuda")
        peak = torch.empty((2, 4), dtype=torch.float64, device="cuda")  # 32-byte caf_peak records
        full.surface_dev(nd.data_ptr(), hs.data_ptr(), 2, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                         peak.data_ptr())
        torch.cuda.synchronize()
        for b, k in enumerate(("0", "4")):
            m = manifest["bench"][k]
            assert np.array_equal(ridx[b].cpu().numpy().astype(np.uint64), golden[f"bench{k}_row_idx"])
            assert np.max(np.abs(surf[b].cpu().numpy()[manifest["full_rows"]] - golden[f"bench{k}_rows"])) \
                <= TOL64 * m["surface_max"]
            pk = np.frombuffer(peak[b].cpu().numpy().tobytes(), dtype=[("val", "<f8"), ("freq", "<f8"),
                                                                        ("idx", "<u8"), ("row", "<i8")])[0]
            assert (pk["freq"], pk["idx"]) == (m["best_freq"], m["best_idx"])
        # 2-way shard of the freq list (what rank 0 / rank 1 of a 2-GPU job run)
        vals, keys = [], []
        for rank in range(2):
            lo, hi = caf.shard_range(400, rank, 2)
            sh = eng.plan(4096, fr, FS, row_begin=lo, row_end=hi)
            assert sh.rows == hi - lo
            r_i = torch.empty((1, sh.rows), dtype=torch.int64, device="cuda")
            r_v = torch.empty((1, sh.rows), dtype=torch.float64, device="cuda")
            pk_t = torch.empty((1, 4), dtype=torch.float64, device="cuda")
            sh.surface_dev(nd[0].data_ptr(), hs[0].data_ptr(), 1, None, r_i.data_ptr(), r_v.data_ptr(),
                           pk_t.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(r_v[0], rval[0, lo:hi]) and torch.equal(r_i[0], ridx[0, lo:hi])
            pk = np.frombuffer(pk_t[0].cpu().numpy().tobytes(), dtype=[("val", "<f8"), ("freq", "<f8"),
                                                                       ("idx", "<u8"), ("row", "<i8")])[0]
            vals.append(float(pk["val"]))
            keys.append((int(pk["row"]), int(pk["idx"]), float(pk["freq"])))
            sh.close()
        best = max(vals)
        row, idx, freq = min(k for v, k in zip(vals, keys) if v == best)
        assert (freq, idx) == (69.0, 202) and row == 338
        full.close()
    finally:
        torch.cuda.synchronize()
        eng.set_stream(None)


def test_batch_of_all_kat_pairs_ticket_path(eng, oracle, golden):
    """Ten surfaces in one launch (4000 rows: the dynamic row-ticket assignment; one surface
    alone takes the static stride) must equal the single-surface results bit for bit, on a
    second launch as well (the ticket counter is re-armed by the prepare kernel)."""
    import torch
    fr = oracle.bench_shifts()
    pairs = [_pair(oracle, k) for k in range(10)]
    nd = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    hs = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    plan = eng.plan(4096, fr, FS)
    surf = torch.empty((10, 400, 8192), dtype=torch.float64, device="cuda")
    ridx = torch.empty((10, 400), dtype=torch.int64, device="cuda")
    rval = torch.empty((10, 400), dtype=torch.float64, device="cuda")
    peak = torch.empty((10, 4), dtype=torch.float64, device="cuda")
    one_s = torch.empty((1, 400, 8192), dtype=torch.float64, device="cuda")
    one_i = torch.empty((1, 400), dtype=torch.int64, device="cuda")
    one_v = torch.empty((1, 400), dtype=torch.float64, device="cuda")
    one_p = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    for rep in range(2):
        surf.fill_(-1.0)
        eng.synchronize(); torch.cuda.synchronize()
        plan.surface_dev(nd.data_ptr(), hs.data_ptr(), 10, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                         peak.data_ptr())
        eng.synchronize()
        for b in range(10):
            plan.surface_dev(nd[b].data_ptr(), hs[b].data_ptr(), 1, one_s.data_ptr(), one_i.data_ptr(),
                             one_v.data_ptr(), one_p.data_ptr())
            eng.synchronize()
            assert torch.equal(surf[b], one_s[0]) and torch.equal(ridx[b], one_i[0]) and torch.equal(rval[b], one_v[0])
            assert torch.equal(peak[b], one_p[0])
    # bench-config goldens for the two pairs that have them
    for b in (0, 4):
        assert np.array_equal(ridx[b].cpu().numpy().astype(np.uint64), golden[f"bench{b}_row_idx"])
    plan.close()


def test_ragged_shard_batches_both_row_assignments(eng, oracle):
    """Odd row shard (rows 13..390 of the 400) with 9 surfaces (3393 rows: ticket path) and with
    2 surfaces (754 rows: static stride), complex128 and complex64: every surface must equal the
    one-surface call of the same plan bit for bit, and the c128 argmax rows the numpy oracle."""
    import torch
    fr = oracle.bench_shifts()
    pairs = [_pair(oracle, k) for k in range(9)]
    for dtype, tdt, cdt in (("c128", torch.float64, np.complex128), ("c64", torch.float32, np.complex64)):
        nd = torch.from_numpy(np.stack([p[0] for p in pairs]).astype(cdt)).cuda()
        hs = torch.from_numpy(np.stack([p[1] for p in pairs]).astype(cdt)).cuda()
        plan = eng.plan(4096, fr, FS, dtype=dtype, row_begin=13, row_end=390)
        rows = plan.rows
        assert rows == 377
        one_s = torch.empty((1, rows, 8192), dtype=tdt, device="cuda")
        one_i = torch.empty((1, rows), dtype=torch.int64, device="cuda")
        one_v = torch.empty((1, rows), dtype=tdt, device="cuda")
        one_p = torch.empty((1, 4), dtype=torch.float64, device="cuda")
        singles = []
        for b in range(9):
            plan.surface_dev(nd[b].data_ptr(), hs[b].data_ptr(), 1, one_s.data_ptr(), one_i.data_ptr(),
                             one_v.data_ptr(), one_p.data_ptr())
            eng.synchronize()
            singles.append((one_s[0].clone(), one_i[0].clone(), one_v[0].clone(), one_p[0].clone()))
        for batch in (9, 2):
            surf = torch.full((batch, rows, 8192), -1.0, dtype=tdt, device="cuda")
            ridx = torch.empty((batch, rows), dtype=torch.int64, device="cuda")
            rval = torch.empty((batch, rows), dtype=tdt, device="cuda")
            peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            plan.surface_dev(nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                             peak.data_ptr())
            eng.synchronize()
            for b in range(batch):
                s1, i1, v1, p1 = singles[b]
                assert torch.equal(surf[b], s1) and torch.equal(ridx[b], i1) and torch.equal(rval[b], v1)
                assert torch.equal(peak[b], p1)
        if dtype == "c128":
            osurf, oidx, oval = oracle.np_caf_surface(pairs[3][0], pairs[3][1], fr[13:390], FS)
            assert np.array_equal(singles[3][1].cpu().numpy().astype(np.uint64), oidx)
            assert np.max(np.abs(singles[3][0].cpu().numpy() - osurf)) <= TOL64 * osurf.max()
        plan.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 3], ids=["sequential", "lane-half", "radix8", "two-chain"])
def test_row_kernel_variants_agree(variant, meng, oracle, golden, monkeypatch):
    """All four n = 4096 row kernels (CAF_ROW_KERNEL of the MEASUREMENT library)
    produce the reference's answer: the product uses 0 for complex128 and 3 for complex64, the
    others are measurement variants and must stay parity-green."""
    eng = meng
    monkeypatch.setenv("CAF_ROW_KERNEL", str(variant))
    fr = oracle.bench_shifts()
    nd, hs = _pair(oracle, 0)
    for dtype, tol in (("c128", TOL64), ("c64", TOL32)):
        plan = eng.plan(4096, fr, FS, dtype=dtype)
        surf, ridx, rval, peak = _plan_arrays(plan, eng, nd, hs, dtype)
        plan.close()
        g = golden["bench0_row_val"]
        assert (peak["freq"], int(peak["idx"])) == (69.0, 202)
        assert np.max(np.abs(rval - g)) <= tol * g.max()
        if dtype == "c128":
            assert np.array_equal(ridx.astype(np.uint64), golden["bench0_row_idx"])


def _plan_arrays(plan, eng, nd, hs, dtype):
    import torch
    import caf_cookoff_amd as caf
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    d_nd = torch.from_numpy(nd.astype(cdt)[None]).cuda()
    d_hs = torch.from_numpy(hs.astype(cdt)[None]).cuda()
    surf = torch.empty((1, plan.rows, 8192), dtype=tdt, device="cuda")
    ridx = torch.empty((1, plan.rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((1, plan.rows), dtype=tdt, device="cuda")
    peak = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(d_nd.data_ptr(), d_hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                     peak.data_ptr())
    eng.synchronize()
    pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
    return surf[0].cpu().numpy(), ridx[0].cpu().numpy(), rval[0].cpu().numpy().astype(np.float64), pk


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


# ------------------------------------------------------------ configs[3] shape --
def test_L65536_c64_rows(eng, oracle):
    """BASELINE configs[3] geometry (n = 32768 -> L = 65536, complex64) on a few Doppler
    rows: the chain path (4 chains of 16384 points) against the f64 oracle, tolerance 1e-3 of
    max, and the synthetic pair's known (lag, Doppler) recovered."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    n = 32768
    s0, s1, lag, fo = make_pair(n=n, seed=77, lag=173, foffset=12.0, dtype=np.complex64)
    fr = np.array([11.0, 11.5, 12.0, 12.5, 13.0])
    plan = eng.plan(n, fr, FS, dtype="c64")
    assert plan.path == "chain" and plan.kernel_name == "caf::k_chain_rows<float, 14, 4, 1, 0>"
    plan.close()
    surf, ridx, rval, peak = eng.surface_arrays(s0, s1, fr, FS, dtype="c64")
    osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr, FS)
    assert surf.shape == (5, 65536)
    assert np.max(np.abs(surf - osurf)) <= TOL32 * osurf.max()
    assert np.array_equal(ridx, oidx)
    assert (peak.freq, peak.idx) == (12.0, lag) == oracle.np_find_peak(fr, oidx, oval)
    print(f"L=65536 c64: max|d|/max = {np.max(np.abs(surf - osurf)) / osurf.max():.3e}")


def test_L65536_c128_chain_path_and_negative_lag(eng, oracle):
    """Same geometry in complex128 (chain path, 8 chains of 8192 points) (tolerance 1e-6 of max), needle delayed w.r.t. the haystack
    (negative lag -> index >= n), all-zero input, and a 2-surface batch through the plan."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    n = 32768
    s0, s1, lag, fo = make_pair(n=n, seed=78, lag=90, foffset=-7.5)
    fr = np.array([7.0, 7.5, 8.0])                                      # swapped roles: lag AND Doppler change sign
    surf, ridx, rval, peak = eng.surface_arrays(s1, s0, fr, FS)
    osurf, oidx, oval = oracle.np_caf_surface(s1, s0, fr, FS)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
    assert np.array_equal(ridx, oidx) and (peak.freq, int(peak.idx)) == (7.5, 65536 - lag)
    z = np.zeros(n, dtype=np.complex128)
    surf, ridx, rval, peak = eng.surface_arrays(z, z, fr, FS)
    assert not surf.any() and (peak.freq, peak.idx, peak.row) == (0.0, 0, -1)
    # batch of two surfaces, no surface output
    plan = eng.plan(n, np.array([-7.5, 7.5]), FS)
    nd = torch.from_numpy(np.stack([s0, s1])).cuda()
    hs = torch.from_numpy(np.stack([s1, s0])).cuda()
    r_i = torch.empty((2, 2), dtype=torch.int64, device="cuda")
    r_v = torch.empty((2, 2), dtype=torch.float64, device="cuda")
    pk = torch.empty((2, 4), dtype=torch.float64, device="cuda")
    plan.surface_dev(nd.data_ptr(), hs.data_ptr(), 2, None, r_i.data_ptr(), r_v.data_ptr(), pk.data_ptr())
    eng.synchronize()
    torch.cuda.synchronize()
    pkn = pk.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    assert (pkn[0]["freq"], int(pkn[0]["idx"])) == (-7.5, lag) and (pkn[1]["freq"], int(pkn[1]["idx"])) == (7.5, 65536 - lag)
    plan.close()


def test_L65536_two_pass_variant(meng, oracle, monkeypatch):
    """CAF_BIG_PATH=1 (measurement library): the 16 x 4096 two-pass form of the n = 32768 row
    (kernels_q65536.hpp) against the numpy oracle, complex64 and complex128."""
    from caf_cookoff_amd.synth import make_pair
    eng = meng
    monkeypatch.setenv("CAF_BIG_PATH", "1")
    monkeypatch.setenv("CAF_CHAIN", "0")
    n = 32768
    fr = np.array([11.5, 12.0, 12.5, -3.0])
    for dtype, cdt, tol in (("c64", np.complex64, TOL32), ("c128", np.complex128, TOL64)):
        s0, s1, lag, fo = make_pair(n=n, seed=91, lag=77, foffset=12.0, dtype=cdt)
        plan = eng.plan(n, fr, FS, dtype=dtype)
        assert plan.path == "tiled65536" and "k_q_rows" in plan.kernel_name
        surf, ridx, rval, pk = _plan_arrays_n(plan, eng, s0, s1, dtype, n)
        plan.close()
        osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr, FS)
        assert np.max(np.abs(surf - osurf)) <= tol * osurf.max()
        assert (pk["freq"], int(pk["idx"])) == (12.0, lag)
        if dtype == "c128":
            assert np.array_equal(ridx.astype(np.uint64), oidx)


def _plan_arrays_n(plan, eng, nd, hs, dtype, n):
    import torch
    import caf_cookoff_amd as caf
    tdt = torch.float64 if dtype == "c128" else torch.float32
    d_nd, d_hs = torch.from_numpy(nd[None]).cuda(), torch.from_numpy(hs[None]).cuda()
    surf = torch.empty((1, plan.rows, 2 * n), dtype=tdt, device="cuda")
    ridx = torch.empty((1, plan.rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((1, plan.rows), dtype=tdt, device="cuda")
    peak = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(d_nd.data_ptr(), d_hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                     peak.data_ptr())
    eng.synchronize()
    pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
    return surf[0].cpu().numpy().astype(np.float64), ridx[0].cpu().numpy(), rval[0].cpu().numpy(), pk


@pytest.mark.parametrize("dtype", ["c64", "c128"])
def test_tiled65536_via_measurement_build(meng, oracle, monkeypatch, dtype):
    """The four-step tiled path of round 1 (kernels_big65536.hpp): the product's n = 32768 plans moved
    to the chain path in round 2 (complex64 as 4 x 16384, complex128 as 8 x 8192), so the tiled form
    lives in the measurement library only, behind CAF_CHAIN=0 -- kept as the A/B partner of the chain
    kernels, and still parity-green."""
    from caf_cookoff_amd.synth import make_pair
    monkeypatch.setenv("CAF_CHAIN", "0")
    n = 32768
    cdt, tol = (np.complex64, TOL32) if dtype == "c64" else (np.complex128, TOL64)
    s0, s1, lag, fo = make_pair(n=n, seed=77, lag=173, foffset=12.0, dtype=cdt)
    fr = np.array([11.5, 12.0, 12.5])
    plan = meng.plan(n, fr, FS, dtype=dtype)
    assert plan.path == "tiled65536" and "k_big_rows" in plan.kernel_name
    surf, ridx, rval, pk = _plan_arrays_n(plan, meng, s0, s1, dtype, n)
    plan.close()
    osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr, FS)
    assert np.max(np.abs(surf - osurf)) <= tol * osurf.max() and (pk["freq"], int(pk["idx"])) == (12.0, lag)


def test_generic_path_still_covers_other_big_sizes(eng, oracle):
    """n = 131072 complex128 (L = 262144) exceeds every LDS-resident form (sixteen chains of 8192 points end at
    n = 65536): generic HBM-pass path."""
    from caf_cookoff_amd.synth import make_pair
    n = 131072
    s0, s1, lag, fo = make_pair(n=n, seed=79, lag=33, foffset=5.0)
    fr = np.array([4.5, 5.0, 5.5])
    plan = eng.plan(n, fr, FS)
    assert plan.path == "generic"
    plan.close()
    surf, ridx, rval, peak = eng.surface_arrays(s0, s1, fr, FS)
    osurf, oidx, oval = oracle.np_caf_surface(s0, s1, fr, FS)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max() and (peak.freq, peak.idx) == (5.0, lag)


# ------------------------------------------------- "next" rows of SURVEY.md 8(f) --
def test_go_and_python_views(eng, oracle):
    """The other cook-off implementations' conventions as views of the same surface,
    checked against direct restatements of caf_go/caf.go:95-116 and caf_python/caf.py:15-18."""
    from scipy import signal
    nd, hs = _pair(oracle, 4)
    n = len(nd)
    fr = np.array([82.5, 83.0, 83.5])
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    go = eng.surface_view(surf, "go")
    py = eng.surface_view(surf, "python")
    assert go.shape == (3, 2 * n) and py.shape == (3, n)
    z = np.zeros(n, dtype=np.complex128)
    for r, f in enumerate(fr):
        shifted = nd * np.exp(2j * np.pi * f * np.arange(n) / FS)          # apply_fdoa (caf.go:118-126)
        corr = np.fft.ifft(np.fft.fft(np.concatenate([shifted, z])) *
                           np.conj(np.fft.fft(np.concatenate([z, hs]))))   # xcor (caf.go:95-116)
        assert np.max(np.abs(go[r] - np.abs(corr))) <= 1e-9 * np.abs(corr).max()
        same = np.abs(signal.correlate(shifted, hs, mode="same", method="fft"))  # caf.py:15-18
        assert np.max(np.abs(py[r] - same)) <= 1e-9 * same.max()
    # main.go:35 and caf.py:145 recover the same (tau, f) as find_peak
    fdx, tdx = np.unravel_index(np.argmax(go), go.shape)
    assert (n - tdx, fr[fdx]) == (peak.idx, peak.freq) == (70, 83.0)
    fdx, tmax = np.unravel_index(np.argmax(py), py.shape)
    assert (n // 2 - tmax, fr[fdx]) == (70, 83.0)


def test_python_view_vs_reference_amb_surf_fixture(eng, oracle):
    """CAF_VIEW_PYTHON against the output of the reference's OWN caf_python/caf.py amb_surf
    (caf.py:89-117) on its __main__ pair (caf.py:126-133), stored by tests/golden/make_py_fixture.py.
    The reference computes in complex64 (np.empty_like(ray), caf.py:30; scipy correlate keeps
    single precision): tolerance 2e-6 of the surface maximum (measured 2.1e-7)."""
    from conftest import GOLDEN
    g = np.load(GOLDEN / "py_amb_surf.npz")
    nd, hs = oracle.load_pair(DATA, str(g["needle"]), str(g["haystack"]))
    fr = g["freqs"]
    assert np.array_equal(fr, oracle.bench_shifts())  # np.arange(-100, 100, .5) == the Rust bench grid
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    py = eng.surface_view(surf, "python")
    assert py.shape == tuple(g["shape"])
    tol = 2e-6 * g["row_max"].max()
    assert np.max(np.abs(py[g["full_rows"]] - g["rows"])) <= tol
    assert np.max(np.abs(py.reshape(-1)[::int(g["stride"])] - g["strided"])) <= tol
    assert np.max(np.abs(py.max(axis=1) - g["row_max"])) <= tol
    fmax, tmax = np.unravel_index(py.argmax(), py.shape)                  # caf.py:144-146
    assert (len(nd) // 2 - tmax, fr[fmax]) == (int(g["tau"]), float(g["freq"])) == (70, 83.0)
    assert (peak.idx, peak.freq) == (70, 83.0)


def test_refine_peak_coarse_to_fine(eng, oracle):
    """Coarse 1 Hz grid then the fine grids of the reference's KATs (test.rs:174,212)."""
    for k, coarse, fine, want in ((2, (25.0, 40.0, 1.0), 0.05, (32.15, 169)), (4, (70.0, 100.0, 1.0), 0.1, (82.9, 70))):
        nd, hs = _pair(oracle, k)
        cf = oracle.gen_float_shifts(*coarse)
        (cfq, cidx), (ffq, fidx), ff = eng.refine_peak(nd, hs, FS, cf, fine)
        assert cidx == want[1] and abs(cfq - want[0]) <= 0.5
        assert fidx == want[1] and abs(ffq - want[0]) < 1e-9


def test_cli_demo(tmp_path):
    import subprocess, sys
    from conftest import ROOT
    dump = tmp_path / "surf.bin"
    r = subprocess.run([sys.executable, "-m", "caf_cookoff_amd", str(DATA / "chirp_0_raw.c64"),
                        str(DATA / "chirp_0_T+202samp_F+69.25Hz.c64"), "--dump-surf", str(dump), "--view", "go"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["Frequency offset: 69.0Hz", "Time offset: 202 samples (4.208ms)"]  # main.rs:29-31
    surf = np.fromfile(dump, dtype="<f8").reshape(400, 8192)
    fdx, tdx = np.unravel_index(np.argmax(surf), surf.shape)
    assert 4096 - tdx == 202  # main.go:35


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
    # bad shard range / dtype / length / fs
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, _lib.CAF_C128, 2, 1, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, _lib.CAF_C128, 0, 4, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, FS, 7, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert lib.caf_plan_create(eng._h, 4095, dp, 3, FS, _lib.CAF_C128, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_LENGTH
    assert lib.caf_plan_create(eng._h, 0, dp, 3, FS, _lib.CAF_C128, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_LENGTH
    assert lib.caf_plan_create(eng._h, 4096, dp, 3, 0, _lib.CAF_C128, 0, 3, ctypes.byref(h)) == _lib.CAF_ERR_BAD_ARG
    assert b"fs == 0" in lib.caf_last_error_string()
    assert h.value is None
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


def test_apply_freq_shift_c64_and_find_peak_direct(eng, golden):
    """The c64 twin of mod.rs:46-65 (phase in f64, one rounding) and find_peak on caller rows."""
    from caf_cookoff_amd import CafSurfaceRow
    a = golden["vec4096_a"].astype(np.complex64)
    out = eng.apply_freq_shift(a, 77.77, FS)
    assert out.dtype == np.complex64 and out[0] == a[0]
    ref = golden["vec4096_shift_77p77"]
    assert np.max(np.abs(out.astype(np.complex128) - ref)) <= 2e-7 * np.max(np.abs(ref)) + 1e-9
    rows = [CafSurfaceRow(1.0, None, 10, 3.0), CafSurfaceRow(2.0, None, 20, 5.0), CafSurfaceRow(3.0, None, 30, 5.0),
            CafSurfaceRow(4.0, None, 40, 0.0)]
    assert eng.find_peak(rows) == (2.0, 20)           # first strictly-greater row wins (mod.rs:36)
    assert eng.find_peak(rows[3:]) == (0.0, 0)        # nothing above the initial 0.0 (mod.rs:32-35)


# ------------------------------------------------ full-size domain properties --
def test_full_size_properties(eng, oracle):
    """Size-independent properties on the full 400 x 8192 shape (no oracle involved):
    |alpha|^2 scaling of the surface, delay covariance of the lag axis, and invariance of
    the row peaks under a common phase rotation of both inputs."""
    nd, hs = _pair(oracle, 7)
    fr = oracle.bench_shifts()
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    smax = surf.max()
    # scaling: needle * alpha  ->  surface * |alpha|^2, same argmax everywhere
    alpha = 0.5 - 1.25j
    s2, i2, v2, p2 = eng.surface_arrays(alpha * nd, hs, fr, FS)
    assert np.max(np.abs(s2 - abs(alpha) ** 2 * surf)) <= 1e-12 * abs(alpha) ** 2 * smax
    assert np.array_equal(i2, ridx) and (p2.freq, p2.idx) == (peak.freq, peak.idx)
    # common phase rotation of both inputs leaves |.|^2 unchanged
    rot = np.exp(0.7j)
    s3, i3, v3, p3 = eng.surface_arrays(rot * nd, rot * hs, fr, FS)
    assert np.max(np.abs(s3 - surf)) <= 1e-12 * smax and np.array_equal(i3, ridx)
    # delay covariance: a haystack with exact zeros at both ends, delayed by d samples (nothing
    # wraps or is truncated), moves every lag of the 2n-periodic lag axis by d
    d = 37
    hs_a = np.concatenate([np.zeros(50, dtype=nd.dtype), nd[:4096 - 100] * np.exp(2j * np.pi * 20.0 * np.arange(3996) / FS),
                           np.zeros(50, dtype=nd.dtype)])
    hs_b = np.concatenate([np.zeros(d, dtype=nd.dtype), hs_a[:-d]])
    fr2 = fr[::16]
    s_a, i_a, v_a, p_a = eng.surface_arrays(nd, hs_a, fr2, FS)
    s_b, i_b, v_b, p_b = eng.surface_arrays(nd, hs_b, fr2, FS)
    assert np.max(np.abs(np.roll(s_a, d, axis=1) - s_b)) <= 1e-12 * s_a.max()
    assert (p_b.freq, p_b.idx) == (p_a.freq, p_a.idx + d) and p_a.idx == 50


def test_short_soak():
    """Three seconds of tools/soak.py: random (dtype, batch, row shard) cases, batched results must
    equal one-surface results bit for bit (static and ticket row assignment, both product kernels)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "soak.py"), "3"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout + r.stderr


def test_short_stream_soak():
    """tools/stream_soak.py, 2048 surfaces x 6 rounds x 5 streaming forms x 2 dtypes: every row peak and record of
    every round equals round 0's bit for bit.  (What it caught in round 2: row words written to the pinned
    result buffers with plain stores reached the host after the sequence word of the launch, 1-3 surfaces in
    10^5; they are system-scope stores now.)"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "stream_soak.py"), "2048", "6"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "SOAK ok" in r.stdout, r.stdout + r.stderr

