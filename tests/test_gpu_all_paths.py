"""Seeded fuzz over every kernel family against the oracle, and run-to-run bit determinism of every family.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from gpu_common import FS, TOL32, TOL64, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


@pytest.mark.parametrize("n,dtype", [(32768, "c64"), (16384, "c128"), (16384, "c64"), (4096, "c128"), (4096, "c64")])
def test_run_to_run_bit_determinism(n, dtype, eng):
    """The same launch twice, and the same rows as a shard of another plan, give identical BITS
    (no atomics on the data path).  Caught in round 2: a 16-byte scratch-slab store whose data
    registers the following inline-asm arithmetic rewrote two wait states too early (CDNA3 ISA 4.5
    hazard the compiler does not insert s_nop for in front of inline asm): 2 % of a surface's lags
    were stale values of the previous row, within every tolerance but different from run to run."""
    import torch
    from caf_cookoff_amd.synth import make_pair
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    F = 520
    fr = np.arange(F) * 0.05 - 13.0
    s0, s1, lag, fo = make_pair(n=n, seed=5, lag=77, foffset=float(fr[300]), dtype=cdt)
    nd, hs = torch.from_numpy(s0[None]).cuda(), torch.from_numpy(s1[None]).cuda()

    def run(lo, hi):
        plan = eng.plan(n, fr, FS, dtype=dtype, row_begin=lo, row_end=hi)
        rows = hi - lo
        surf = torch.full((1, rows, 2 * n), -1.0, dtype=tdt, device="cuda")
        ridx = torch.zeros((1, rows), dtype=torch.int64, device="cuda")
        rval = torch.zeros((1, rows), dtype=tdt, device="cuda")
        peak = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        plan.surface_dev(nd.data_ptr(), hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
        eng.synchronize()
        plan.close()
        return surf[0], ridx[0], rval[0]

    a = run(0, F)
    b = run(0, F)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    c = run(100, 400)
    assert torch.equal(a[0][100:400], c[0]) and torch.equal(a[1][100:400], c[1])
    assert int(a[1][300]) == lag


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
