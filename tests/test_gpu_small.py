"""Lane-group rows (k_small_rows / k_small: n = 1 ... 512) against the oracle.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from gpu_common import TOL32, TOL64

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


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
