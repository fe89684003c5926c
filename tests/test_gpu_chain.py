"""The LDS-resident chain kernels (k_chain_rows<T, LOGM, R>: every n from 1024 to 131072 except 4096; BASELINE configs[3] =
n 32768 complex64) against the oracle: each instantiation, edge shapes, the full-size shard decomposition, the seeded fuzz of
the long chains, and the measurement library's alternative forms (tiled65536, 16 x 4096, 32 points per thread).
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from gpu_common import FS, TOL32, TOL64, _plan_arrays_n, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


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


# ----------------------------------------- (a) BASELINE configs[3] at full size, 8 shards --
def test_config3_full_size_shards_equal_unsharded(eng, oracle):
    """4096 x 65536 complex64 on ONE GPU: the unsharded surface (16 launch chunks) and the eight
    512-row shards an 8-GPU job computes (2 chunks each, rows [r*512,(r+1)*512) on rank r,
    SURVEY.md section 8e) must agree bit for bit -- surface slice, row peaks -- and the reduction
    of the eight shard peaks (reduce_global_peak's rule: max value, lowest global row among
    equals) must equal the unsharded find_peak.  >= 16 sampled rows against the f64 oracle at
    1e-3 of max, the planted (lag, Doppler) recovered."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    n, F, G = 32768, 4096, 8
    fr = np.arange(F) * 0.05 - 102.4
    s0, s1, lag, fo = make_pair(n=n, seed=3, lag=201, foffset=float(fr[1800]), dtype=np.complex64)
    nd, hs = torch.from_numpy(s0[None]).cuda(), torch.from_numpy(s1[None]).cuda()
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        def run(lo, hi):
            plan = eng.plan(n, fr, FS, dtype="c64", row_begin=lo, row_end=hi)
            rows = hi - lo
            surf = torch.full((1, rows, 2 * n), -1.0, dtype=torch.float32, device="cuda")
            ridx = torch.full((1, rows), -1, dtype=torch.int64, device="cuda")
            rval = torch.full((1, rows), -1.0, dtype=torch.float32, device="cuda")
            peak = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
            plan.surface_dev(nd.data_ptr(), hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                             peak.data_ptr())
            torch.cuda.synchronize()
            path = plan.path
            plan.close()
            return surf[0], ridx[0], rval[0], peak, path

        f_surf, f_ridx, f_rval, f_peak, path = run(0, F)
        print("configs[3] path:", path)
        fpk = f_peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
        assert (int(fpk["row"]), float(fpk["freq"]), int(fpk["idx"])) == (1800, float(fr[1800]), lag)
        assert int((f_surf < 0).sum()) == 0, "every lag of every row is written"
        vals, rows_g, idxs = [], [], []
        for r in range(G):
            lo, hi = caf.shard_range(F, r, G)
            assert (lo, hi) == (512 * r, 512 * (r + 1))
            s_surf, s_ridx, s_rval, s_peak, _ = run(lo, hi)
            assert torch.equal(s_surf, f_surf[lo:hi]), f"shard {r}: surface differs from the unsharded rows"
            assert torch.equal(s_ridx, f_ridx[lo:hi]) and torch.equal(s_rval, f_rval[lo:hi])
            pk = s_peak.cpu()
            vals.append(pk[:, 0]); rows_g.append(pk.view(torch.int64)[:, 3]); idxs.append(pk.view(torch.int64)[:, 2])
            del s_surf
        # find_peak over the shards, as dist.reduce_global_peak combines them (no process group here:
        # the same rule spelled out -- max value, then lowest global row among the holders)
        v = torch.stack(vals)[:, 0]
        rw = torch.stack(rows_g)[:, 0]
        ix = torch.stack(idxs)[:, 0]
        gmax = v.max()
        holders = (v == gmax) & (rw >= 0)
        win = int(torch.argmin(torch.where(holders, rw, torch.full_like(rw, 1 << 40))))
        assert (float(gmax), int(rw[win]), int(ix[win])) == (float(fpk["val"]), int(fpk["row"]), int(fpk["idx"]))
        # row peaks are consistent with the stored surface: value = row maximum, index = FIRST lag
        # holding it (torch.argmax does not promise the first of equal values, so spell it out)
        mx = f_surf.max(dim=1).values
        assert torch.equal(mx, f_rval)
        lag_axis = torch.arange(2 * n, device="cuda", dtype=torch.int64)
        for r0 in range(0, F, 256):
            blk = f_surf[r0:r0 + 256]
            first = torch.where(blk == mx[r0:r0 + 256, None], lag_axis, 2 * n).min(dim=1).values
            assert torch.equal(first, f_ridx[r0:r0 + 256])
        # sampled rows against the f64 oracle
        sample = sorted({0, 1, 255, 256, 257, 511, 512, 1023, 1799, 1800, 1801, 2047, 2048, 3071, 3583, 4095, 4094, 777})
        assert len(sample) >= 16
        osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr[sample], FS)
        got = f_surf[sample].cpu().numpy().astype(np.float64)
        smax = float(fpk["val"])
        err = np.max(np.abs(got - osurf)) / smax
        print(f"configs[3] full size: max|d|/max over {len(sample)} sampled rows = {err:.3e}")
        assert err <= TOL32
        clear = np.array([(np.partition(osurf[i], -2)[-1] - np.partition(osurf[i], -2)[-2]) > 1e-4 * smax
                          for i in range(len(sample))])
        assert np.array_equal(f_ridx[sample].cpu().numpy()[clear].astype(np.uint64), oidx[clear])
    finally:
        eng.set_stream(None)


# ------------------------------------------ (7) LDS-resident chain path, every covered n --
CHAIN_CASES = [(1024, "c128", "caf::k_chain_rows<double, 10, 2, 1, 0>"), (2048, "c128", "caf::k_chain_rows<double, 11, 2, 1, 0>"),
               (8192, "c128", "caf::k_chain_rows<double, 13, 2, 1, 0>"), (16384, "c128", "caf::k_chain_rows<double, 13, 4, 1, 0>"),
               (1024, "c64", "caf::k_chain_rows<float, 10, 2, 1, 0>"), (2048, "c64", "caf::k_chain_rows<float, 11, 2, 1, 0>"),
               (8192, "c64", "caf::k_chain_rows<float, 13, 2, 1, 0>"), (16384, "c64", "caf::k_chain_rows<float, 14, 2, 1, 0>"),
               (32768, "c64", "caf::k_chain_rows<float, 14, 4, 1, 0>"), (32768, "c128", "caf::k_chain_rows<double, 13, 8, 1, 0>"),
               (65536, "c64", "caf::k_chain_rows<float, 14, 8, 1, 0>"), (65536, "c128", "caf::k_chain_rows<double, 13, 16, 1, 0>"),
               (131072, "c64", "caf::k_chain_rows<float, 14, 16, 1, 0>")]


@pytest.mark.parametrize("n,dtype,kernel", CHAIN_CASES, ids=lambda v: str(v) if not isinstance(v, str) or len(v) < 6 else None)
def test_chain_path_vs_oracle(n, dtype, kernel, eng, oracle):
    """Every power-of-two n the LDS-resident chain kernels cover (kernels_chain.hpp; "any power
    of two" used to mean log2(L) radix-2 passes over HBM): whole surfaces against the numpy
    restatement of mod.rs:121-166, 1e-6 / 1e-3 of max, row argmax equal wherever the oracle's row
    has a clear winner, planted (lag, Doppler) recovered, negative lag (index >= n), a batch of
    two pairs and a row shard through the device API."""
    import torch
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_pair
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    lag = 7 + n // 37
    s0, s1, _, fo = make_pair(n=n, seed=n + (dtype == "c64"), lag=lag, foffset=-31.5, dtype=cdt)
    fr = np.array([-40.0, -32.0, -31.5, -31.0, 0.0, 31.5, 977.25])
    plan = eng.plan(n, fr, FS, dtype=dtype)
    assert plan.path == "chain" and plan.kernel_name == kernel
    plan.close()
    surf, ridx, rval, peak = eng.surface_arrays(s0, s1, fr, FS, dtype=dtype)
    osurf, oidx, oval = oracle.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr, FS)
    err = np.max(np.abs(surf - osurf)) / osurf.max()
    print(f"chain n={n} {dtype}: max|d|/max = {err:.3e}")
    assert err <= tol
    part = np.partition(osurf, -2, axis=1)
    clear = (part[:, -1] - part[:, -2]) > (1e-9 if dtype == "c128" else 1e-4) * osurf.max()
    assert clear.any() and np.array_equal(ridx[clear], oidx[clear])
    assert (peak.freq, int(peak.idx)) == oracle.np_find_peak(fr, oidx, oval) == (-31.5, lag)
    assert np.array_equal(surf.argmax(axis=1).astype(np.uint64)[clear], ridx[clear])
    assert np.array_equal(surf.max(axis=1), rval)
    # swapped roles: negative lag -> index 2n - lag, Doppler changes sign
    s2, i2, v2, p2 = eng.surface_arrays(s1, s0, fr, FS, dtype=dtype)
    assert (p2.freq, int(p2.idx)) == (31.5, 2 * n - lag)
    # device API: batch of two pairs x row shard [2, 6)
    tdt = torch.float64 if dtype == "c128" else torch.float32
    nd = torch.from_numpy(np.stack([s0, s1])).cuda()
    hs = torch.from_numpy(np.stack([s1, s0])).cuda()
    shard = eng.plan(n, fr, FS, dtype=dtype, row_begin=2, row_end=6)
    d_s = torch.empty((2, 4, 2 * n), dtype=tdt, device="cuda")
    d_i = torch.empty((2, 4), dtype=torch.int64, device="cuda")
    d_v = torch.empty((2, 4), dtype=tdt, device="cuda")
    d_p = torch.empty((2, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    shard.surface_dev(nd.data_ptr(), hs.data_ptr(), 2, d_s.data_ptr(), d_i.data_ptr(), d_v.data_ptr(), d_p.data_ptr())
    eng.synchronize()
    shard.close()
    assert np.array_equal(d_s[0].cpu().numpy(), surf[2:6]) and np.array_equal(d_s[1].cpu().numpy(), s2[2:6])
    assert np.array_equal(d_i[0].cpu().numpy().astype(np.uint64), ridx[2:6])
    pk = d_p.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    assert (int(pk[0]["row"]), int(pk[0]["idx"])) == (2, lag) and (int(pk[1]["row"]), int(pk[1]["idx"])) == (5, 2 * n - lag)
    # all-zero input and a NaN sample (mod.rs:143-151)
    z = np.zeros(n, dtype=cdt)
    sz, iz, vz, pz = eng.surface_arrays(z, z, fr[:2], FS, dtype=dtype)
    assert not sz.any() and (pz.freq, pz.idx, pz.row) == (0.0, 0, -1)
    bad = s0.copy()
    bad[n // 2 + 3] = complex(np.nan, 0.0)
    sn, i_n, vn, pn = eng.surface_arrays(bad, s1, fr[:2], FS, dtype=dtype)
    assert np.isnan(sn).all() and not i_n.any() and (pn.freq, pn.idx, pn.row) == (0.0, 0, -1)


def test_chain_path_edge_shapes(eng, oracle):
    """Chain-path plans at the edges of the device API: an empty frequency list, an empty row
    shard, a single row, and a batch larger than the resident grid's share (rows handed out by
    stride), complex128 n = 2048 and complex64 n = 1024."""
    import torch
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(9)
    for n, dtype, cdt, tdt, tol in ((2048, "c128", np.complex128, torch.float64, TOL64), (1024, "c64", np.complex64, torch.float32, TOL32)):
        a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
        b = (np.roll(a, 11) * np.exp(2j * np.pi * 25.0 * np.arange(n) / FS)).astype(cdt)
        # empty frequency list through the host API
        surf, ridx, rval, peak = eng.surface_arrays(a, b, np.array([]), FS, dtype=dtype)
        assert surf.shape == (0, 2 * n) and (peak.freq, peak.idx, peak.row) == (0.0, 0, -1)
        # empty shard and single-row shard
        fr = np.array([0.0, 25.0, 50.0])
        empty = eng.plan(n, fr, FS, dtype=dtype, row_begin=2, row_end=2)
        assert empty.path == "chain" and empty.rows == 0
        empty.close()
        one = eng.plan(n, fr, FS, dtype=dtype, row_begin=1, row_end=2)
        batch = 700  # > resident workgroups of any chain kernel on 256 CUs for these sizes? no: exercises the stride loop
        nd = torch.from_numpy(np.tile(a, (batch, 1))).cuda()
        hs = torch.from_numpy(np.tile(b, (batch, 1))).cuda()
        hs[5] = 0  # one all-zero haystack in the batch
        d_s = torch.empty((batch, 1, 2 * n), dtype=tdt, device="cuda")
        d_i = torch.empty((batch, 1), dtype=torch.int64, device="cuda")
        d_v = torch.empty((batch, 1), dtype=tdt, device="cuda")
        d_p = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        one.surface_dev(nd.data_ptr(), hs.data_ptr(), batch, d_s.data_ptr(), d_i.data_ptr(), d_v.data_ptr(), d_p.data_ptr())
        eng.synchronize()
        one.close()
        osurf, oidx, oval = oracle.np_caf_surface(a.astype(np.complex128), b.astype(np.complex128), fr[1:2], FS)
        got = d_s.cpu().numpy()
        assert np.max(np.abs(got[0, 0] - osurf[0])) <= tol * osurf.max()
        assert np.array_equal(got[0], got[699]) and np.array_equal(got[0], got[350])   # every batch entry identical
        assert not got[5].any() and int(d_i[5, 0]) == 0
        pk = d_p.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
        assert int(pk[0]["idx"]) == 11 and int(pk[0]["row"]) == 1 and int(pk[5]["row"]) == -1
        assert int(d_i[699, 0]) == 11


# ------------------------------------------------------ configs[3] with 32 points per thread (measured and rejected) --
def test_r32_variant_matches_oracle_and_product_kernel(eng, oracle, monkeypatch):
    """kernels_r32.hpp (16384 = 32 x 32 x 16, 512 threads, two LDS exchanges per transform) lives
    in the MEASUREMENT library (CAF_R32=1): parity-green against the oracle, row argmax and peak equal to the product
    chain kernel's on a multi-row launch that wraps the persistent grid (300 rows > 256 workgroups), and
    bit-identical from run to run.  It runs at the product kernel's speed, not faster -- HISTORY.md section 5."""
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
