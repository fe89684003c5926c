"""Multi-GPU decompositions behind the C ABI on this one-GPU box (device ids repeat): Doppler-row shards of one surface
(caf_multi_surface_run), of B surfaces per call (caf_multi_surface_run_batch), host join and in-process RCCL join with one
rank; surface-parallel streams (caf_multi_stream_*); the torch.distributed peak reduction through RCCL with one rank.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import DATA
from gpu_common import FS, TOL32, TOL64, _dev_view, _pair, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


def test_peak_reduction_through_rccl_single_rank():
    """The global-peak exchange of the row-sharded multi-GPU path (dist.reduce_global_peak) through the REAL
    RCCL backend, in a group of one rank on this GPU (a child process: the process group must not leak into
    the test process): int64 bit-pattern all_gather and the MAX / MIN-key all_reduce form on device tensors,
    ties, a surface without a peak -- equal to the identity.  (More ranks cannot share one GPU under RCCL;
    the N-rank logic is covered on gloo in tests/test_dist_gloo.py.)"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from caf_cookoff_amd.dist import reduce_global_peak
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
val = torch.tensor([3.5, 0.0, 7.25, 7.25], dtype=torch.float64, device="cuda")
row = torch.tensor([12, -1, 399, 0], dtype=torch.int64, device="cuda")
idx = torch.tensor([202, 0, 8191, 70], dtype=torch.int64, device="cuda")
for method in ("allgather", "allreduce"):
    g, r, i = reduce_global_peak(val, row, idx, method=method, always_collective=True)
    torch.cuda.synchronize()
    assert g.tolist() == [3.5, 0.0, 7.25, 7.25], (method, g)
    assert r.tolist() == [12, -1, 399, 0] and i.tolist() == [202, 0, 8191, 70], (method, r, i)
# the same exchange with the library's element kernels either side of the two collectives (dist.PeakExchange): same answer
import numpy as np
import caf_cookoff_amd as caf
from caf_cookoff_amd.dist import PeakExchange
eng = caf.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
freqs = np.arange(400) * 0.5 - 100.0
rec = torch.zeros((4, 4), dtype=torch.float64, device="cuda")
rec[:, 0] = val
reci = rec.view(torch.int64)
reci[:, 2], reci[:, 3] = idx, row
px = PeakExchange(eng, 4, freqs, torch.device("cuda", 0))
out = px(rec, always_collective=True)
g, r, i = PeakExchange.peaks_of(out)
torch.cuda.synchronize()
assert g.tolist() == [3.5, 0.0, 7.25, 7.25] and r.tolist() == [12, -1, 399, 0] and i.tolist() == [202, 0, 8191, 70], (g, r, i)
assert out[:, 1].tolist() == [freqs[12], 0.0, freqs[399], freqs[0]]
eng.close()
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
""" % str(root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout + r.stderr


def test_peak_exchange_stages_equal_the_host_rule(eng):
    """caf_peak_exchange_stage (the three element kernels a one-process-per-GPU host puts around ITS OWN two collectives) against
    the host rule of the same join (caf_multi_surface_reduce): 3 shards x 500 surfaces of random shard records with ties across
    shards (values from a small set), shards without a peak and surfaces where no shard has one; the two all-reduces are played
    by torch.maximum / torch.minimum over the shards' buffers, as MAX over f64 and MIN over int64 would."""
    import torch
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(77)
    G, count, F = 3, 500, 400
    freqs = np.arange(F) * 0.5 - 100.0
    shards = np.zeros((G, count), dtype=caf.Stream.PEAK_DTYPE)
    for g in range(G):
        lo, hi = caf.multi_surface_shard(F, G, g)
        has = rng.random(count) < 0.8
        shards[g]["row"] = np.where(has, rng.integers(lo, hi, count), -1)
        shards[g]["val"] = np.where(has, rng.choice([1.0, 2.5, 2.5, 7.0, 0.0], count), 0.0)
        shards[g]["idx"] = np.where(has, rng.integers(0, 8192, count), 0)
        shards[g]["freq"] = np.where(has, freqs[np.maximum(shards[g]["row"], 0)], 0.0)
    shards[:, :7]["row"] = -1                                   # seven surfaces without any peak
    want = np.array([caf.multi_surface_reduce(shards[:, b]) for b in range(count)], dtype=caf.Stream.PEAK_DTYPE)
    dev_recs = [torch.from_numpy(shards[g].view(np.float64).reshape(count, 4).copy()).cuda() for g in range(G)]
    reds = [torch.empty(4 * count, dtype=torch.float64, device="cuda") for _ in range(G)]
    d_freqs = torch.from_numpy(freqs).cuda()
    out = torch.empty((count, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for g in range(G):
        eng.peak_exchange_stage(0, dev_recs[g].data_ptr(), count, reds[g].data_ptr())
    eng.synchronize()
    gmax = torch.stack([r[count:2 * count] for r in reds]).max(dim=0).values          # all-reduce(MAX) over f64
    for r in reds:
        r[count:2 * count] = gmax
    torch.cuda.synchronize()
    for g in range(G):
        eng.peak_exchange_stage(1, dev_recs[g].data_ptr(), count, reds[g].data_ptr())
    eng.synchronize()
    gkey = torch.stack([r.view(torch.int64)[3 * count:] for r in reds]).min(dim=0).values   # all-reduce(MIN) over int64
    for r in reds:
        r.view(torch.int64)[3 * count:] = gkey
    torch.cuda.synchronize()
    eng.peak_exchange_stage(2, 0, count, reds[1].data_ptr(), d_freqs.data_ptr(), F, out.data_ptr())
    eng.synchronize()
    got = out.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    assert got.tobytes() == want.tobytes()
    assert int((want["row"] == -1).sum()) >= 7 and len(np.unique(want["val"])) >= 3
    # argument checks
    from caf_cookoff_amd import _lib
    for bad in (lambda: eng.peak_exchange_stage(3, dev_recs[0].data_ptr(), count, reds[0].data_ptr()),
                lambda: eng.peak_exchange_stage(0, 0, count, reds[0].data_ptr()),
                lambda: eng.peak_exchange_stage(2, 0, count, reds[0].data_ptr(), d_freqs.data_ptr(), F, 0),
                lambda: eng.peak_exchange_stage(0, dev_recs[0].data_ptr(), count, reds[0].data_ptr() + 4)):
        with pytest.raises(caf.CafError) as ei:
            bad()
        assert ei.value.code == _lib.CAF_ERR_BAD_ARG


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


# ------------------------------------------------------ B surfaces per call: caf_multi_surface_run_batch --
def _unsharded_batch(eng, nd, hs, fr, dtype):
    """The same B pairs through the device-pointer API on ONE plan over all rows: (surface, row_idx, row_val, peaks) on the device."""
    import torch
    B, n = nd.shape
    tdt = torch.float64 if dtype == "c128" else torch.float32
    plan = eng.plan(n, fr, FS, dtype=dtype)
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    surf = torch.empty((B, len(fr), 2 * n), dtype=tdt, device="cuda")
    ridx = torch.empty((B, len(fr)), dtype=torch.int64, device="cuda")
    rval = torch.empty((B, len(fr)), dtype=tdt, device="cuda")
    peak = torch.empty((B, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
    eng.synchronize()
    plan.close()
    return surf, ridx, rval, peak


def _check_batch_against_unsharded(ms, devices, ref, ridx, rval, peaks, fr, n, dtype):
    """slabs, row records and peaks of a batch call BIT-EQUAL to the unsharded caf_surface_dev batch"""
    import torch
    import caf_cookoff_amd as caf
    surf0, ridx0, rval0, peak0 = ref
    B = surf0.shape[0]
    ts = "<f8" if dtype == "c128" else "<f4"
    for w in range(len(devices)):
        _, lo, hi, _ = ms.worker_info(w)
        res = ms.batch_results(w)
        assert res["batch"] == B
        if hi > lo:
            slab = _dev_view(res["slab"], (B, hi - lo, 2 * n), ts)
            assert torch.equal(slab, surf0[:, lo:hi, :]), f"worker {w}: slab differs from rows [{lo},{hi}) of the unsharded batch"
            assert torch.equal(_dev_view(res["row_idx"], (B, hi - lo), "<i8"), ridx0[:, lo:hi])
            assert torch.equal(_dev_view(res["row_val"], (B, hi - lo), ts), rval0[:, lo:hi])
    assert np.array_equal(ridx.astype(np.int64), ridx0.cpu().numpy()) and np.array_equal(rval, rval0.cpu().numpy())
    pk0 = peak0.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    for f in ("val", "freq", "idx", "row"):
        assert np.array_equal(peaks[f], pk0[f]), f
    assert np.array_equal(fr[pk0["row"]], pk0["freq"])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["2workers", "3workers"])
@pytest.mark.parametrize("B", [7, 256])
def test_multi_batch_bit_equal_to_unsharded_batch(B, devices, eng, oracle):
    """BASELINE configs[1] through the batched row-shard call: B pairs x 400 rows x 8192 lags complex128 over two / three
    workers on this GPU -- every worker's slab [B][rows][2n], the row records and the B joined peaks are bit-equal to the
    unsharded caf_surface_dev batch; the planted peaks are found; re-running the resident pairs gives the same bits."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    fr = caf.bench_shifts()
    nd, hs, lags, fos = make_batch(B, 4096, FS, seed0=7100)
    ref = _unsharded_batch(eng, nd, hs, fr, "c128")
    ms = caf.MultiSurface(devices, 4096, fr, FS, surface_on_device=True)
    ridx, rval, peaks = ms.run_batch(nd, hs)
    _check_batch_against_unsharded(ms, devices, ref, ridx, rval, peaks, fr, 4096, "c128")
    assert np.array_equal(peaks["idx"], np.asarray(lags))
    assert np.all(np.abs(peaks["freq"] - np.asarray(fos)) <= 0.5 + 1e-9)      # (bench.py's own gate on the 0.5 Hz grid)
    r2, v2, p2 = ms.run_batch(batch=B)                       # needles = haystacks = NULL: the pairs already on the devices
    assert np.array_equal(r2, ridx) and np.array_equal(v2, rval) and p2.tobytes() == peaks.tobytes()
    _, _, p3 = ms.run_batch(batch=B, want_rows=False)        # row records stay on the devices
    assert p3.tobytes() == peaks.tobytes()
    with pytest.raises(caf.CafError) as ei:
        ms.run_batch(batch=B + 1)
    assert ei.value.code == caf._lib.CAF_ERR_STATE
    ms.close()


def test_multi_batch_kat_pairs_vs_oracle(eng, oracle, coracle):
    """The reference's ten s0/s1 pairs as ONE batch on the bench's 400-row grid, three workers: every peak, every row record and
    two full slabs against the oracle (argmax exact, values within 1e-6 of the surface maximum)."""
    import caf_cookoff_amd as caf
    fr = caf.bench_shifts()
    pairs = [_pair(oracle, k) for k in range(10)]
    nd, hs = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ms = caf.MultiSurface([0, 0, 0], 4096, fr, FS, surface_on_device=True)
    ridx, rval, peaks = ms.run_batch(nd, hs)
    for k in range(10):
        osurf, oidx, oval = coracle.caf_surface(nd[k], hs[k], fr, FS, want_surface=True, hoist=True, nthreads=4)
        assert np.array_equal(ridx[k], oidx), f"pair {k}"
        assert np.max(np.abs(rval[k] - oval)) <= TOL64 * osurf.max()
        assert (float(peaks[k]["freq"]), int(peaks[k]["idx"])) == coracle.find_peak(fr, oidx, oval), f"pair {k}"
        if k in (0, 4):
            for w in range(3):
                _, lo, hi, _ = ms.worker_info(w)
                slab = _dev_view(ms.batch_results(w)["slab"], (10, hi - lo, 8192), "<f8")[k].cpu().numpy()
                assert np.max(np.abs(slab - osurf[lo:hi])) <= TOL64 * osurf.max()
    assert (float(peaks[0]["freq"]), int(peaks[0]["idx"])) == (69.0, 202)      # the bench configuration's answer
    ms.close()


def test_multi_batch_ties_no_peak_and_ragged_shards(eng):
    """find_peak over the joined rows (mod.rs:31-42) per surface of a batch: equal maxima in different shards -> the lowest
    global row; an all-zero pair -> (0.0, 0), row -1; more workers than rows (empty shards) and no surface storage."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(11)
    x, y = _planted(rng, 4096, FS, 25.0, 100)
    x2, y2 = _planted(rng, 4096, FS, -40.0, 33)
    z = np.zeros(4096, dtype=np.complex128)
    fr = np.array([25.0, -40.0, 10.0, 25.0, 3.0, 25.0, -40.0])     # rows 0, 3, 5 identical; rows 1, 6 identical
    ms = caf.MultiSurface([0, 0, 0], 4096, fr, FS)                   # shards [0,2) [2,4) [4,7); no slabs (flag not set)
    ridx, rval, peaks = ms.run_batch(np.stack([x, z, x2]), np.stack([y, z, y2]))
    assert rval[0, 0] == rval[0, 3] == rval[0, 5] and int(peaks[0]["row"]) == 0 and int(peaks[0]["idx"]) == 100
    assert (peaks[1]["val"], peaks[1]["freq"], int(peaks[1]["idx"]), int(peaks[1]["row"])) == (0.0, 0.0, 0, -1) and not rval[1].any()
    assert rval[2, 1] == rval[2, 6] and int(peaks[2]["row"]) == 1 and peaks[2]["freq"] == -40.0 and int(peaks[2]["idx"]) == 33
    assert ms.batch_results(0)["slab"] == 0
    sp = [ms.batch_results(w)["shard_peaks"] for w in range(3)]
    assert [int(s[0]["row"]) for s in sp] == [0, 3, 5]                      # one candidate per shard, the join picks the lowest row
    rows2 = [int(s[2]["row"]) for s in sp]
    assert rows2[0] == 1 and rows2[2] == 6 and rows2[1] in (2, 3)            # the middle shard holds no -40 Hz row: a lower peak of its own
    assert [int(s[1]["row"]) for s in sp] == [-1, -1, -1]                    # the all-zero pair: no shard has a peak
    ms.close()
    ms = caf.MultiSurface([0] * 5, 64, np.array([1.0, 2.0]), FS, surface_on_device=True)   # five workers, two rows
    a = rng.standard_normal((4, 64)) + 1j * rng.standard_normal((4, 64))
    ridx, rval, peaks = ms.run_batch(a, np.roll(a, 3, axis=1))
    single = caf.MultiSurface([0], 64, np.array([1.0, 2.0]), FS)
    for b in range(4):
        _, ri, rv, pk = single.run(a[b], np.roll(a[b], 3), want_surface=False)
        assert np.array_equal(ridx[b], ri) and np.array_equal(rval[b], rv) and peaks[b].tobytes() == pk.tobytes()
    single.close()
    ms.close()


@pytest.mark.parametrize("dtype,n,nfreq,B", [("c64", 4096, 400, 9), ("c64", 2048, 11, 5), ("c128", 1024, 37, 3), ("c128", 64, 23, 6),
                                             ("c64", 32768, 6, 2)])
def test_multi_batch_every_path_vs_unsharded_and_oracle(dtype, n, nfreq, B, eng, oracle):
    """Every kernel family behind the batch call (tuned n = 4096 complex64, chain R = 2 / 4, lane-group rows), three workers:
    bit-equal to the unsharded batch, peaks equal to the oracle's."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(n + nfreq)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    fr = np.linspace(-30.0, 30.0, nfreq, endpoint=False)
    prs = [_planted(rng, n, FS, float(fr[(3 * b + 1) % nfreq]), 5 + 7 * b, cdt) for b in range(B)]
    nd, hs = np.stack([p[0] for p in prs]), np.stack([p[1] for p in prs])
    ref = _unsharded_batch(eng, nd, hs, fr, dtype)
    ms = caf.MultiSurface([0, 0, 0], n, fr, FS, dtype=dtype, surface_on_device=True)
    ridx, rval, peaks = ms.run_batch(nd, hs)
    _check_batch_against_unsharded(ms, [0, 0, 0], ref, ridx, rval, peaks, fr, n, dtype)
    for b in range(B):
        _, oidx, oval = oracle.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr, FS, want_surface=False)
        assert (float(peaks[b]["freq"]), int(peaks[b]["idx"])) == oracle.np_find_peak(fr, oidx, oval)
        assert int(peaks[b]["idx"]) == 5 + 7 * b and int(peaks[b]["row"]) == (3 * b + 1) % nfreq
    ms.close()


@pytest.mark.parametrize("B", [7, 256])
def test_multi_batch_rccl_join_single_rank(B, eng):
    """CAF_MULTI_REDUCE_RCCL for a batch with ONE rank on this one-GPU box: ONE grouped ncclAllReduce(max) over the B shard
    values + ONE ncclAllReduce(min) over the B keys per call; the B peaks equal the host join's bit for bit; a single-surface
    run on the same object still works afterwards (its reduction buffers grew, they did not move under it)."""
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    fr = caf.bench_shifts()
    nd, hs, lags, _ = make_batch(B, 4096, FS, seed0=7300)
    nd[1], hs[1] = 0, 0                                                   # one pair without a peak
    host = caf.MultiSurface([0], 4096, fr, FS)
    _, _, want = host.run_batch(nd, hs, want_rows=False)
    host.close()
    ms = caf.MultiSurface([0], 4096, fr, FS, rccl=True)
    for _ in range(2):
        ridx, rval, peaks = ms.run_batch(nd, hs)
        assert peaks.tobytes() == want.tobytes()
    assert int(peaks[1]["row"]) == -1 and peaks[1]["val"] == 0.0 and np.array_equal(np.delete(peaks["idx"], 1), np.delete(np.asarray(lags), 1))
    stats, _ = ms.run_stats()
    assert stats["reduce_s"] > 0
    _, _, _, pk = ms.run(nd[0], hs[0], want_surface=False)
    assert pk.tobytes() == want[0].tobytes()
    ms.close()


def test_multi_surface_on_device_default_and_device_kept(eng):
    """An object created with surface_on_device=True runs with its default arguments (the rows stay in the workers' HBM) and
    refuses a host surface with a Python error; the calls leave the caller's current device as they found it."""
    import torch
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(2)
    x, y = _planted(rng, 1024, FS, 2.0, 9)
    fr = np.array([0.0, 2.0, 4.0])
    before = torch.cuda.current_device()
    ms = caf.MultiSurface([0, 0], 1024, fr, FS, surface_on_device=True)
    surf, ridx, rval, pk = ms.run(x, y)                    # default: no host surface
    assert surf is None and int(pk["row"]) == 1 and int(pk["idx"]) == 9 and ms.slab_ptr(0) != 0
    with pytest.raises(ValueError):
        ms.run(x, y, want_surface=True)
    ms.close()
    assert torch.cuda.current_device() == before


def test_multi_batch_kat_pairs_c64(eng, oracle, coracle):
    """The reference's ten pairs as one complex64 batch on the bench grid (BASELINE configs[2] behind the batch call), two
    workers: tau exact for every pair, the row the f64 oracle picks or its neighbour (chirp_0's two best rows differ by 6.8e-6
    of the maximum, SURVEY.md section 7: below complex64's reach), peak values within 1e-3 of the maximum."""
    import caf_cookoff_amd as caf
    fr = caf.bench_shifts()
    pairs = [_pair(oracle, k) for k in range(10)]
    nd, hs = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    ms = caf.MultiSurface([0, 0], 4096, fr, FS, dtype="c64", surface_on_device=True)
    ridx, rval, peaks = ms.run_batch(nd.astype(np.complex64), hs.astype(np.complex64))
    exact = 0
    for k in range(10):
        _, oidx, oval = coracle.caf_surface(nd[k], hs[k], fr, FS, want_surface=False, hoist=True, nthreads=4)
        of, oi = coracle.find_peak(fr, oidx, oval)
        assert int(peaks[k]["idx"]) == oi, f"pair {k}"
        assert abs(float(peaks[k]["freq"]) - of) <= 0.5 + 1e-9, f"pair {k}"
        assert abs(float(peaks[k]["val"]) - oval.max()) <= TOL32 * oval.max()
        assert np.max(np.abs(rval[k].astype(np.float64) - oval)) <= TOL32 * oval.max()
        exact += float(peaks[k]["freq"]) == of
    assert exact >= 9          # (all ten on every box so far; chirp_0 is the one that may move to the neighbouring row)
    ms.close()
