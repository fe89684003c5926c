"""CPU tests of the multi-device rules the C ABI exports without needing a GPU: the contiguous row-shard rule of
caf_multi_surface_*, the find_peak join over shard records (caf_multi_surface_reduce: largest value, then lowest global
row == the reference's first-strictly-greater scan over the joined rows, caf_rust/src/caf/mod.rs:31-42), and their
agreement with the torch.distributed form (caf_cookoff_amd.dist) and with the oracle's find_peak."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def caf():
    import __graft_entry__ as g
    g.build()
    import caf_cookoff_amd
    return caf_cookoff_amd


def test_shard_rule_is_contiguous_and_complete(caf):
    for nfreq in (0, 1, 2, 3, 7, 400, 401, 4096, 65537):
        for workers in (1, 2, 3, 5, 8, 64):
            edges = [caf.multi_surface_shard(nfreq, workers, w) for w in range(workers)]
            assert edges == [caf.shard_range(nfreq, w, workers) for w in range(workers)]   # the torch.distributed path's rule
            assert edges[0][0] == 0 and edges[-1][1] == nfreq
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))                      # contiguous, in row order
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
    assert caf.multi_surface_shard(4096, 8, 3) == (1536, 2048)                              # BASELINE configs[3]: 512 rows per GPU
    with pytest.raises(caf.CafError):
        caf.multi_surface_shard(10, 2, 2)
    with pytest.raises(caf.CafError):
        caf.multi_surface_shard(10, 0, 0)


def _rec(caf, val, freq, idx, row):
    r = np.zeros(1, dtype=caf.Stream.PEAK_DTYPE)
    r[0] = (val, freq, idx, row)
    return r[0]


def test_reduce_rule_ties_and_no_peak(caf):
    P = caf.Stream.PEAK_DTYPE
    shards = np.array([(7.5, 10.0, 11, 5), (7.5, 20.0, 21, 150), (7.5, 30.0, 31, 300)], dtype=P)
    best = caf.multi_surface_reduce(shards)
    assert (best["val"], best["freq"], int(best["idx"]), int(best["row"])) == (7.5, 10.0, 11, 5)      # lowest global row among equal maxima
    best = caf.multi_surface_reduce(shards[::-1].copy())                                              # ... whatever the order of the records
    assert int(best["row"]) == 5
    none = np.array([(0.0, 0.0, 0, -1)] * 3, dtype=P)
    best = caf.multi_surface_reduce(none)
    assert (best["val"], best["freq"], int(best["idx"]), int(best["row"])) == (0.0, 0.0, 0, -1)       # mod.rs:32-35 initial maximum
    mixed = np.array([(1.0, 1.0, 12, 2), (0.0, 0.0, 0, -1), (9.0, 3.0, 22, 301)], dtype=P)
    best = caf.multi_surface_reduce(mixed)
    assert (best["val"], int(best["idx"]), int(best["row"])) == (9.0, 22, 301)
    best = caf.multi_surface_reduce(np.zeros(0, dtype=P))
    assert int(best["row"]) == -1
    # a record without a row never wins, whatever its value field says
    odd = np.array([(5.0, 0.0, 0, -1), (2.0, 4.0, 9, 7)], dtype=P)
    assert int(caf.multi_surface_reduce(odd)["row"]) == 7


def test_reduce_rule_equals_reference_scan_and_dist_path(caf, oracle):
    """Random row peaks (with planted exact ties) cut into contiguous shards: the join of the shard records equals
    find_peak over all rows (the oracle's restatement of mod.rs:31-42) and the torch.distributed reduction's key rule."""
    import torch
    from caf_cookoff_amd.dist import decode_key, encode_key
    rng = np.random.default_rng(5)
    P = caf.Stream.PEAK_DTYPE
    for trial in range(200):
        F = int(rng.integers(1, 60))
        G = int(rng.integers(1, 9))
        fr = rng.uniform(-100, 100, F)
        rval = rng.choice([0.0, 1.0, 2.5, 2.5, 7.0], F) if trial % 2 else rng.uniform(0, 5, F)
        ridx = rng.integers(0, 8192, F).astype(np.uint64)
        shards = np.zeros(G, dtype=P)
        for w in range(G):
            lo, hi = caf.multi_surface_shard(F, G, w)
            bf, bi = oracle.np_find_peak(fr[lo:hi], ridx[lo:hi], rval[lo:hi])       # the shard's own find_peak
            row = -1
            if hi > lo and rval[lo:hi].max() > 0:
                row = lo + int(np.argmax(rval[lo:hi]))
            shards[w] = (rval[row] if row >= 0 else 0.0, bf, bi, row)
        best = caf.multi_surface_reduce(shards)
        want_f, want_i = oracle.np_find_peak(fr, ridx, rval)
        assert (best["freq"], int(best["idx"])) == (want_f, want_i)
        # the RCCL form's arithmetic: max of the values, then min of (row << 32 | idx) among the holders
        vals = np.where(shards["row"] >= 0, shards["val"], 0.0)
        gmax = vals.max()
        keys = [int(encode_key(torch.tensor(int(s["row"])), torch.tensor(int(s["idx"])))) for s, v in zip(shards, vals)
                if s["row"] >= 0 and v == gmax and gmax > 0]
        if keys:
            row, idx = decode_key(torch.tensor(min(keys)))
            assert (int(row), int(idx)) == (int(best["row"]), int(best["idx"]))
        else:
            assert int(best["row"]) == -1


def test_multi_objects_fail_loudly_without_a_gpu(caf):
    """No CPU fallback behind the multi-device entry points either."""
    lib = caf.load()
    if lib.caf_device_count() > 0:
        pytest.skip("a GPU is visible")
    fr = np.array([0.0, 1.0])
    with pytest.raises(caf.CafError) as ei:
        caf.MultiSurface([0, 1], 4096, fr, 48000)
    assert ei.value.code == 5
    with pytest.raises(caf.CafError) as ei:
        caf.MultiStream([0], 4096, fr, 48000, want_surface=True)
    assert ei.value.code == 5


def test_guard_band_switch_without_a_gpu(caf):
    """The debug switch itself is host state: setting and clearing it needs no device, and with nothing allocated the
    checker reports zero allocations."""
    caf.debug_guard_bands(4096)
    caf.debug_guard_bands(0)
    if caf.load().caf_device_count() == 0:
        assert caf.debug_check_guards() == (0, 0)
    with pytest.raises(caf.CafError):
        caf.debug_guard_bands(1 << 30)
