"""world_size-2 (and 3) gloo tests of the multi-GPU decomposition on CPU: row shards +
the MAX / MIN-key reduction reproduce the reference's find_peak (mod.rs:31-42), ties
included.  Per-rank shard results come from the CPU oracle here (test infrastructure);
on the GPU box the same reduction runs over RCCL on caf_surface_dev's peak records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DATA

FS = 48000


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import caf_cookoff_amd as caf
        from caf_cookoff_amd.dist import reduce_global_peak
        from oracle import caf_oracle as O
        if case == "kat":
            co = O.COracle()
            results = []
            for k in (0, 2, 4):  # chirp_2 has the tightest row margin (9.5e-6)
                _, hf, (s, e, st), exp = O.KATS[k]
                nd, hs = O.load_pair(DATA, f"chirp_{k}_raw.c64", hf)
                fr = O.gen_float_shifts(s, e, st)
                lo, hi = caf.shard_range(len(fr), rank, world)
                _, ridx, rval = co.caf_surface(nd, hs, fr[lo:hi], FS, want_surface=False, hoist=True)
                # local find_peak over the shard, reported in GLOBAL row positions
                best, brow, bidx = 0.0, -1, 0
                for r in range(hi - lo):
                    if rval[r] > best:
                        best, brow, bidx = float(rval[r]), lo + r, int(ridx[r])
                results.append((best, brow, bidx, fr, exp))
            val = torch.tensor([r[0] for r in results], dtype=torch.float64)
            row = torch.tensor([r[1] for r in results], dtype=torch.int64)
            idx = torch.tensor([r[2] for r in results], dtype=torch.int64)
            gmax, grow, gidx = reduce_global_peak(val, row, idx)
            got = [(float(results[i][3][int(grow[i])]), int(gidx[i])) for i in range(len(results))]
            out_q.put((rank, got, [r[4] for r in results]))
        elif case == "ties":
            # surface 0: equal maxima on both ranks -> the LOWER global row wins
            # surface 1: nobody has a peak            -> (row -1, idx 0, val 0)
            # surface 2: the max lives on the last rank only
            # surface 3: equal value, same row cannot happen across ranks; equal value on rank 0 rows only
            val = torch.tensor([[7.5, 0.0, 1.0, 3.0], [7.5, 0.0, 9.0, 2.0], [7.5, 0.0, 2.0, 3.0]][rank],
                               dtype=torch.float64)
            row = torch.tensor([[5, -1, 2, 4], [150, -1, 199, 120], [300, -1, 301, 333]][rank], dtype=torch.int64)
            idx = torch.tensor([[11, 0, 12, 13], [21, 0, 22, 23], [31, 0, 32, 33]][rank], dtype=torch.int64)
            gmax, grow, gidx = reduce_global_peak(val, row, idx)
            g2, r2, i2 = reduce_global_peak(val, row, idx, method="allreduce")
            assert torch.equal(gmax, g2) and torch.equal(grow, r2) and torch.equal(gidx, i2)
            out_q.put((rank, gmax.tolist(), grow.tolist(), gidx.tolist()))
    finally:
        dist.destroy_process_group()


def _run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_kats_match_reference(world):
    outs = _run(world, "kat")
    for rank, got, exp in outs:
        assert got == [tuple(e) for e in exp], f"rank {rank}: {got} != {exp}"


def test_tie_break_and_no_peak_semantics():
    outs = _run(3, "ties")
    for rank, gmax, grow, gidx in outs:
        assert gmax == [7.5, 0.0, 9.0, 3.0]
        assert grow == [5, -1, 199, 4]       # lowest global row among equal maxima
        assert gidx == [11, 0, 22, 13]


def test_single_process_identity():
    from caf_cookoff_amd.dist import decode_key, encode_key, reduce_global_peak
    v = torch.tensor([1.0, 0.0]); r = torch.tensor([3, -1]); i = torch.tensor([9, 0])
    g, gr, gi = reduce_global_peak(v, r, i)
    assert g.tolist() == [1.0, 0.0] and gr.tolist() == [3, -1] and gi.tolist() == [9, 0]
    k = encode_key(torch.tensor([399, 0]), torch.tensor([8191, 0]))
    rr, ii = decode_key(k)
    assert rr.tolist() == [399, 0] and ii.tolist() == [8191, 0]
