#!/usr/bin/env python3
"""Run the same n = 32768 complex64 surface twice, and as 512-row shards: report bit differences."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_pair  # noqa: E402

n, F = 32768, 1024
fr = np.arange(F) * 0.05 - 25.6
s0, s1, lag, fo = make_pair(n=n, seed=3, lag=201, foffset=float(fr[600]), dtype=np.complex64)
nd, hs = torch.from_numpy(s0[None]).cuda(), torch.from_numpy(s1[None]).cuda()
eng = caf.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)


def run(lo, hi):
    plan = eng.plan(n, fr, 48000, dtype="c64", row_begin=lo, row_end=hi)
    rows = hi - lo
    surf = torch.full((1, rows, 2 * n), -1.0, dtype=torch.float32, device="cuda")
    ridx = torch.zeros((1, rows), dtype=torch.int64, device="cuda")
    rval = torch.zeros((1, rows), dtype=torch.float32, device="cuda")
    peak = torch.zeros((1, 4), dtype=torch.float64, device="cuda")
    plan.surface_dev(nd.data_ptr(), hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    plan.close()
    return surf[0]


a = run(0, F)
b = run(0, F)
d = (a != b)
print("full vs full: differing elements", int(d.sum()), "rows affected", int(d.any(dim=1).sum()),
      "max abs diff", float((a - b).abs().max()), "surface max", float(a.max()))
c = run(0, 512)
d = (a[:512] != c)
rows = torch.nonzero(d.any(dim=1)).flatten()
print("full vs shard[0:512]: differing elements", int(d.sum()), "rows affected", rows.numel(), rows[:20].tolist(),
      "max abs diff", float((a[:512] - c).abs().max()))
if rows.numel():
    r = int(rows[0])
    cols = torch.nonzero(d[r]).flatten()
    print("row", r, "differing lags", cols.numel(), cols[:16].tolist(), "...", cols[-4:].tolist())
