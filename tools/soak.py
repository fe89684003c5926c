#!/usr/bin/env python3
"""Soak test: random batch sizes / row shards / dtypes through the device API, every batched
result compared bit for bit with one-surface calls of the same plan (catches rare races in the
row-ticket hand-out, the LDS exchanges and the streaming slots).  usage: soak.py [seconds]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(2024)
eng = caf.Engine(0)
fr = caf.bench_shifts()
t_end = time.time() + budget
iters = 0
while time.time() < t_end:
    dtype = "c128" if rng.random() < 0.5 else "c64"
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    batch = int(rng.integers(1, 14))
    lo = int(rng.integers(0, 200))
    hi = int(rng.integers(lo + 1, 401))
    nd_h, hs_h, lags, fos = make_batch(batch, 4096, 48000, seed0=int(rng.integers(0, 1 << 30)), dtype=cdt)
    nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
    plan = eng.plan(4096, fr, 48000, dtype=dtype, row_begin=lo, row_end=hi)
    rows = plan.rows
    surf = torch.full((batch, rows, 8192), -1.0, dtype=tdt, device="cuda")
    ridx = torch.empty((batch, rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((batch, rows), dtype=tdt, device="cuda")
    peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        plan.surface_dev(nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                         peak.data_ptr())
    eng.synchronize()
    one_s = torch.empty((1, rows, 8192), dtype=tdt, device="cuda")
    one_i = torch.empty((1, rows), dtype=torch.int64, device="cuda")
    one_v = torch.empty((1, rows), dtype=tdt, device="cuda")
    one_p = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    for b in range(batch):
        plan.surface_dev(nd[b].data_ptr(), hs[b].data_ptr(), 1, one_s.data_ptr(), one_i.data_ptr(), one_v.data_ptr(),
                         one_p.data_ptr())
        eng.synchronize()
        ok = torch.equal(surf[b], one_s[0]) and torch.equal(ridx[b], one_i[0]) and torch.equal(rval[b], one_v[0]) \
            and torch.equal(peak[b], one_p[0])
        if not ok:
            print(f"MISMATCH iter {iters} dtype {dtype} batch {batch} shard [{lo},{hi}) surface {b}")
            sys.exit(1)
        pk = one_p.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
        want_row = int(np.argmin(np.abs(fr - fos[b])))
        if lo <= want_row < hi and dtype == "c128":
            assert int(pk["idx"]) == lags[b], (int(pk["idx"]), lags[b])
    plan.close()
    iters += 1
print(f"soak ok: {iters} random (dtype, batch, shard) cases, all batched results == single-surface results")
