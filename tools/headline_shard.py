#!/usr/bin/env python3
"""What one rank of an N-GPU headline run computes (bench.py --gpus N: rows [r*400/N, (r+1)*400/N) of 256*N surfaces
per step), timed on ONE GPU next to the N = 1 step: the scaling efficiency the row-shard decomposition can reach
before any collective.  usage: headline_shard.py [N ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

eng = caf.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
fr = caf.bench_shifts()
nd16, hs16, lags, _ = make_batch(16, 4096, 48000, seed0=77)
base = None
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    batch = 256 * N
    lo, hi = caf.shard_range(400, 0, N)
    rows = hi - lo
    reps = batch // 16
    nd = torch.from_numpy(np.tile(nd16, (reps, 1))).cuda()
    hs = torch.from_numpy(np.tile(hs16, (reps, 1))).cuda()
    plan = eng.plan(4096, fr, 48000, row_begin=lo, row_end=hi)
    surf = torch.empty((batch, rows, 8192), dtype=torch.float64, device="cuda")
    ridx = torch.empty((batch, rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((batch, rows), dtype=torch.float64, device="cuda")
    peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
    args = (nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
    for _ in range(3):
        plan.surface_dev(*args)
    torch.cuda.synchronize()
    K = 30
    plan.timing_begin()
    t0 = time.perf_counter()
    for _ in range(K):
        plan.surface_dev(*args)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    ms, nl = plan.timing_end()
    if base is None:
        base = el
    print(f"N={N}: {batch} surfaces x {rows} rows per step: {el * 1e3:.3f} ms per step (row kernel {ms / nl:.3f} ms), "
          f"{batch / el * 1e-3 / N:.1f} k surfaces/s per GPU-equivalent = {base / el:.3f} of the N = 1 step")
    plan.close()
    del surf, ridx, rval, peak, nd, hs
    torch.cuda.empty_cache()
