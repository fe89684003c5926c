#!/usr/bin/env python3
"""BASELINE configs[3] shape (4096 x 65536 complex64) on ONE GPU's row shard through the
plan.s kernel path (tiled65536): 512 rows (= the 1/8 shard an 8-GPU job gives each rank)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_pair  # noqa: E402

n, F, G = 32768, 4096, 8
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fr = np.arange(F) * 0.05 - 102.4 + 0.0        # 0.05 Hz grid, truth on-grid
s0, s1, lag, fo = make_pair(n=n, seed=3, lag=201, foffset=float(fr[1800]), dtype=np.complex64)
eng = caf.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
lo, hi = caf.shard_range(F, rank, G)
plan = eng.plan(n, fr, 48000, dtype="c64", row_begin=lo, row_end=hi)
rows = hi - lo
nd, hs = torch.from_numpy(s0[None]).cuda(), torch.from_numpy(s1[None]).cuda()
surf = torch.empty((1, rows, 2 * n), dtype=torch.float32, device="cuda")
ridx = torch.empty((1, rows), dtype=torch.int64, device="cuda")
rval = torch.empty((1, rows), dtype=torch.float32, device="cuda")
peak = torch.empty((1, 4), dtype=torch.float64, device="cuda")
args = (nd.data_ptr(), hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
plan.surface_dev(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    plan.surface_dev(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
out_bytes = rows * 2 * n * 4
print(f"path={plan.path} shard rows [{lo},{hi}) of {F} x {2 * n} c64: {dt * 1e3:.1f} ms per shard "
      f"-> {1 / dt:.2f} shard-surfaces/s, output {out_bytes / 1e6:.0f} MB ({out_bytes / dt / 1e9:.0f} GB/s algorithmic); "
      f"peak row {int(pk['row'])} ({pk['freq']:.2f} Hz) idx {int(pk['idx'])}; truth row 1800 ({fo:.2f} Hz) lag {lag}")
