#!/usr/bin/env python3
"""What does a deadline (caf_multi_surface_set_timeout: polled waits instead of hipStreamSynchronize) cost the in-process
headline?  ONE box, one object per setting, A/B/A/B: blocks of `steps` resident caf_multi_surface_run_batch calls (B = 256,
RCCL join with one rank), with and without a 60 s deadline.  usage: timeout_cost.py [steps] [visits]"""
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
visits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = 256
fr = caf.bench_shifts()
nd, hs, lags, _ = make_batch(B, 4096, 48000, seed0=1000)
ms = caf.MultiSurface([0], 4096, fr, 48000, rccl=True, surface_on_device=True)
ms.run_batch(nd, hs, want_rows=False)
res = {0.0: [], 60.0: []}
for v in range(visits):
    for tmo in (0.0, 60.0):
        ms.set_timeout(tmo)
        for _ in range(5):
            ms.run_batch(batch=B, want_rows=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            _, _, pk = ms.run_batch(batch=B, want_rows=False)
        res[tmo].append((time.perf_counter() - t0) / steps * 1e3)
        assert np.array_equal(pk["idx"], np.asarray(lags))
ms.close()
for tmo, v in res.items():
    print(f"deadline {'none' if not tmo else '%g s' % tmo:>5s}: ms per call " + " ".join(f"{x:.4f}" for x in v)
          + f"   median {statistics.median(v):.4f} -> {B / statistics.median(v) * 1e3:.0f} surfaces/s")
