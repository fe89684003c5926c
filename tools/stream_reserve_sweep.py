#!/usr/bin/env python3
"""How many workgroup slots should a streaming slot's persistent row launch leave free so that the next slot's staging +
spectrum launch overlaps it?  Measurement library only (CAF_STREAM_RESERVE overrides the plan's reserve at capture time).
1000-surface passes of the fixed streaming form (eight surfaces per replay, four slots), median of 7, A/B/A/B over the values.
usage: stream_reserve_sweep.py [values...]"""
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

BATCH = int(os.environ.get("SWEEP_BATCH", "8"))
NSLOTS = int(os.environ.get("SWEEP_NSLOTS", "4"))
DTYPES = os.environ.get("SWEEP_DTYPES", "c128,c64").split(",")
vals = [int(v) for v in sys.argv[1:]] or [0, 8, 16, 32, 64, 0, 16, 32]
eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
fr = caf.bench_shifts()
nd, hs, lags, _ = make_batch(64, 4096, 48000, seed0=5000)
total = 2048
a, b = np.tile(nd, (32, 1))[:total], np.tile(hs, (32, 1))[:total]
want = np.tile(np.asarray(lags), 32)[:total]
for dtype in DTYPES:
    cdt = np.complex128 if dtype == "c128" else np.complex64
    aa, bb = a.astype(cdt), b.astype(cdt)
    for r in vals:
        os.environ["CAF_STREAM_RESERVE"] = str(r)
        plan = eng.plan(4096, fr, 48000, dtype=dtype)
        st = caf.Stream(plan, batch=BATCH, nslots=NSLOTS, want_surface=True)
        st.run(aa, bb)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            pk, _, _ = st.run(aa, bb)
            ts.append(time.perf_counter() - t0)
        ok = int(np.sum(pk["idx"] == want))
        print(f"{dtype} batch {BATCH} x {NSLOTS} slots reserve {r:3d}: median {total / statistics.median(ts):8.0f} surfaces/s  (min {total / max(ts):.0f} max {total / min(ts):.0f})  tau ok {ok}/{total}", flush=True)
        st.close()
        plan.close()
