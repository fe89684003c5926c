#!/usr/bin/env python3
"""BASELINE configs[4] through the native loop (caf_stream_run): surfaces/s for the streaming forms.
usage: stream_native.py [count]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=5000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
for batch, nslots, split, three in ((1, 2, False, False), (1, 3, False, False), (1, 4, False, False), (1, 2, False, True),
                                    (1, 3, False, True), (1, 4, False, True), (4, 2, True, False), (4, 2, True, True),
                                    (4, 2, False, False), (16, 2, False, False)):
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, split=split, three_kernels=three)
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        peaks, _, _ = st.run(nd, hs)
        dt = time.perf_counter() - t0
        best = max(best, count / dt)
    ok = int(np.sum(peaks["idx"] == lags))
    print(f"batch={batch:2d} slots={nslots} {'split  ' if split else 'batched'} {'three-kernel' if three else 'one-launch  '}: "
          f"{best:8.0f} surfaces/s (best of 3), tau ok {ok}/{count}")
    st.close()
