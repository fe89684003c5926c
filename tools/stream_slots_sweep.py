#!/usr/bin/env python3
"""configs[4] at single-surface granularity vs slot count (and vs the HIP runtime's hardware-queue limit: run with
GPU_MAX_HW_QUEUES=8 in the environment to lift the default of 4).  usage: stream_slots_sweep.py [count] [rounds] [slots...]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
slots = [int(a) for a in sys.argv[3:]] or [2, 3, 4, 5, 6, 8]
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=5000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
for s in slots:
    st = caf.Stream(plan, batch=1, nslots=s, want_surface=True)
    st.run(nd[:64], hs[:64])
    r = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        peaks, _, _ = st.run(nd, hs)
        r.append(count / (time.perf_counter() - t0))
    r.sort()
    print(f"slots={s}: median {r[len(r) // 2]:8.0f} surfaces/s [{r[0]:.0f} .. {r[-1]:.0f}], tau ok {int(np.sum(peaks['idx'] == lags))}/{count}", flush=True)
    st.close()
