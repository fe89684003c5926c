#!/usr/bin/env python3
"""The fixed streaming form (eight surfaces per replay, four slots) through the native loop, for rocprofv3 --kernel-trace:
usage stream_batched_min.py [surfaces] [batch] [nslots] [c128|c64] [memcpy]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nslots = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dtype = sys.argv[4] if len(sys.argv) > 4 else "c128"
memcpy = len(sys.argv) > 5 and sys.argv[5] == "memcpy"
cdt = np.complex128 if dtype == "c128" else np.complex64
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000, dtype=dtype)
nd, hs, lags, _ = make_batch(64, 4096, 48000, seed0=5000, dtype=cdt)
reps = (total + 63) // 64
a, b = np.tile(nd, (reps, 1))[:total], np.tile(hs, (reps, 1))[:total]
st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, memcpy_nodes=memcpy)
st.run(a, b)
t0 = time.perf_counter()
pk, _, _ = st.run(a, b)
dt = time.perf_counter() - t0
print(f"batch {batch} x {nslots} slots {dtype}{' memcpy nodes' if memcpy else ''}: {total / dt:.0f} surfaces/s; host thread {st.run_stats()}")
st.close()
