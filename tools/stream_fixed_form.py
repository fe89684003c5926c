#!/usr/bin/env python3
"""The fixed streaming form of extra.configs4_stream (eight surfaces per graph replay, four slots) through
caf_stream_run, by itself -- the command behind profiles/r03_stream/fixed_form_kernel_stats.csv
(rocprofv3 --kernel-trace --stats).  usage: stream_fixed_form.py [surfaces per run] [runs]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(64, 4096, 48000, seed0=5000)
reps = (count + 63) // 64
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
st = caf.Stream(plan, batch=8, nslots=4, want_surface=True)
st.run(nd[:64], hs[:64])
rates = []
for _ in range(runs):
    t0 = time.perf_counter()
    peaks, _, _ = st.run(nd, hs)
    rates.append(count / (time.perf_counter() - t0))
    assert int(np.sum(peaks["idx"] == lags)) == count
rs = st.run_stats()
fill, launch, wait, collect = rs["fill_s"], rs["launch_s"], rs["wait_s"], rs["collect_s"]
print(f"batched8_4slots: median {np.median(rates):.0f} surfaces/s over {runs} runs of {count} (min {min(rates):.0f}, max {max(rates):.0f}); "
      f"host thread per surface of the last run: fill {fill / count * 1e6:.2f} us, launch {launch / count * 1e6:.2f}, "
      f"wait {wait / count * 1e6:.2f}, collect {collect / count * 1e6:.2f}")
st.close()
plan.close()
