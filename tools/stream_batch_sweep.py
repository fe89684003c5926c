#!/usr/bin/env python3
"""configs[4] with more than one surface per graph replay: surfaces/s of caf_stream_run for (batch, slots) combinations,
batched chains (one k_seq_prepare + one row kernel + one find_peak per replay).  usage: stream_batch_sweep.py [count] [rounds] [batch:slots ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=5000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
forms = [(1, 4), (4, 3), (8, 2), (8, 3), (8, 4), (16, 2), (16, 3), (16, 4), (32, 2), (32, 3), (64, 2), (64, 3)]
if len(sys.argv) > 3:
    forms = [tuple(int(x) for x in a.split(":")) for a in sys.argv[3:]]
streams = [caf.Stream(plan, batch=b, nslots=s, want_surface=True) for b, s in forms]
rates = [[] for _ in forms]
oks = [0] * len(forms)
for st in streams:
    st.run(nd[:256], hs[:256])
for rnd in range(rounds):
    for i, st in enumerate(streams):
        t0 = time.perf_counter()
        peaks, _, _ = st.run(nd, hs)
        rates[i].append(count / (time.perf_counter() - t0))
        oks[i] = int(np.sum(peaks["idx"] == lags))
for (b, s), r, ok, st in zip(forms, rates, oks, streams):
    r.sort()
    hs_ = st.run_stats()
    print(f"batch={b:3d} slots={s}: median {r[len(r) // 2]:8.0f} surfaces/s [{r[0]:.0f} .. {r[-1]:.0f}], tau ok {ok}/{count}; "
          f"host thread per surface: fill {hs_['fill_s'] / count * 1e6:.2f} us, launch {hs_['launch_s'] / count * 1e6:.2f}, "
          f"wait {hs_['wait_s'] / count * 1e6:.2f}, collect {hs_['collect_s'] / count * 1e6:.2f}")
for st in streams:
    st.close()
