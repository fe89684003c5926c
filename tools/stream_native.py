#!/usr/bin/env python3
"""BASELINE configs[4] through the native loop (caf_stream_run): surfaces/s for the streaming forms, every form
measured `rounds` times in alternation (median [min .. max]).  usage: stream_native.py [count] [rounds]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=5000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
NAMES = ("one-launch  ", "three-kernel", "two-kernel  ")
forms = [(1, 2, False, 0), (1, 3, False, 0), (1, 4, False, 0), (1, 2, False, 2), (1, 3, False, 2), (1, 4, False, 2),
         (1, 2, False, 1), (1, 3, False, 1), (4, 2, True, 0), (4, 2, True, 2), (4, 2, True, 1), (4, 2, False, 0), (16, 2, False, 0)]
streams = [caf.Stream(plan, batch=b, nslots=s, want_surface=True, split=sp, three_kernels=k == 1, two_kernels=k == 2)
           for b, s, sp, k in forms]
rates = [[] for _ in forms]
oks = [0] * len(forms)
for st in streams:
    st.run(nd[:64], hs[:64])  # warm-up
for rnd in range(rounds):
    for i, st in enumerate(streams):
        t0 = time.perf_counter()
        peaks, _, _ = st.run(nd, hs)
        rates[i].append(count / (time.perf_counter() - t0))
        oks[i] = int(np.sum(peaks["idx"] == lags))
for (b, s, sp, k), r, ok in zip(forms, rates, oks):
    r.sort()
    print(f"batch={b:2d} slots={s} {'split  ' if sp else 'batched'} {NAMES[k]}: median {r[len(r) // 2]:8.0f} surfaces/s "
          f"[{r[0]:.0f} .. {r[-1]:.0f}], tau ok {ok}/{count}")
for st in streams:
    st.close()
