#!/usr/bin/env python3
"""BASELINE configs[4] is 1000 back-to-back surfaces: which replay size suits exactly that count?  (A ragged last replay is padded
with zeros and the pipeline fills and drains once per run, so the best form for 1000 surfaces is not necessarily the best for a
long stream.)  Product library; median of 9 passes over 1000 pairs per form, forms visited twice.
usage: stream_1000.py [c128|c64]"""
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "c128"
cdt = np.complex128 if dtype == "c128" else np.complex64
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000, dtype=dtype)
nd, hs, lags, _ = make_batch(64, 4096, 48000, seed0=5000, dtype=cdt)
total = 1000
a, b = np.tile(nd, (16, 1))[:total], np.tile(hs, (16, 1))[:total]
want = np.tile(np.asarray(lags), 16)[:total]
forms = [(8, 4, False), (20, 2, False), (25, 2, False), (32, 2, False), (40, 2, False), (50, 2, False), (32, 2, True), (40, 2, True), (50, 2, True), (8, 4, True)]
for rnd in range(2):
    for batch, nslots, mc in forms:
        st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, memcpy_nodes=mc)
        st.run(a, b)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            pk, _, _ = st.run(a, b)
            ts.append(time.perf_counter() - t0)
        ok = int(np.sum(pk["idx"] == want))
        print(f"{dtype} {batch:3d} per replay x {nslots} slots{' memcpy nodes' if mc else '             '}: median {total / statistics.median(ts):7.0f} surfaces/s "
              f"(min {total / max(ts):.0f}, max {total / min(ts):.0f}) tau ok {ok}/{total}", flush=True)
        st.close()
