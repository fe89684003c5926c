#!/usr/bin/env python3
"""Streaming (BASELINE configs[4]) host-side probe: surfaces/s and where a step spends host time,
for batched slots, single-surface slots and CAF_STREAM_SPLIT slots (K single-surface node chains
per graph replay).  usage: stream_probe.py [total_surfaces]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng = caf.Engine(0)
fr = caf.bench_shifts()
plan = eng.plan(4096, fr, 48000)
nd, hs, lags, _ = make_batch(64, 4096, 48000, seed0=5000)
for batch, nslots, split in ((1, 2, False), (1, 3, False), (2, 2, True), (4, 2, True), (4, 3, True), (8, 2, True), (4, 2, False),
                             (16, 2, False)):
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, split=split)
    bufs = [st.buffers(s) for s in range(nslots)]
    steps = max(nslots + 1, total // batch)
    for rep in range(2):
        tf = ts = tw = 0.0
        ok = 0
        t0 = time.perf_counter()
        infl = []
        for step in range(steps):
            slot = step % nslots
            if len(infl) == nslots:
                a = time.perf_counter()
                s0, st0 = infl.pop(0)
                peaks, _, _ = st.wait(s0, want_rows=False)
                tw += time.perf_counter() - a
                ok += all(int(peaks[j]["idx"]) == lags[(st0 * batch + j) % 64] for j in range(batch))
            a = time.perf_counter()
            k0 = (step * batch) % 64
            if k0 + batch <= 64:
                bufs[slot][0][:] = nd[k0:k0 + batch]
                bufs[slot][1][:] = hs[k0:k0 + batch]
            else:
                for j in range(batch):
                    bufs[slot][0][j] = nd[(k0 + j) % 64]
                    bufs[slot][1][j] = hs[(k0 + j) % 64]
            b = time.perf_counter()
            st.submit(slot)
            c = time.perf_counter()
            tf += b - a
            ts += c - b
            infl.append((slot, step))
        for s0, st0 in infl:
            peaks, _, _ = st.wait(s0, want_rows=False)
            ok += all(int(peaks[j]["idx"]) == lags[(st0 * batch + j) % 64] for j in range(batch))
        dt = time.perf_counter() - t0
    print(f"batch={batch:2d} slots={nslots} {'split  ' if split else 'batched'}: {steps * batch / dt:8.0f} surfaces/s; per step: "
          f"fill {tf / steps * 1e6:.1f} us, submit {ts / steps * 1e6:.1f} us, wait {tw / steps * 1e6:.1f} us, "
          f"total {dt / steps * 1e6:.1f} us; tau ok {ok}/{steps}")
    st.close()
