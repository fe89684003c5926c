#!/usr/bin/env python3
"""BASELINE configs[4]: N back-to-back 400x8192 c128 surfaces from HOST memory through the
streaming API (pinned double-buffered H2D, one captured hipGraph per slot); sustained
surfaces/s including H2D of the inputs and D2H of the peaks, surfaces left on the device."""
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
pool_n = 64
nd, hs, lags, fos = make_batch(pool_n, 4096, 48000, seed0=5000)
for batch, nslots in ((1, 2), (1, 4), (4, 2), (16, 2), (32, 2)):
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True)
    bufs = [st.buffers(s) for s in range(nslots)]
    steps = max(nslots + 1, total // batch)

    def fill(slot, step):
        a, b = bufs[slot]
        for j in range(batch):
            k = (step * batch + j) % pool_n
            a[j], b[j] = nd[k], hs[k]

    ok = 0
    for warm in range(2):
        t0 = time.perf_counter()
        inflight = []
        for step in range(steps):
            slot = step % nslots
            if len(inflight) == nslots:
                s0, step0 = inflight.pop(0)
                peaks, _, _ = st.wait(s0, want_rows=False)
                ok += int(peaks[0]["idx"]) == lags[(step0 * batch) % pool_n]
            fill(slot, step)
            st.submit(slot)
            inflight.append((slot, step))
        for s0, step0 in inflight:
            peaks, _, _ = st.wait(s0, want_rows=False)
            ok += int(peaks[0]["idx"]) == lags[(step0 * batch) % pool_n]
        dt = time.perf_counter() - t0
    print(f"stream batch={batch:2d} slots={nslots}: {steps * batch} surfaces in {dt * 1e3:.1f} ms = "
          f"{steps * batch / dt:8.0f} surfaces/s ({dt / steps * 1e6:.1f} us/step), tau correct {ok}/{2 * steps}")
    st.close()
