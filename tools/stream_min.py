#!/usr/bin/env python3
"""Minimal single-surface streaming loop (for rocprofv3 --kernel-trace): usage stream_min.py [steps] [nslots] [three|auto] [c128|c64]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
nslots = int(sys.argv[2]) if len(sys.argv) > 2 else 2
three = len(sys.argv) > 3 and sys.argv[3] == "three"
dtype = sys.argv[4] if len(sys.argv) > 4 else "c128"
eng = caf.Engine(0)
import numpy as np  # noqa: E402
plan = eng.plan(4096, caf.bench_shifts(), 48000, dtype=dtype)
nd, hs, lags, _ = make_batch(16, 4096, 48000, seed0=5000, dtype=np.complex128 if dtype == "c128" else np.complex64)
st = caf.Stream(plan, batch=1, nslots=nslots, want_surface=True, three_kernels=three)
bufs = [st.buffers(s) for s in range(nslots)]
for rep in range(2):
    infl, ok = [], 0
    t0 = time.perf_counter()
    for step in range(steps):
        slot = step % nslots
        if len(infl) == nslots:
            s0, k0 = infl.pop(0)
            pk, _, _ = st.wait(s0, want_rows=False)
            ok += int(pk[0]["idx"]) == lags[k0]
        bufs[slot][0][0] = nd[step % 16]
        bufs[slot][1][0] = hs[step % 16]
        st.submit(slot)
        infl.append((slot, step % 16))
    for s0, k0 in infl:
        pk, _, _ = st.wait(s0, want_rows=False)
        ok += int(pk[0]["idx"]) == lags[k0]
    dt = time.perf_counter() - t0
print(f"slots={nslots} {'three-kernel' if three else 'one-launch'}: {steps / dt:.0f} surfaces/s, tau ok {ok}/{steps}")
st.close()
