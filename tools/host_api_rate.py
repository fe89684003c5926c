#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer API (caf_surface_c128): 128 KiB H2D + kernels +
26 MB D2H per call.  Reported in DESIGN.md; never bench.py's `value`."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_pair  # noqa: E402

eng = caf.Engine(0)
fr = caf.bench_shifts()
s0, s1, lag, fo = make_pair(seed=7)
for want in (True, False):
    eng.surface_arrays(s0, s1, fr, 48000, want_surface=want)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        surf, ridx, rval, pk = eng.surface_arrays(s0, s1, fr, 48000, want_surface=want)
    dt = (time.perf_counter() - t0) / n
    print(f"host API, surface copied back={want}: {dt * 1e3:.3f} ms/surface = {1 / dt:.0f} surfaces/s "
          f"(peak {pk.freq} Hz, idx {pk.idx}, truth {fo:.2f} Hz, {lag})")
