#!/usr/bin/env python3
"""Soak of the host-pointer entry point caf_surface_c128 in its polled form (n = 4096, peaks only: one direct launch
of k_seq_surface, completion read from a pinned sequence word, row peaks written to pinned memory by the kernel):
N calls over a cycle of different inputs; EVERY result (row_idx, row_val, peak) must equal the bits of the first
round.  A result that reached the host after the sequence word would show up here (the streaming path's soak,
tools/stream_sweep.py soak, found exactly that class of bug once).   usage: host_api_soak.py [calls] [dtype]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402


def run(calls=100000, dtype="c128", pool=7, log=True):
    cdt = np.complex128 if dtype == "c128" else np.complex64
    eng = caf.Engine(0)
    fr = caf.bench_shifts()
    nd, hs, lags, _ = make_batch(pool, 4096, 48000, seed0=8800, dtype=cdt)
    ref = []
    for k in range(pool):
        _, ri, rv, pk = eng.surface_arrays(nd[k], hs[k], fr, 48000, want_surface=False, dtype=dtype)
        assert int(pk.idx) == lags[k]
        ref.append((ri.copy(), rv.copy(), (pk.val, pk.freq, pk.idx, pk.row)))
    bad = 0
    t0 = time.perf_counter()
    for i in range(calls):
        k = (i * 3 + (i >> 4)) % pool
        _, ri, rv, pk = eng.surface_arrays(nd[k], hs[k], fr, 48000, want_surface=False, dtype=dtype)
        if not (np.array_equal(ri, ref[k][0]) and np.array_equal(rv, ref[k][1]) and (pk.val, pk.freq, pk.idx, pk.row) == ref[k][2]):
            bad += 1
            if log:
                print(f"call {i} (input {k}): result differs from round 0", flush=True)
    el = time.perf_counter() - t0
    eng.close()
    if log:
        print(f"host API soak {dtype}: {calls} calls, {bad} mismatches, {el / calls * 1e6:.1f} us per call (ctypes + numpy included)")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    dt = sys.argv[2] if len(sys.argv) > 2 else "c128"
    sys.exit(1 if run(n, dt) else 0)
