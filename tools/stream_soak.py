#!/usr/bin/env python3
"""Soak of the streaming forms whose cross-workgroup hand-offs are hand-made (k_seq_surface, one and two
nodes): every row peak (index and value bits) and every caf_peak of every surface of every round must equal
round 0's.  usage: stream_soak.py [surfaces per round] [rounds]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = caf.Engine(0)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=7000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
lags = np.tile(np.asarray(lags16), reps)[:count]
bad = 0
for dtype in ("c128", "c64"):
    plan = eng.plan(4096, caf.bench_shifts(), 48000, dtype=dtype)
    ref = None
    for name, kw in (("one node, 2 slots", dict(batch=1, nslots=2, one_kernel=True)),
                     ("one node, 3 slots", dict(batch=1, nslots=3, one_kernel=True)),
                     ("two nodes, 3 slots", dict(batch=1, nslots=3, two_kernels=True)),
                     ("two nodes, 4 chains x 2 slots", dict(batch=4, nslots=2, split=True, two_kernels=True)),
                     ("one node, 4 chains x 2 slots", dict(batch=4, nslots=2, split=True, one_kernel=True))):
        st = caf.Stream(plan, want_surface=True, **kw)
        t0 = time.perf_counter()
        nbad = 0
        for rnd in range(rounds):
            peaks, ridx, rval = st.run(nd, hs, want_rows=True)
            if ref is None:
                ref = (peaks.copy(), ridx.copy(), rval.copy())
                assert np.array_equal(peaks["idx"], lags), "reference round: wrong lags"
            else:
                badmask = np.any(ridx != ref[1], axis=1) | np.any(rval != ref[2], axis=1) | (peaks != ref[0])
                nbad += int(np.sum(badmask))
                nsl = kw["nslots"] * kw["batch"]
                for k in np.nonzero(badmask)[0][:3]:
                    dr = np.nonzero((ridx[k] != ref[1][k]) | (rval[k] != ref[2][k]))[0]
                    prev = k - nsl  # the surface that used the same pinned words one replay earlier
                    stale = (prev >= 0 and len(dr) and np.array_equal(ridx[k][dr], ref[1][prev][dr])
                             and np.array_equal(rval[k][dr], ref[2][prev][dr]))
                    print(f"    round {rnd} surface {k}: {len(dr)} rows differ (first {dr[:6]}), peak record "
                          f"{'differs' if peaks[k] != ref[0][k] else 'equal'}; differing rows equal the previous "
                          f"occupant's values: {bool(stale)}; got {ridx[k][dr[:2]]}/{rval[k][dr[:2]]} want "
                          f"{ref[1][k][dr[:2]]}/{ref[2][k][dr[:2]]}", flush=True)
        dt = time.perf_counter() - t0
        print(f"{dtype} {name:30s}: {rounds * count} surfaces in {dt:.1f} s ({rounds * count / dt:.0f}/s), "
              f"{nbad} surfaces differ from the first round", flush=True)
        bad += nbad
        st.close()
    plan.close()
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
