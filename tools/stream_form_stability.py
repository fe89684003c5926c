#!/usr/bin/env python3
"""How stable is a streaming form's rate across caf_stream objects created one after the other in one process (the
runtime maps HIP streams onto a few hardware queues as it sees fit)?  Each form is created `ncreate` times in rotation;
per creation: median of `rounds` runs of `count` surfaces.  usage: stream_form_stability.py [count] [rounds] [ncreate]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ncreate = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = caf.Engine(0)
plan = eng.plan(4096, caf.bench_shifts(), 48000)
nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=5000)
reps = (count + 15) // 16
nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
forms = [(1, 4), (8, 4), (16, 4), (24, 2), (32, 2), (32, 3), (48, 2), (64, 2)]
res = {f: [] for f in forms}
for it in range(ncreate):
    for b, s in forms:
        st = caf.Stream(plan, batch=b, nslots=s, want_surface=True)
        st.run(nd[:128], hs[:128])
        r = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            st.run(nd, hs)
            r.append(count / (time.perf_counter() - t0))
        r.sort()
        res[(b, s)].append(r[len(r) // 2])
        st.close()
for (b, s), v in res.items():
    print(f"batch={b:2d} slots={s}: per-creation medians (k surfaces/s): " + " ".join(f"{x / 1e3:5.1f}" for x in v) +
          f"   min {min(v) / 1e3:.1f}  max {max(v) / 1e3:.1f}", flush=True)
