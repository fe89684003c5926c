#!/usr/bin/env python3
"""caf_multi_surface_run_batch WITH inputs (the PCIe-inclusive form: the reference's bench hands over fresh inputs every
iteration, benches/caf_bench.rs:150-168): into how many pieces should the pipelined upload be cut?  Each piece is its copies
on the worker's copy stream + an event + its own row launch (k_seq_prepare + rows + find_peak), so more pieces hide more of
the copy and pay more launches.  Measurement library (CAF_UPLOAD_PIECES is read at every call), ONE box, A/B/A/B over the
values; per visit the median of `calls` calls; the resident call (no upload) beside it.
usage: upload_pieces.py [batch] [calls] [visits] [pieces ...]       (one table, then frozen)"""
import os
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 15
visits = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pieces = [int(x) for x in sys.argv[4:]] or [1, 2, 4, 8]
fr = caf.bench_shifts()
nd, hs, lags, _ = make_batch(B, 4096, 48000, seed0=1000)
lags = np.asarray(lags)
for pinned in (False, True):
    ms = caf.MultiSurface([0], 4096, fr, 48000, rccl=True, surface_on_device=True, lib=caf.MEASURE_LIB_PATH)
    a, b = nd, hs
    if pinned:   # inputs in memory the object registered: the copies are plain DMA, no staging through the runtime's bounce buffers
        a = ms.host_empty(nd.shape, nd.dtype)
        b = ms.host_empty(hs.shape, hs.dtype)
        a[:], b[:] = nd, hs
    ms.run_batch(a, b, want_rows=False)
    res = {p: [] for p in pieces}
    resident = []
    for v in range(visits):
        for p in pieces:
            os.environ["CAF_UPLOAD_PIECES"] = str(p)
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                _, _, pk = ms.run_batch(a, b, want_rows=False)
                ts.append(time.perf_counter() - t0)
            assert np.array_equal(pk["idx"], lags)
            res[p].append(statistics.median(ts) * 1e3)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            ms.run_batch(batch=B, want_rows=False)
            ts.append(time.perf_counter() - t0)
        resident.append(statistics.median(ts) * 1e3)
    os.environ.pop("CAF_UPLOAD_PIECES", None)
    print(f"B = {B}, inputs in {'registered pinned' if pinned else 'pageable'} memory, median of {calls} calls per visit, {visits} visits (ms per call):")
    r = statistics.median(resident)
    print(f"  resident (no upload): " + " ".join(f"{x:6.3f}" for x in resident) + f"   -> {B / r * 1e3:7.0f} surfaces/s")
    for p in pieces:
        m = statistics.median(res[p])
        print(f"  pieces = {p}:          " + " ".join(f"{x:6.3f}" for x in res[p]) + f"   -> {B / m * 1e3:7.0f} surfaces/s, {m / r:5.3f} x resident")
    del a, b
    ms.close()
