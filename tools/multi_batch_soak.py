#!/usr/bin/env python3
"""Soak of caf_multi_surface_run_batch: random (n, dtype, freq list, worker count, batch, join form) cases, every worker's slab,
every row record and every joined peak compared BIT FOR BIT with the unsharded caf_surface_dev batch of the same pairs, uploads
(piecewise, on the copy stream) and resident re-runs alike.  Workers share GPU 0 (device ids repeat); the RCCL join runs with one
worker (one rank per GPU).  usage: multi_batch_soak.py [seconds]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402


def dev_view(ptr, shape, typestr):
    class _Dev:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Dev(), device="cuda")


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(505)
eng = caf.Engine(0)
t_end = time.time() + budget
iters = surfaces = 0
while time.time() < t_end:
    dtype = "c128" if rng.random() < 0.5 else "c64"
    cdt, tdt, ts = (np.complex128, torch.float64, "<f8") if dtype == "c128" else (np.complex64, torch.float32, "<f4")
    n = int(rng.choice([64, 512, 1024, 2048, 4096, 4096, 4096]))
    F = int(rng.integers(1, 60)) if n != 4096 or rng.random() < 0.5 else 400
    fr = caf.bench_shifts() if F == 400 else np.sort(rng.uniform(-100.0, 100.0, F))
    G = int(rng.integers(1, 5))
    rccl = G == 1 and rng.random() < 0.5
    B = int(rng.choice([1, 2, 5, 17, 64, 70, 130])) if n * F <= 4096 * 60 else int(rng.choice([1, 3, 64, 96]))
    nd_h, hs_h, lags, fos = make_batch(B, n, 48000, seed0=int(rng.integers(0, 1 << 30)), dtype=cdt)
    plan = eng.plan(n, fr, 48000, dtype=dtype)
    nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
    surf = torch.empty((B, F, 2 * n), dtype=tdt, device="cuda")
    ridx = torch.empty((B, F), dtype=torch.int64, device="cuda")
    rval = torch.empty((B, F), dtype=tdt, device="cuda")
    peak = torch.empty((B, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(nd.data_ptr(), hs.data_ptr(), B, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
    eng.synchronize()
    pk0 = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
    ms = caf.MultiSurface([0] * G, n, fr, 48000, dtype=dtype, rccl=rccl, surface_on_device=True)
    for resident in (False, True, False):
        gi, gv, pk = ms.run_batch(batch=B) if resident else ms.run_batch(nd_h, hs_h)
        ok = pk.tobytes() == pk0.tobytes() and np.array_equal(gi.astype(np.int64), ridx.cpu().numpy()) and np.array_equal(gv, rval.cpu().numpy())
        for w in range(G):
            _, lo, hi, _ = ms.worker_info(w)
            if hi > lo:
                ok = ok and torch.equal(dev_view(ms.batch_results(w)["slab"], (B, hi - lo, 2 * n), ts), surf[:, lo:hi, :])
        if not ok:
            print(f"MISMATCH iter {iters}: n {n} {dtype} F {F} workers {G} B {B} rccl {rccl} resident {resident}")
            sys.exit(1)
    ms.close()
    plan.close()
    iters += 1
    surfaces += 3 * B
print(f"multi batch soak ok: {iters} random (n, dtype, F, workers, B, join) cases, {surfaces} surfaces through run_batch, all bit-equal to the unsharded batch")
