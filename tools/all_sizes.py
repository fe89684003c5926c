#!/usr/bin/env python3
"""Row-kernel rate of EVERY power-of-two n from 1 to 65536 in both dtypes through the device API
(profiles/rNN_all_sizes.txt): one line per (n, dtype) with the kernel path, ms per launch, GB/s of surface
written and rows/s.  Batch and row count are chosen so that a launch writes ~256 MiB of surface."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

eng = caf.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
only = [int(a) for a in sys.argv[1:]]
for dtype in ("c128", "c64"):
    cdt, tdt, rs = (np.complex128, torch.float64, 8) if dtype == "c128" else (np.complex64, torch.float32, 4)
    for lg in range(0, 18 if dtype == "c64" else 17):
        n = 1 << lg
        if only and n not in only:
            continue
        row_bytes = 2 * n * rs
        target = 256 << 20
        F = int(min(4096, max(64, target // row_bytes))) if n < 1024 else (400 if n <= 4096 else 1024 if n < 65536 else 256)
        batch = int(max(1, min(4096, target // (F * row_bytes))))
        fr = np.linspace(-100.0, 100.0, F, endpoint=False)
        nd_h, hs_h, lags, fos = make_batch(min(batch, 16), n, 48000, seed0=77, dtype=cdt)
        reps = (batch + len(lags) - 1) // len(lags)
        nd_h, hs_h = np.tile(nd_h, (reps, 1))[:batch], np.tile(hs_h, (reps, 1))[:batch]
        lags = (list(lags) * reps)[:batch]
        nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
        try:
            plan = eng.plan(n, fr, 48000, dtype=dtype)
        except caf.CafError as e:
            print(f"n={n} {dtype}: {e}")
            continue
        surf = torch.empty((batch, F, 2 * n), dtype=tdt, device="cuda")
        ridx = torch.empty((batch, F), dtype=torch.int64, device="cuda")
        rval = torch.empty((batch, F), dtype=tdt, device="cuda")
        peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
        args = (nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
        for _ in range(2):
            plan.surface_dev(*args)
        torch.cuda.synchronize()
        plan.timing_begin()
        for _ in range(5):
            plan.surface_dev(*args)
        ms, nl = plan.timing_end()
        pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
        # the generator plants lags in [7, 256): below n = 512 the lag is not checked here
        tau = (f"tau ok {sum(int(pk[b]['idx']) == lags[b] for b in range(batch))}/{batch}" if n >= 512 else
               "tau n/a (parity of these sizes: tests/test_gpu_round3.py::test_small_path_vs_oracle)")
        out_bytes = batch * F * row_bytes
        print(f"n={n:6d} {dtype} path={plan.path:9s} {plan.kernel_name:44s} F={F:5d} batch={batch:5d}: {ms / nl:8.4f} ms per launch, "
              f"{out_bytes / (ms / nl) / 1e6:7.0f} GB/s of surface, {batch * F / (ms / nl) * 1e-3:10.2f} M rows/s, {tau}",
              flush=True)
        plan.close()
        del surf, ridx, rval, peak, nd, hs
        torch.cuda.empty_cache()
