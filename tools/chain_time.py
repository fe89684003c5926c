#!/usr/bin/env python3
"""Time one plan shape through the device API: usage chain_time.py n nfreq dtype [batch] [nosurf]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

n, F, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
nosurf = len(sys.argv) > 5 and sys.argv[5] == "nosurf"
import os
eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH if any(k.startswith(("CAF_CHAIN", "CAF_R32")) for k in os.environ) else None)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
fr = np.linspace(-100.0, 100.0, F, endpoint=False)
cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
nd_h, hs_h, lags, fos = make_batch(batch, n, 48000, seed0=77, dtype=cdt)
nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
plan = eng.plan(n, fr, 48000, dtype=dtype)
surf = None if nosurf else torch.empty((batch, F, 2 * n), dtype=tdt, device="cuda")
ridx = torch.empty((batch, F), dtype=torch.int64, device="cuda")
rval = torch.empty((batch, F), dtype=tdt, device="cuda")
peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
args = (nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr() if surf is not None else None, ridx.data_ptr(),
        rval.data_ptr(), peak.data_ptr())
for _ in range(3):
    plan.surface_dev(*args)
torch.cuda.synchronize()
plan.timing_begin()
reps = 10
for _ in range(reps):
    plan.surface_dev(*args)
ms, nl = plan.timing_end()
pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0]
ok = sum(int(pk[b]["idx"]) == lags[b] for b in range(batch))
rs = 8 if dtype == "c128" else 4
out_bytes = batch * F * 2 * n * rs
print(f"{plan.kernel_name} n={n} F={F} {dtype} batch={batch} surface={'no' if nosurf else 'yes'}: {ms / nl:.4f} ms per launch, "
      f"{out_bytes / (ms / nl) / 1e6:.0f} GB/s of surface, {batch * F / (ms / nl) * 1e-3:.2f} M rows/s, tau ok {ok}/{batch}" + (f" ABL={os.environ['CAF_CHAIN_ABL']} (wrong results, timing only)" if os.environ.get("CAF_CHAIN_ABL") else ""))
