#!/usr/bin/env python3
"""Run the stamped DIAG build of the lane-half 512-thread row kernel (k_fused_rows<T, true>,
CAF_ROW_KERNEL=1 of the measurement library) and print where one Doppler row spends its
shader cycles.  Diagnostic only: the
stamps serialise LDS traffic (lgkmcnt(0) at each), so read SHARES, not totals."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

NST = 17
NAMES = ["row start", "mixer (a prefetched)", "DFT16#1 + twA", "ex1 write", "barrier ex1", "ex1 read", "DFT16#2 + twB",
         "ex2 write+read", "DFT16#3", "H load+mul + DFT16#4", "ex3 w+r + twB", "DFT16#5 + ex4 write", "barrier ex4",
         "ex4 read", "barrier next", "twA+DFT16#6+combine+stores", "argmax reduce"]

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "c128"
os.environ["CAF_ROW_KERNEL"] = "1"  # the variant this stamp layout belongs to (read at plan creation)
eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
fr = caf.bench_shifts()
cdt = np.complex128 if dtype == "c128" else np.complex64
rdt = torch.float64 if dtype == "c128" else torch.float32
nd_h, hs_h, _, _ = make_batch(batch, 4096, 48000, seed0=1000, dtype=cdt)
nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
plan = eng.plan(4096, fr, 48000, dtype=dtype)
surf = torch.empty((batch, 400, 8192), dtype=rdt, device="cuda")
ridx = torch.empty((batch, 400), dtype=torch.int64, device="cuda")
rval = torch.empty((batch, 400), dtype=rdt, device="cuda")
peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
dbg = torch.zeros((32, 8, NST), dtype=torch.int64, device="cuda")
lib = eng.lib
lib.caf_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
args = (nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
plan.surface_dev(*args)
torch.cuda.synchronize()
assert lib.caf_debug_set_stamps(plan._h, ctypes.c_void_p(dbg.data_ptr()), dbg.numel()) == 0, lib.caf_last_error_string()
plan.surface_dev(*args)
torch.cuda.synchronize()
lib.caf_debug_set_stamps(plan._h, None, 0)
d = dbg.cpu().numpy()
iters = min(32, (batch * 400 + 255) // 256) - 1
d = d[1:iters]  # skip the first row (cold) and the unwritten tail
seg = np.diff(d, axis=2).astype(np.float64)  # [iter][wave][NST-1]
nxt = (d[1:, :, 0] - d[:-1, :, NST - 1]).astype(np.float64)  # loop back edge
row = (d[1:, :, 0] - d[:-1, :, 0]).astype(np.float64)
print(f"batch {batch} {dtype}: rows sampled {seg.shape[0]}, mean cycles per row (wave 0 | wave 4 | all waves)")
tot = seg.mean(axis=(0, 1)).sum()
for i in range(NST - 1):
    print(f"  {NAMES[i + 1]:28s} {seg[:, 0, i].mean():8.0f} {seg[:, 4, i].mean():8.0f} {seg[:, :, i].mean():8.0f}"
          f"  {100 * seg[:, :, i].mean() / tot:5.1f}%")
print(f"  {'sum of segments':28s} {tot:8.0f};  row period {row.mean():.0f} cycles (loop edge {nxt.mean():.0f})")
