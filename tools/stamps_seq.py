#!/usr/bin/env python3
"""Stamped DIAG build of the sequential-chain row kernel: where one row spends its cycles."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_batch  # noqa: E402

NST = 28
SEG = ["mixer(+a wait)", "DFT1+twA+ex1 W", "barrier ex1", "ex1 R+DFT2+twB+ex2 W", "ex2 R+DFT3", "H mul+DFT4+ex3 W",
       "ex3 R+twB+DFT5+ex4 W", "barrier ex4", "ex4 R+barrier", "twA+DFT6"]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)  # the stamped builds exist only in the measurement library
eng.set_stream(torch.cuda.current_stream().cuda_stream)
fr = caf.bench_shifts()
nd_h, hs_h, _, _ = make_batch(batch, 4096, 48000, seed0=1000)
nd, hs = torch.from_numpy(nd_h).cuda(), torch.from_numpy(hs_h).cuda()
plan = eng.plan(4096, fr, 48000)
surf = torch.empty((batch, 400, 8192), dtype=torch.float64, device="cuda")
ridx = torch.empty((batch, 400), dtype=torch.int64, device="cuda")
rval = torch.empty((batch, 400), dtype=torch.float64, device="cuda")
peak = torch.empty((batch, 4), dtype=torch.float64, device="cuda")
dbg_all = torch.zeros(32 * 4 * NST + 4 * 8 * 256, dtype=torch.int64, device="cuda")  # stamps + 4 words per workgroup
dbg = dbg_all[:32 * 4 * NST].view(32, 4, NST)
lib = eng.lib
lib.caf_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
args = (nd.data_ptr(), hs.data_ptr(), batch, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(), peak.data_ptr())
plan.surface_dev(*args)
torch.cuda.synchronize()
assert lib.caf_debug_set_stamps(plan._h, ctypes.c_void_p(dbg_all.data_ptr()), dbg_all.numel()) == 0, lib.caf_last_error_string()
plan.surface_dev(*args)
torch.cuda.synchronize()
rec = dbg_all[32 * 4 * NST:].view(-1, 4).cpu().numpy()[:512]
t0, t1 = rec[:, 0].astype(np.float64), rec[:, 1].astype(np.float64)
base = t0.min()
dur = t1 - t0
hw = rec[:, 2]
cu = ((hw >> 32) & 15) * 256 + ((hw >> 8) & 255)        # xcc | se,sh,cu
print(f"workgroups: start spread {np.ptp(t0):.0f} ticks, duration min/median/max {dur.min():.0f}/{np.median(dur):.0f}/{dur.max():.0f} "
      f"ticks, last end - first start {t1.max() - base:.0f}; rows per WG {rec[:, 3].min()}..{rec[:, 3].max()}")
ucu, cnt = np.unique(cu, return_counts=True)
print(f"distinct CUs used {len(ucu)}, workgroups per CU histogram {dict(zip(*np.unique(cnt, return_counts=True)))}")
for c in (1, 2, 3):
    sel = np.isin(cu, ucu[cnt == c])
    if sel.any():
        print(f"  CUs holding {c} WG(s): WG duration median {np.median(dur[sel]):.0f} ticks, mean end {np.mean(t1[sel] - base):.0f}")
d = dbg.cpu().numpy()[2:20].astype(np.float64)  # [iter][wave][stamp]
tot = (d[:, :, 22] - d[:, :, 0]).mean()
print(f"rows sampled {d.shape[0]}; mean cycles per row {tot:.0f} (row period {np.diff(d[:, 0, 0]).mean():.0f})")
for ch in (0, 1):
    base = ch * 11
    for i, name in enumerate(SEG):
        seg = (d[:, :, base + i + 1] - d[:, :, base + i]).mean()
        print(f"  chain {'EO'[ch]} {name:24s} {seg:8.0f}  {100 * seg / tot:5.1f}%")
    if ch == 0:
        gap = (d[:, :, 11] - d[:, :, 10]).mean()
        print(f"  chain E->O gap               {gap:8.0f}  {100 * gap / tot:5.1f}%")
ep = (d[:, :, 22] - d[:, :, 21]).mean()
print(f"  epilogue (combine/stores/argmax) {ep:6.0f}  {100 * ep / tot:5.1f}%")
for name, a, b in (("    last stage + |.|^2 + next-row loads", 21, 23), ("    pairing + stores", 23, 24), ("    argmax reduce + publish", 24, 22)):
    seg = (d[:, :, b] - d[:, :, a]).mean()
    print(f"  {name:40s} {seg:6.0f}  {100 * seg / tot:5.1f}%")

# ---- wall time of the stamped build vs the normal build (same data) ----
def timeit(n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        plan.surface_dev(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

t_diag = timeit()
lib.caf_debug_set_stamps(plan._h, None, 0)
t_norm = timeit()
print(f"step time: stamped build {t_diag:.4f} ms, normal build {t_norm:.4f} ms (batch {batch})")
