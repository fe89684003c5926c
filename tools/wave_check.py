#!/usr/bin/env python3
"""tools/wave_check.py -- parity and A/B timing of the one-wave-per-row complex64 kernel (measurement library,
CAF_ROW_KERNEL=4, measure/kernels_wave4096.hpp) against the product kernel k_duo_rows<float> (CAF_ROW_KERNEL=3).
Parity: 400 x 8192 complex64 surfaces of 6 synthetic pairs against the numpy ORACLE (1e-3 of the maximum, row
argmax where clear, the global peak) and determinism (two launches, same bits).  Timing: batch 256, HIP events around
the row kernel, A/B/A/B."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
FS = 48000


def run_variant(variant, nd, hs, fr, batch_time=256, steps=10, surface=True):
    import torch
    import caf_cookoff_amd as caf
    os.environ["CAF_ROW_KERNEL"] = str(variant)
    eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    n, B, F = nd.shape[1], nd.shape[0], len(fr)
    plan = eng.plan(n, fr, FS, dtype="c64")
    dn, dh = torch.from_numpy(nd).cuda(), torch.from_numpy(hs).cuda()
    ds = torch.empty((B, F, 2 * n), dtype=torch.float32, device="cuda")
    di = torch.zeros((B, F), dtype=torch.int64, device="cuda")
    dv = torch.zeros((B, F), dtype=torch.float32, device="cuda")
    dp = torch.zeros((B, 4), dtype=torch.float64, device="cuda")
    outs = []
    for _ in range(2):
        ds.fill_(-1.0)
        plan.surface_dev(dn.data_ptr(), dh.data_ptr(), B, ds.data_ptr(), di.data_ptr(), dv.data_ptr(), dp.data_ptr())
        torch.cuda.synchronize()
        outs.append((ds.cpu().numpy(), di.cpu().numpy(), dv.cpu().numpy(), dp.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[:, 0].copy()))
    name = plan.kernel_name
    # timing at the bench's batch
    reps = (batch_time + B - 1) // B
    tn = torch.from_numpy(np.tile(nd, (reps, 1))[:batch_time]).cuda()
    th = torch.from_numpy(np.tile(hs, (reps, 1))[:batch_time]).cuda()
    ts = torch.empty((batch_time, F, 2 * n), dtype=torch.float32, device="cuda")
    ti = torch.zeros((batch_time, F), dtype=torch.int64, device="cuda")
    tv = torch.zeros((batch_time, F), dtype=torch.float32, device="cuda")
    tp = torch.zeros((batch_time, 4), dtype=torch.float64, device="cuda")
    for _ in range(3):
        plan.surface_dev(tn.data_ptr(), th.data_ptr(), batch_time, ts.data_ptr() if surface else None, ti.data_ptr(), tv.data_ptr(), tp.data_ptr())
    torch.cuda.synchronize()
    plan.timing_begin()
    for _ in range(steps):
        plan.surface_dev(tn.data_ptr(), th.data_ptr(), batch_time, ts.data_ptr() if surface else None, ti.data_ptr(), tv.data_ptr(), tp.data_ptr())
    ms, nl = plan.timing_end()
    plan.close()
    eng.close()
    return outs, name, ms / max(1, nl)


def main():
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    from oracle import caf_oracle as O
    fr = caf.bench_shifts()
    nd, hs, lags, fos = make_batch(6, 4096, FS, seed0=4242, dtype=np.complex64)
    rep = {}
    res = {}
    for v in (4, 3, 4, 3):
        outs, name, kms = run_variant(v, nd, hs, fr)
        rep.setdefault(name, []).append(kms)
        res[v] = outs
    (s4, i4, v4, p4), (s4b, i4b, v4b, p4b) = res[4]
    s3, i3, v3, p3 = res[3][0]
    ok = {"deterministic": bool(np.array_equal(s4, s4b) and np.array_equal(i4, i4b) and np.array_equal(v4, v4b))}
    worst = 0.0
    for b in range(len(nd)):
        osurf, oidx, oval = O.np_caf_surface(nd[b].astype(np.complex128), hs[b].astype(np.complex128), fr, FS)
        mx = osurf.max()
        worst = max(worst, float(np.max(np.abs(s4[b] - osurf)) / mx))
        part = np.partition(osurf, -2, axis=1)
        clear = (part[:, -1] - part[:, -2]) > 4e-3 * mx
        assert np.array_equal(i4[b][clear], oidx[clear].astype(np.int64)), f"surface {b}: row argmax"
        assert np.max(np.abs(v4[b] - oval)) <= 1e-3 * mx
        of, oi = O.np_find_peak(fr, oidx, oval)
        assert (p4[b]["freq"], int(p4[b]["idx"])) == (of, oi) and oi == lags[b], f"surface {b}: peak {p4[b]} vs {(of, oi)}"
        assert np.array_equal(v4[b], s4[b][np.arange(len(fr)), i4[b]])
    ok["max_rel_err_vs_oracle"] = worst
    ok["max_rel_diff_vs_duo"] = float(np.max(np.abs(s4 - s3)) / s3.max())
    assert worst <= 1e-3 and ok["deterministic"]
    # variations of the one-wave kernel: default-policy stores, no surface stores at all
    extra = {}
    os.environ["CAF_WAVE_STORE"] = "0"
    extra["default_policy_stores_ms"] = run_variant(4, nd, hs, fr)[2]
    os.environ.pop("CAF_WAVE_STORE")
    extra["no_surface_ms"] = run_variant(4, nd, hs, fr, surface=False)[2]
    extra["duo_no_surface_ms"] = run_variant(3, nd, hs, fr, surface=False)[2]
    out = {"parity": ok, "kernel_ms_batch256": rep, "variations": extra,
           "surfaces_per_s": {k: [256 / (m * 1e-3) for m in v] for k, v in rep.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
