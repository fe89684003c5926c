#!/usr/bin/env python3
"""bench.py -- CAF surfaces/sec on MI355X (BASELINE.json metric), one rank per GPU.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torchrun)

A "step" is one pass of the hot path over one batch of synthetic input: `--batch`
(default 128) distinct (needle, haystack) pairs per GPU, each a 400 x 8192 complex128 filterbank
CAF (BASELINE configs[1]: n = 4096 samples, 400 shifts -100..99.5 Hz, fs = 48 kHz),
inputs resident in HBM, surfaces + per-row peaks + global peak left in HBM.

N > 1 (weak scaling): the step covers N*batch surfaces; rank r
computes the contiguous Doppler-row shard [r*F/N, (r+1)*F/N) of EVERY surface, then
one RCCL all-reduce(max) over the N*batch peak values and one all-reduce(min) over
(global_row<<32|idx) keys of the ranks that hold the max give every surface's global
(tau, f) with the reference's first-row-wins tie-break (--peak-reduce allgather does the
same with a single all_gather).  Per-GPU work is constant.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel, HBM bound, algorithmic bytes / HIP-event kernel time) and
`cpu_baseline` (the C restatement of caf_rust timed on this box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FS = 48000
N_SAMP = 4096
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy


def algorithmic_bytes(n_surfaces: int, rows_local: int, n: int, dtype: str) -> int:
    """inputs once + outputs once.  Per surface and row shard:
    needle+haystack 2*n*csize, surface rows*2n*rsize, row peaks rows*(8+rsize);
    freq list rows*8 once per launch."""
    csize, rsize = (16, 8) if dtype == "c128" else (8, 4)
    per_surface = 2 * n * csize + rows_local * (2 * n * rsize + 8 + rsize)
    return n_surfaces * per_surface + rows_local * 8


def profiled_traffic(kernel_name: str, nsurf: int, dtype: str):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under
    profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command, corrected as
    MI355X_MICROARCH.md prescribes; tools/profile_pack.py).  None if no matching profile."""
    best = None
    for f in sorted((ROOT / "profiles").glob("*/traffic.json")):
        try:
            t = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if t.get("kernel") and t["kernel"] in kernel_name and t.get("surfaces_per_launch") == nsurf and \
                t.get("dtype") == ("f64" if dtype == "c128" else "f32"):
            best = (t["traffic_bytes_per_launch"], str(f.relative_to(ROOT)))
    return best


def cpu_baseline(seconds: float, threads: int):
    """C restatement of caf_rust (oracle/caf_oracle.c; 3 FFTs per row like
    xcor_rustfft.rs:58-61, one task per row like CafRustFFTThreadpool) on the
    reference's own chirp_0 bench input, timed on this host's cores."""
    from oracle import caf_oracle as O
    co = O.COracle()
    nd, hs = O.load_pair(O.default_data_dir(), "chirp_0_raw.c64", O.KATS[0][1])
    fr = O.bench_shifts()
    _, ridx, rval = co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=False, nthreads=threads)  # warm-up
    assert co.find_peak(fr, ridx, rval) == (69.0, 202)
    t0 = time.perf_counter()
    reps = 0
    while True:
        co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=False, nthreads=threads)
        reps += 1
        el = time.perf_counter() - t0
        if el >= seconds and reps >= 3:
            break
    mt = el / reps
    # single-thread figure (README.md:28 comparator) on a short sample
    t1 = time.perf_counter()
    co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=False, nthreads=1)
    st = time.perf_counter() - t1
    return {
        "value": 1.0 / mt, "unit": "surfaces/s", "cores": threads, "kind": "port",
        "sample": f"{reps} x (400x8192 c128, chirp_0 pair, 3 FFTs/row, {threads} threads) in {el:.1f}s",
        "ms_per_surface": mt * 1e3, "single_thread_ms_per_surface": st * 1e3,
        "published_reference_ms": {"rust RustFFT 1 thread (R9-3900X)": 177, "rust RustFFT threadpool (R9-3900X)": 28},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="surfaces per GPU per step (128 x 400 rows = 100 rows "
                    "per resident workgroup on 256 CUs x 2; throughput saturates from ~128: profiles/r01_v4)")
    ap.add_argument("--dtype", choices=["c128", "c64"], default="c128")
    ap.add_argument("--nfreq", type=int, default=400)
    ap.add_argument("--n", type=int, default=N_SAMP, help="samples per input (4096 = configs[1]/[2]; "
                    "32768 with --nfreq 4096 --dtype c64 --batch 1 = configs[3])")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--peak-reduce", choices=["allreduce", "allgather"], default="allreduce",
                    help="N>1: RCCL all-reduce(max) + all-reduce(min key) (BASELINE north_star), or one all_gather")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    # Rehearsal mode for a one-GPU box (never used by the driver): every rank shares cuda:0 and
    # the peak reduction runs over gloo on CPU copies; everything else is the N>1 code path.
    rehearse = os.environ.get("CAF_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    eng = caf.Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    cu, devname = eng.device_info()

    F = args.nfreq
    n_samp = args.n
    freqs = caf.bench_shifts() if F == 400 else np.linspace(-100.0, 100.0, F, endpoint=False)
    lo, hi = caf.shard_range(F, rank, world)
    rows = hi - lo
    nsurf = args.batch * world  # surfaces per step (whole job)
    cdt = np.complex128 if args.dtype == "c128" else np.complex64
    rdt = torch.float64 if args.dtype == "c128" else torch.float32
    nd_h, hs_h, lags, fos = make_batch(nsurf, n_samp, FS, seed0=1000, dtype=cdt)
    nd = torch.from_numpy(nd_h).to(dev)
    hs = torch.from_numpy(hs_h).to(dev)
    plan = eng.plan(n_samp, freqs, FS, dtype=args.dtype, row_begin=lo, row_end=hi)
    surf = torch.empty((nsurf, rows, 2 * n_samp), dtype=rdt, device=dev)
    ridx = torch.empty((nsurf, rows), dtype=torch.int64, device=dev)
    rval = torch.empty((nsurf, rows), dtype=rdt, device=dev)
    peak = torch.empty((nsurf, 4), dtype=torch.float64, device=dev)  # caf_peak records (32 B)
    peak_i = peak.view(torch.int64)
    from caf_cookoff_amd.dist import reduce_global_peak

    def step():
        plan.surface_dev(nd.data_ptr(), hs.data_ptr(), nsurf, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                         peak.data_ptr())
        if world == 1:
            return None
        # find_peak across the row shards: RCCL all-reduce(max) + all-reduce(min) on 8 B per surface
        if rehearse:
            pk_c = peak.cpu()
            pk_ci = pk_c.view(torch.int64)
            return reduce_global_peak(pk_c[:, 0], pk_ci[:, 3], pk_ci[:, 2], method=args.peak_reduce)
        return reduce_global_peak(peak[:, 0], peak_i[:, 3], peak_i[:, 2], method=args.peak_reduce)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def allreduce_max_time(seconds: float) -> float:
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        out = step()
    sync_all()

    # ---- correctness gate on the warmed-up result (cheap; outside the timed region) ----
    if not args.no_check:
        torch.cuda.synchronize()
        if world == 1:
            pk = peak.cpu().numpy().view([("val", "<f8"), ("freq", "<f8"), ("idx", "<u8"), ("row", "<i8")])[:, 0]
            g_idx = pk["idx"].astype(np.int64)
            g_freq = pk["freq"]
        else:
            gmax, grow, gidx = out
            g_idx = gidx.cpu().numpy()
            g_freq = freqs[grow.cpu().numpy()]
        for b in range(nsurf):
            want_f = freqs[np.argmin(np.abs(freqs - fos[b]))]
            assert int(g_idx[b]) == lags[b], f"surface {b}: tau {g_idx[b]} != {lags[b]}"
            assert abs(float(g_freq[b]) - want_f) <= 0.5 + 1e-9, f"surface {b}: f {g_freq[b]} vs {fos[b]}"

    # ---- timed region: exactly K steps between barriers --------------------------------
    K = args.steps
    sync_all()
    plan.timing_begin()  # HIP events around the dominant kernel, on the launch stream
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kern_ms_total, launches = plan.timing_end()
    el = allreduce_max_time(el)

    if rank == 0:
        value = nsurf * K / el
        kern_ms = kern_ms_total / max(1, launches)
        abytes = algorithmic_bytes(nsurf, rows, n_samp, args.dtype)
        achieved = abytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = profiled_traffic(plan.kernel_name, nsurf, args.dtype) if world == 1 else None
        res = {
            "metric": "CAF surfaces/sec (400 freqs x 8192 samp, c128)"
                      if (F == 400 and args.dtype == "c128" and n_samp == N_SAMP)
                      else f"CAF surfaces/sec ({F} freqs x {2 * n_samp} samp, {args.dtype})",
            "value": value, "unit": "surfaces/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.dtype == "c128" else "f32", "data": "synthetic",
            "config": {"workload": f"{F}x{2 * n_samp} {'complex128' if args.dtype == 'c128' else 'complex64'} "
                                   f"filterbank CAF (BASELINE configs["
                                   f"{3 if n_samp == 32768 else 1 if args.dtype == 'c128' else 2}]), n={n_samp}, fs=48000",
                       "surfaces_per_step": nsurf, "batch_per_gpu": args.batch,
                       "rows_per_gpu": rows, "parallelism": f"doppler-row-shard x{world}" if world > 1 else "single",
                       "kernel_path": plan.path, "device": devname, "cus": cu},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": plan.kernel_name,
                         "kernel_ms": kern_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_launch": abytes,
                         "frac_of_achievable_6.29TBs": achieved / HBM_ACHIEVABLE_GBS,
                         "whole_step_frac": (abytes * K / el / 1e9) / HBM_PEAK_GBS},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = os.cpu_count() or 1
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                pass
            threads = min(threads, args.cpu_threads)  # a 1-GPU box's CPU share is 16 cores
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, threads)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    plan.close()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
