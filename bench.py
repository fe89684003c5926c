#!/usr/bin/env python3
"""bench.py -- CAF surfaces/sec on MI355X (BASELINE.json metric), one rank per GPU.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process (before it imports torch or touches HIP)
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`
as a CHILD process and relays rank 0's JSON line.  If that tree ends non-zero or without a line,
a SECOND fresh child runs the same headline through the one-process path (`--in-process`: no
torchrun, no rendezvous, no torch.distributed) and its line is relayed with
`config.fallback_from` = {path, rc, stderr_tail} of the path that failed.  Launched by torchrun
directly (RANK/WORLD_SIZE set) it runs as one rank; there rank 0 starts the same fallback child
itself when a phase BEFORE the headline fails or overruns (the other ranks wait for it, so the
launcher does not end rank 0 early).  A failed CORRECTNESS gate is never answered by the
fallback (a wrong answer fails the run).  No process is ever re-exec'ed or restarted
(bench_launch.py).

A "step" is one pass of the hot path over one batch of synthetic input: `--batch` (default 256)
distinct (needle, haystack) pairs per GPU, each a 400 x 8192 complex128 filterbank CAF
(BASELINE configs[1]: n = 4096 samples, 400 shifts -100..99.5 Hz, fs = 48 kHz), inputs resident
in HBM, surfaces + per-row peaks + global peak left in HBM.

N > 1 (weak scaling): the step covers N*batch surfaces; rank r computes the
contiguous Doppler-row shard [r*F/N, (r+1)*F/N) of EVERY surface, then one RCCL all-reduce(max)
over the N*batch peak values and one all-reduce(min) over (global_row<<32|idx) keys of the ranks
that hold the max give every surface's global (tau, f) with the reference's first-row-wins
tie-break (--peak-reduce allgather does the same with a single all_gather).  Per-GPU work is
constant.

`--in-process`: the same headline with NO torchrun, NO rendezvous and NO torch.distributed: ONE
process drives the N devices through the C ABI's caf_multi_surface_run_batch (row shards of the
N*batch surfaces on per-device host threads, ONE launch of the row kernel per device and step,
slabs kept in each device's HBM, the peaks joined by ONE grouped ncclAllReduce(max) + ONE
ncclAllReduce(min key) per step inside the library) -- the path a compiled host takes.

No run hangs silently: a watchdog thread (bench_common.PhaseWatchdog) bounds the rendezvous, the
warm-up, the correctness gate, the timed region and everything after it; a rank that overruns a
phase writes one stderr line (rank, device, phase) and leaves with status 3, without a result line
unless the headline had been measured before (then rank 0 prints it with `extra.error`).

Rank 0 prints ONE JSON line on stdout -- the LAST and only stdout line, at most 4 096 bytes
(bench_common.compact_line / LINE_LIMIT) -- with the contract's keys and
  `config`       the workload, who sat where (`rank_devices`), the collective backend and world
                 size (`rccl_world`) and every rank's row-kernel time (`rank_kernel_ms`, + spread
                 and a flag above 10 %);
  `roofline`     dominant kernel, HBM bound, algorithmic bytes / HIP-event kernel time; for the
                 complex128 headline also `secondary` = the FP64-VALU issue ceiling of the
                 shipped instruction stream, measured live with the math-only ablation of the
                 measurement library (no LDS traffic, no loads, no stores);
  `cpu_baseline` the C restatement of caf_rust timed on this box's host cores (model and
                 core count stated);
  `extra`        one scalar pair per other BASELINE config: (N = 1) configs[2] complex64,
                 configs[3] 4096 x 65536 complex64 (whole surface and the 512-row shard one of 8
                 GPUs gets), configs[4] streaming in its ONE fixed form and its ONE literal
                 hipMemcpyAsync-node form, `host_api` / `compiled_host_bench` (the drop-in calls
                 timed from C / C++), `in_process_headline`; (N > 1) `configs3_c64_sharded` and
                 `configs4_stream_surface_parallel`; and `detail_file`.
The FULL record (every figure, how it was taken, the per-form tables) goes to
bench_detail.json next to this file and, as one line, to stderr (whose LAST line is a copy of
the stdout line, so that a capture that appends stderr to stdout still ends with it).  `--sweeps` adds the
comparison forms (other streaming forms, both joins, `with_upload`, two contexts on one GPU) to
the full record; they never enter the line.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from datetime import timedelta
from pathlib import Path

from bench_common import (EXTRAS_LIMIT_S, FS, HBM_ACHIEVABLE_GBS, HBM_PEAK_GBS, LINE_LIMIT, N_SAMP, ROOT, Case, PhaseWatchdog,  # noqa: F401
                          algorithmic_bytes, block_stats, compact_line, emit_line, emit_result, guard_stdout, kernel_source_hash,
                          profiled_traffic, roofline_entry, secondary_entry, traffic_fields, under_rocprofiler)
from bench_launch import RankFallback, self_launch  # noqa: F401
from bench_extras import (cpu_baseline, cpu_baseline_config3, host_api_times, in_process_config3, in_process_headline,  # noqa: F401
                          issue_ceiling, multi_gpu_extras, multi_stream_case, n1_extras, stream_case, stream_run)

# init_process_group and every collective of the process group.  LONGER than every phase limit of the watchdog that can
# apply to a collective run (bench_common.PhaseWatchdog.LIMITS: <= 240 s): the phase watchdog must be the one that ends a hung
# rank -- it names rank, device and phase and, before the headline exists, lets rank 0 fall back to the one-process path --
# not torch's own NCCL watchdog, which aborts the process without either.
RENDEZVOUS_TIMEOUT_S = 360


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: ~0.75 s of GPU time at N = 1)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="surfaces per GPU per step (256 x 400 rows = 200 rows "
                    "per resident workgroup on 256 CUs x 2; 64 -> 61.1 k, 128 -> 65.2 k, 256 -> 66.8 k, 512 -> 67.4 k, "
                    "1024 -> 67.7 k surfaces/s on one box: set-up and tail amortise)")
    ap.add_argument("--dtype", choices=["c128", "c64"], default="c128")
    ap.add_argument("--nfreq", type=int, default=400)
    ap.add_argument("--n", type=int, default=N_SAMP, help="samples per input (4096 = configs[1]/[2]; "
                    "32768 with --nfreq 4096 --dtype c64 --batch 1 = configs[3])")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="cap on the whole CPU baseline leg (each figure "
                    "is the median of >= 20 runs; ~3 s on a 16-core share)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the headline CPU figure: a 1-GPU box's "
                    "CPU share is 16 cores; the all-usable-cores figure is reported beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[2]/[3]/[4] measurements (N = 1)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the live FP64-VALU ceiling measurement")
    ap.add_argument("--peak-reduce", choices=["fused", "allreduce", "allgather"], default="fused",
                    help="N>1, the global-peak exchange per step: RCCL all-reduce(max) + all-reduce(min key) (BASELINE north_star) with the "
                         "library's three element kernels either side (fused, caf_peak_exchange_stage: ~10 us of device time per step), "
                         "the same two collectives with ~20 torch tensor operations around them (allreduce: what rounds 1-5 ran, and "
                         "what `fused` falls back to on CPU tensors: gloo rehearsals, --plumbing-only), or one all_gather (allgather)")
    ap.add_argument("--overlap-peak-exchange", action="store_true",
                    help="N>1: run step k's peak exchange on a side stream under step k+1's kernels (alternating caf_peak "
                         "buffers): ~2 %% more surfaces/s with ONE rank under RCCL, but no multi-GPU box has run it yet, so the "
                         "default keeps the exchange on the main stream")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all --gpus N devices through the C ABI's caf_multi_surface_run_batch (row shards of "
                         "N*batch surfaces per step on per-device host threads, peaks joined by in-library RCCL): the headline "
                         "(BASELINE configs[1]) without torchrun / torch.distributed")
    ap.add_argument("--in-process-devices", default=None,
                    help="--in-process: comma-separated device ids instead of 0..N-1 (ids may repeat, e.g. 0,0 on a one-GPU box: "
                         "then the peaks are joined on the host, RCCL needs one rank per GPU)")
    ap.add_argument("--blocks", type=int, default=10, help="further timed blocks of --steps launches after the reported one "
                    "(extra.headline_blocks: median / min / max of the step time)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="G",
                    help="ONE GPU, no collective: launch exactly what rank 0 of a G-GPU run launches per step (batch*G surfaces x "
                         "rows [0, F/G)); for profiling the per-rank launch shapes (profiles/r04_rankshape_*), never a scaling figure")
    ap.add_argument("--sweeps", action="store_true",
                    help="also measure the comparison forms (the other streaming forms, both peak joins, the same call with the "
                         "upload inside, two contexts on one GPU, the in-process configs[3] calls): they go into bench_detail.json, "
                         "never into the line")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N>1: a failure before the headline ends the run (non-zero, no line) instead of starting the one-process "
                         "path as a fresh child")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work: launch + rendezvous + peak reduction + JSON relay on fabricated shard peaks "
                         "(gloo); what the CPU test suite runs")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------ the line --
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra")
# keys of the full record's `roofline` (bench_detail.json); the line carries bench_common._ROOFLINE_LINE_KEYS of them
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_is", "traffic_source", "kernel",
                 "kernel_ms", "launches_timed", "algorithmic_bytes_per_launch", "frac_of_achievable_6.29TBs", "whole_step_frac")
MULTI_EXTRA_KEYS = ("headline_blocks", "configs3_c64_sharded", "configs4_stream_surface_parallel")
# who sat where and that every rank worked: in `config`, which the driver's record keeps whole (`extra` it reduces to a key list)
# (rounds 1-4's driver records kept the SCALAR entries of `config`; so the same facts are there as scalars too: world size,
#  backend, "rank:device" pairs as one string, the smallest and largest kernel time, how many ranks reported one)
RANK_EVIDENCE_KEYS = ("rank_devices", "rank_device_list", "rccl_world", "rccl_world_size", "rccl_backend", "rank_kernel_ms",
                      "rank_kernel_ms_min", "rank_kernel_ms_max", "ranks_with_kernel_time", "rank_kernel_ms_spread", "rank_kernel_ms_flag")
KERNEL_SPREAD_FLAG = 0.10   # config.rank_kernel_ms_flag when (max - min) / min of the ranks' row-kernel times exceeds this


def assemble_line(args, *, F, n_samp, world, n_gpus_seen, nsurf, rows, K, el, kern_ms, launches, kernel_name, kernel_path,
                  devname, cu, ndev, peak_exchange, rank_devices=None, rccl_world=None, rank_kernel_ms=None):
    """The ONE place the bench record is put together: the measured run at any N, the in-process run and the --plumbing-only
    rehearsal (fabricated measurements) go through it, so that every record carries the same keys (tests/test_bench_launch.py
    compares them).  `cpu_baseline` and `extra` are filled in by the caller after the timed region; both keys always exist."""
    value = nsurf * K / el if el and el > 0 else None
    abytes = algorithmic_bytes(nsurf, rows, n_samp, args.dtype)
    roof = roofline_entry(abytes, kern_ms) if kern_ms else {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}
    tf = traffic_fields(kernel_name, nsurf, args.dtype, abytes)
    if tf["traffic"] is None and tf["traffic_source"] is None:
        tf["traffic_source"] = ("none: no PMC profile of this launch shape (%d surfaces x %d rows per launch) is committed under "
                                "profiles/" % (nsurf, rows))
    roof.update(tf)
    roof.update({"kernel": kernel_name, "kernel_ms": kern_ms, "launches_timed": launches,
                 "algorithmic_bytes_per_launch": abytes,
                 "frac_of_achievable_6.29TBs": roof["achieved"] / HBM_ACHIEVABLE_GBS if roof["achieved"] else None,
                 "whole_step_frac": (abytes * K / el / 1e9) / HBM_PEAK_GBS if el and el > 0 else None})
    cfg_idx = 3 if n_samp == 32768 else 1 if args.dtype == "c128" else 2
    config = {"workload": f"{F}x{2 * n_samp} {'complex128' if args.dtype == 'c128' else 'complex64'} "
                          f"filterbank CAF (BASELINE configs[{cfg_idx}]), n={n_samp}, fs=48000",
              "surfaces_per_step": nsurf, "batch_per_gpu": args.batch,
              "rows_per_gpu": rows, "parallelism": f"doppler-row-shard x{world}" if world > 1 else "single",
              "peak_exchange": peak_exchange,
              "kernel_path": kernel_path, "device": devname, "cus": cu,
              "devices_visible_per_rank": ndev,
              "rank_devices": rank_devices if rank_devices is not None else [{"rank": 0, "device": 0, "visible": ndev}],
              "rccl_world": rccl_world,
              "rccl_world_size": rccl_world["world_size"] if rccl_world else None,
              "rccl_backend": rccl_world["backend"] if rccl_world else None,
              "kernel_source_hash": kernel_source_hash(kernel_name)}
    config["rank_device_list"] = ",".join(f"{d.get('rank', d.get('worker'))}:{d['device']}" for d in config["rank_devices"])
    config.update(kernel_spread(rank_kernel_ms if rank_kernel_ms is not None else [kern_ms]))
    return {
        "metric": "CAF surfaces/sec (400 freqs x 8192 samp, c128)"
                  if (F == 400 and args.dtype == "c128" and n_samp == N_SAMP)
                  else f"CAF surfaces/sec ({F} freqs x {2 * n_samp} samp, {args.dtype})",
        "value": value, "unit": "surfaces/s", "n_gpus": n_gpus_seen, "steps": K, "warmup": args.warmup,
        "ms_per_step": el / K * 1e3 if K and el else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if args.dtype == "c128" else "f32", "data": "synthetic",
        "config": config,
        "roofline": roof,
        "cpu_baseline": None,
        "extra": {},
    }


def kernel_spread(rank_kernel_ms):
    """every rank's dominant-kernel time, their spread, and a flag when they differ by more than 10 % (a slow or throttled
    device, a rank that did less work: the max-over-ranks clock hides which) -> the three `config` entries"""
    good = [k for k in rank_kernel_ms if k and k > 0]
    out = {"rank_kernel_ms": list(rank_kernel_ms), "rank_kernel_ms_min": min(good) if good else None,
           "rank_kernel_ms_max": max(good) if good else None, "ranks_with_kernel_time": len(good),
           "rank_kernel_ms_spread": None, "rank_kernel_ms_flag": None}
    if len(good) >= 2:
        spread = (max(good) - min(good)) / min(good)
        out["rank_kernel_ms_spread"] = spread
        if spread > KERNEL_SPREAD_FLAG:
            out["rank_kernel_ms_flag"] = (f"ranks' row-kernel times differ by {spread * 100:.1f} % (> {KERNEL_SPREAD_FLAG * 100:.0f} %): "
                                          f"slowest {rank_kernel_ms.index(max(good))}, fastest {rank_kernel_ms.index(min(good))}")
    return out


# ---------------------------------------------------------------------------- in-process --
MULTI_TIMEOUT_S = 60.0   # caf_multi_surface_set_timeout: a multi-device call that does not return within it fails with CAF_ERR_TIMEOUT


def in_process_plumbing(args, devices):
    """`--in-process --plumbing-only`: the one-process control path without a GPU -- the C ABI's own host rules
    (caf_multi_surface_shard / caf_multi_surface_reduce: what the host join of caf_multi_surface_run_batch runs) on fabricated
    shard peaks of G workers, then the record through the same assembler.  What the CPU suite and the fallback rehearsal run."""
    import numpy as np
    import caf_cookoff_amd as caf
    G, F = len(devices), args.nfreq
    B = 4 * G
    wd = PhaseWatchdog(0, "cpu")
    wd.enter("setup")
    shards = [caf.multi_surface_shard(F, G, w) for w in range(G)]
    want_row = [(37 * b) % F for b in range(B)]
    rec = np.zeros((B, G), dtype=caf.Stream.PEAK_DTYPE)
    for b in range(B):
        for w, (lo, hi) in enumerate(shards):
            own = lo <= want_row[b] < hi
            rec[b, w] = (9.0 if own else 1.0 + w, 0.0, 100 + b if own else 0, want_row[b] if own else lo)
    wd.enter("in_process_timed")
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        got = [caf.multi_surface_reduce(rec[b]) for b in range(B)]
    el = time.perf_counter() - t0
    assert [int(g["row"]) for g in got] == want_row and [int(g["idx"]) for g in got] == [100 + b for b in range(B)]
    res = assemble_line(args, F=F, n_samp=args.n, world=G, n_gpus_seen=len(set(devices)), nsurf=args.batch * G, rows=shards[0][1] - shards[0][0],
                        K=args.steps, el=el, kern_ms=None, launches=0, kernel_name="caf::k_seq_rows<double, 15, caf::SeqIo<double> >",
                        kernel_path="fused4096", devname="none (plumbing only)", cu=0, ndev=0,
                        peak_exchange="host join of fabricated shard records (caf_multi_surface_reduce)",
                        rank_devices=[{"worker": i, "device": d, "visible": 0} for i, d in enumerate(devices)],
                        rank_kernel_ms=[0.5 * (w + 1) for w in range(G)])
    res["value"] = None
    res["plumbing_only"] = True
    res["config"]["parallelism"] = f"doppler-row-shard x{G} (in-process: one host thread per device, no torch.distributed)"
    res["extra"]["headline_blocks"] = block_stats([el / max(1, args.steps) * 1e3] * 2)
    wd.enter("cpu_baseline")
    res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(min(args.cpu_seconds, 1.0), args.cpu_threads)
    wd.leave()
    res["extra"]["phase_seconds"] = wd.phase_seconds()
    with wd.line_lock:
        wd.printed = True
        emit_result(res)
    return 0


def in_process_main(args):
    """`bench.py --gpus N --in-process`: the headline (BASELINE configs[1]) as a compiled host reaches it -- one process, the C
    ABI only: caf_multi_surface_run_batch over N devices, the peaks joined by in-library RCCL (bench_extras.in_process_headline).
    The host join is measured FIRST (no communicator needed: a headline exists before RCCL is touched) and is the value when
    RCCL cannot be used (not loadable, communicator creation fails or hangs, the exchange runs into the deadline, repeated
    device ids).  Every multi-device call runs under the library's own deadline (MULTI_TIMEOUT_S).
    Prints ONE JSON line in the contract's format."""
    devices = [int(x) for x in args.in_process_devices.split(",")] if args.in_process_devices else list(range(args.gpus))
    if args.plumbing_only:
        return in_process_plumbing(args, devices)
    import caf_cookoff_amd as caf
    lib = caf.load()
    wd = PhaseWatchdog(0, devices)
    wd.enter("setup")
    ndev = lib.caf_device_count()
    if ndev <= 0:
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    if max(devices) >= ndev or min(devices) < 0:
        sys.exit(f"bench.py --in-process: device {max(devices)} wanted but only {ndev} device(s) are visible")
    G = len(devices)
    distinct = len(set(devices)) == G
    K = args.steps
    eng = caf.Engine(devices[0])
    cu, devname = eng.device_info()
    plan0 = eng.plan(N_SAMP, caf.bench_shifts(), FS, dtype=args.dtype, row_begin=0, row_end=caf.multi_surface_shard(400, G, 0)[1])
    kernel_path = plan0.path
    plan0.close()
    eng.close()
    res_forms = {}

    def measure(form, blocks):
        return in_process_headline(devices, args.batch, K, args.warmup, dtype=args.dtype, forms=(form,), blocks=blocks,
                                   check=not args.no_check, timeout_s=MULTI_TIMEOUT_S, with_upload=args.sweeps)[form]

    def record_of(head_name):
        """the full record with `head_name`'s measurement as the headline"""
        head = res_forms[head_name]
        kms = head["worker_kernel_ms"]
        exchange = {"rccl_join": "in-library RCCL: ONE grouped ncclAllReduce(max) + ONE ncclAllReduce(min key) per step over the %d shard "
                                 "values (caf_multi_surface_run_batch); %d rank(s)" % (head["surfaces_per_step"], G),
                    "host_join": "host join of the G shard records per surface (caf_multi_surface_reduce)"
                                 + ("" if distinct else "; repeated device ids: RCCL needs one rank per GPU")
                                 + ("; RCCL join not usable: see bench_detail.json extra.forms.rccl_join" if distinct else "")}[head_name]
        res = assemble_line(args, F=400, n_samp=N_SAMP, world=G, n_gpus_seen=len(set(devices)), nsurf=head["surfaces_per_step"],
                            rows=head["rows_per_worker"][0], K=K, el=head["elapsed_s"], kern_ms=kms[0] if kms[0] > 0 else None,
                            launches=head["launches_timed"], kernel_name=head["kernel"], kernel_path=kernel_path, devname=devname, cu=cu,
                            ndev=ndev, peak_exchange=exchange,
                            rank_devices=[{"worker": i, "device": d, "visible": ndev} for i, d in enumerate(devices)],
                            rccl_world={"world_size": G, "backend": "rccl (in-library, ncclCommInitAll)"} if head_name == "rccl_join" else None,
                            rank_kernel_ms=kms)
        res["config"]["parallelism"] = f"doppler-row-shard x{G} (in-process: one host thread per device, no torch.distributed)"
        res["config"]["workload"] += "; one caf_multi_surface_run_batch call per step, inputs resident in every worker's HBM"
        res["config"]["roofline_of"] = "worker 0's row shard"
        res["extra"]["forms"] = res_forms
        res["extra"]["headline_blocks"] = block_stats(head["blocks_ms"])
        res["extra"]["headline_blocks"]["how"] = (f"{len(head['blocks_ms'])} further blocks of {K} calls after the reported one; `value` comes from "
                                                  "the reported block only")
        return res

    # The host join FIRST: it needs no communicator, so a measured headline exists before RCCL is touched.  This path is also the
    # fallback of a torchrun path that failed -- quite possibly for a reason RCCL shares -- and ncclCommInitAll over a broken fabric
    # is a call nothing inside the library can bound (the exchange itself is bounded by the library's deadline): if the RCCL leg
    # does not come back within its phase limit, the host-join headline is printed with the reason.
    wd.enter("in_process_timed")
    head_name = "host_join"
    res_forms["host_join"] = measure("host_join", args.blocks if not distinct else 0)
    if distinct:
        provisional = record_of("host_join")

        def print_provisional():
            if not wd.printed:
                wd.printed = True
                provisional["extra"]["error"] = ("the in-library RCCL join (communicator creation or exchange) did not finish within its phase "
                                                 "limit; this is the host join measured before it")
                provisional["extra"]["phase_seconds"] = wd.phase_seconds()
                emit_result(provisional)

        wd.enter("in_process_rccl", on_expiry=print_provisional)
        try:
            res_forms["rccl_join"] = measure("rccl_join", args.blocks)
            head_name = "rccl_join"
        except caf.CafError as e:
            if e.code not in (caf._lib.CAF_ERR_RCCL, caf._lib.CAF_ERR_TIMEOUT):
                raise
            # RCCL is not usable here (not loadable, a communicator that cannot be made, a join that ran into the deadline):
            # say so; the host join is the value (measured again, with its further blocks)
            res_forms["rccl_join"] = {"error": str(e)}
            print(f"bench.py --in-process: RCCL join not usable ({e}); the host join is the value", file=sys.stderr)
            wd.enter("in_process_timed")
            res_forms["host_join"] = measure("host_join", args.blocks)
    res = record_of(head_name)
    # From here on the headline exists: a hang in what follows costs the run its status, not the measurement
    import copy
    headline_only = copy.deepcopy(res)

    def print_headline_with_error():
        if not wd.printed:
            wd.printed = True
            headline_only["extra"]["error"] = "a phase after the timed region did not finish within its limit; the headline was measured before it"
            headline_only["extra"]["phase_seconds"] = wd.phase_seconds()
            emit_result(headline_only)

    if not args.no_extra and args.dtype == "c128":
        wd.enter("extras", on_expiry=print_headline_with_error)
        try:  # BASELINE configs[3] (ONE 4096 x 65536 complex64 surface per call) through the single-surface call of the same object
            forms3 = (("rccl_join",) if head_name == "rccl_join" else ("host_join",)) + (("host_join",) if args.sweeps and head_name == "rccl_join" else ())
            res["extra"]["configs3_single_call"] = in_process_config3(devices, steps=10, warmup=2, forms=forms3, check=not args.no_check,
                                                                       timeout_s=MULTI_TIMEOUT_S)
        except Exception as e:
            res["extra"]["configs3_single_call"] = {"error": f"{type(e).__name__}: {e}"}
    wd.enter("cpu_baseline", on_expiry=print_headline_with_error)
    res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_seconds, args.cpu_threads)
    wd.leave()
    res["extra"]["phase_seconds"] = wd.phase_seconds()
    with wd.line_lock:
        if not wd.printed:
            wd.printed = True
            emit_result(res)
    return 0


# -------------------------------------------------------------------------- plumbing only --
def torch_method(args):
    """the torch-tensor form of the exchange for places without device records (gloo rehearsals, --plumbing-only, the extras)"""
    return "allreduce" if args.peak_reduce == "fused" else args.peak_reduce


def _test_stall(rank, phase, step):
    """CPU tests only: CAF_BENCH_TEST_STALL="rank=1,phase=timed,seconds=60" makes that rank sleep inside that phase's loop;
    "...,raise=1" makes it raise instead (a rank that fails alone while the others sit in the collective)."""
    spec = dict(kv.split("=") for kv in filter(None, os.environ.get("CAF_BENCH_TEST_STALL", "").split(",")))
    if spec and int(spec.get("rank", -1)) == rank and spec.get("phase") == phase and step == 1:
        if spec.get("raise") == "1":
            raise RuntimeError(f"CAF_BENCH_TEST_STALL: rank {rank} fails in phase '{phase}' (test)")
        time.sleep(float(spec.get("seconds", 60)))


def plumbing_only(args):
    """The N>1 control path without a GPU: rendezvous over gloo, reduce fabricated shard peaks with
    the product's reduce_global_peak, rank 0 prints the JSON line.  Same phases, same watchdog, same fallback as a measured run."""
    import torch
    import torch.distributed as dist
    from caf_cookoff_amd.dist import reduce_global_peak
    from caf_cookoff_amd.shifts import shard_range
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    wd = PhaseWatchdog(rank, "cpu")
    fb = RankFallback(args, rank, world)
    if fb.enabled:
        wd.rescue = fb.run
    wd.enter("rendezvous")
    try:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=timedelta(seconds=RENDEZVOUS_TIMEOUT_S))
        wd.enter("setup")
        F, nsurf = args.nfreq, 4 * world
        lo, hi = shard_range(F, rank, world)
        # surface b peaks on global row (37*b) % F; ranks that do not own the row report a lower local peak
        want_row = torch.tensor([(37 * b) % F for b in range(nsurf)], dtype=torch.int64)
        mine = (want_row >= lo) & (want_row < hi)
        val = torch.where(mine, torch.full((nsurf,), 9.0, dtype=torch.float64), torch.full((nsurf,), 1.0 + rank, dtype=torch.float64))
        row = torch.where(mine, want_row, torch.full((nsurf,), lo, dtype=torch.int64))
        idx = torch.where(mine, torch.arange(nsurf) + 100, torch.zeros(nsurf, dtype=torch.int64))
        wd.enter("warmup")
        for i in range(max(1, args.warmup)):
            _test_stall(rank, "warmup", i)
            gmax, grow, gidx = reduce_global_peak(val, row, idx, method=torch_method(args))
        wd.enter("check")
        gate = torch.equal(grow, want_row) and torch.equal(gidx, torch.arange(nsurf) + 100) and bool((gmax == 9.0).all())
        if not gate or os.environ.get("CAF_BENCH_TEST_FAIL_GATE") == "1":   # (the variable: CPU tests of what a failed gate does)
            print(f"bench.py: CORRECTNESS GATE FAILED on rank {rank}: the reduced peaks are not the planted ones", file=sys.stderr, flush=True)
            raise AssertionError("plumbing: the reduced peaks are not the planted ones")
        wd.enter("timed")
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            _test_stall(rank, "timed", i)
            gmax, grow, gidx = reduce_global_peak(val, row, idx, method=torch_method(args))
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
    except Exception as e:   # a failure before the headline: the one-process path, if this run may fall back
        if not fb.enabled or isinstance(e, AssertionError):   # (a WRONG ANSWER is never papered over by another path's number)
            raise
        print(f"bench.py: rank {rank}: {type(e).__name__}: {e}", file=sys.stderr)
        os._exit(fb.run(f"rank {rank} failed before the headline: {type(e).__name__}: {e}"))
    wd.enter("multi_extras")
    n_seen = dist.get_world_size() if world > 1 else 1
    # the N > 1 extras' control path on fabricated numbers: ONE surface whose rows are sharded (peak on global row
    # 2500 of 4096, so exactly one rank owns it) and 1000 pairs shared out round-robin (counts summed over ranks)
    import caf_cookoff_amd as caf
    lo3, hi3 = shard_range(4096, rank, world)
    own = lo3 <= 2500 < hi3
    v3 = torch.tensor([7.0 if own else 0.5], dtype=torch.float64)
    r3 = torch.tensor([2500 if own else lo3], dtype=torch.int64)
    i3 = torch.tensor([123 if own else 9], dtype=torch.int64)
    g3 = reduce_global_peak(v3, r3, i3, method=torch_method(args))
    first, stride, items = caf.multi_stream_share(1000, world, rank)
    cnt = torch.tensor([items], dtype=torch.int64)
    tmax = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    extra = {"configs3_c64_sharded": {"plumbing": True, "global_peak_correct": (float(g3[0][0]), int(g3[1][0]), int(g3[2][0])) == (7.0, 2500, 123),
                                      "rows_rank0": hi3 - lo3 if rank == 0 else None},
             "configs4_stream_surface_parallel": {"plumbing": True, "pairs_total": int(cnt.item()), "pairs_rank0": items,
                                                  "elapsed_ms_max_over_ranks": float(tmax.item()) * 1e3}}
    # the line goes through the same assembler as a measured run (fabricated measurements: no kernel ran)
    kms = torch.tensor([0.5 * (rank + 1)], dtype=torch.float64)
    allk = [torch.zeros_like(kms) for _ in range(world)] if world > 1 else [kms]
    rd = torch.tensor([rank, 0, 0], dtype=torch.int64)   # (rank, device, devices visible): no GPU in this rehearsal
    allrd = [torch.zeros_like(rd) for _ in range(world)] if world > 1 else [rd]
    if world > 1:
        dist.all_gather(allk, kms)
        dist.all_gather(allrd, rd)
    backend = dist.get_backend() if world > 1 else None
    wd.enter("finish")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        res = assemble_line(args, F=F, n_samp=args.n, world=world, n_gpus_seen=n_seen, nsurf=args.batch * world, rows=hi - lo,
                            K=args.steps, el=el, kern_ms=None, launches=0,
                            kernel_name="caf::k_seq_rows<double, 15, caf::SeqIo<double> >", kernel_path="fused4096",
                            devname="none (plumbing only)", cu=0, ndev=0,
                            peak_exchange=f"{torch_method(args)}, on the main stream" if world > 1 else None,
                            rank_devices=[{"rank": int(t[0]), "device": int(t[1]), "visible": int(t[2])} for t in allrd],
                            rccl_world={"world_size": n_seen, "backend": backend} if world > 1 else None,
                            rank_kernel_ms=[float(t.item()) for t in allk])
        res["value"] = None
        res["plumbing_only"] = True
        if world == 1:
            extra = {}
        extra["headline_blocks"] = block_stats([el / max(1, args.steps) * 1e3] * 2)
        res["extra"] = extra
        wd.enter("cpu_baseline")
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(min(args.cpu_seconds, 1.0), args.cpu_threads)
        wd.leave()
        res["extra"]["phase_seconds"] = wd.phase_seconds()
        with wd.line_lock:
            wd.printed = True
            emit_result(res)
    wd.leave()
    return 0


# --------------------------------------------------------------------------------- main --
def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None and not args.in_process:
        sys.exit(self_launch(args))   # (the ranks are children that inherit this stdout; this process prints nothing)
    guard_stdout()
    if args.in_process:
        sys.exit(in_process_main(args))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.plumbing_only:
        sys.exit(plumbing_only(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import caf_cookoff_amd as caf
    from caf_cookoff_amd.dist import PeakExchange, reduce_global_peak

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    # Rehearsal mode for a one-GPU box (never used by the driver): every rank shares cuda:0 and
    # the peak reduction runs over gloo on CPU copies; everything else is the N>1 code path.
    rehearse = os.environ.get("CAF_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    # CAF_BENCH_FORCE_COLLECTIVES=1 (under torchrun with ONE rank): take every collective of the N > 1 path
    # through the real RCCL backend on this one GPU (process group, device-count check, barriers, the peak
    # exchange, the max-over-ranks clock) -- the closest a one-GPU box gets to the driver's --gpus 8 run
    coll = world > 1 or (os.environ.get("CAF_BENCH_FORCE_COLLECTIVES") == "1" and "WORLD_SIZE" in os.environ)
    ndev = torch.cuda.device_count()
    if rehearse:
        local_rank = 0
    elif local_rank >= ndev:
        sys.exit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {ndev} device(s) are visible")
    # from here on no phase can hang silently: the watchdog names the rank, the device and the phase and ends the process
    wd = PhaseWatchdog(rank, f"cuda:{local_rank}")
    fb = RankFallback(args, rank, world)
    if fb.enabled:
        wd.rescue = fb.run   # (phases before the headline only: PhaseWatchdog.PRE_HEADLINE)
    try:
        wd.enter("rendezvous")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        rank_devices = [{"rank": 0, "device": local_rank, "visible": ndev}]
        if coll:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            tmo = timedelta(seconds=RENDEZVOUS_TIMEOUT_S)   # (see RENDEZVOUS_TIMEOUT_S: the phase watchdog ends a hung rank first)
            if rehearse:
                dist.init_process_group("gloo", timeout=tmo)
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            # every rank must see at least N devices (one process per GPU on ONE node); and who sits on which device
            cdev = "cpu" if rehearse else dev
            t = torch.tensor([ndev], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if not rehearse and int(t.item()) < world:
                sys.exit(f"bench.py: a rank sees only {int(t.item())} device(s) for a {world}-GPU run")
            mine = torch.tensor([rank, local_rank, ndev], dtype=torch.int64, device=cdev)
            parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, mine)
            rank_devices = [{"rank": int(p[0]), "device": int(p[1]), "visible": int(p[2])} for p in (q.cpu() for q in parts)]
        n_gpus_seen = dist.get_world_size() if coll else 1

        wd.enter("setup")
        eng = caf.Engine(local_rank)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        cu, devname = eng.device_info()

        F = args.nfreq
        n_samp = args.n
        freqs = caf.bench_shifts() if F == 400 else np.linspace(-100.0, 100.0, F, endpoint=False)
        emul = args.emulate_rank_of if (args.emulate_rank_of > 1 and world == 1) else 0
        lo, hi = caf.shard_range(F, 0, emul) if emul else caf.shard_range(F, rank, world)
        rows = hi - lo
        nsurf = args.batch * (emul or world)  # surfaces per step (whole job)
        if emul:
            args.no_check = args.no_extra = True   # (the planted peaks need not lie in shard 0; the extras are N = 1 measurements)
        case = Case(eng, torch, dev, n_samp, freqs, args.dtype, nsurf, lo, hi)
        plan = case.plan

        # find_peak across the row shards (--peak-reduce): RCCL all-reduce(max) + all-reduce(min key), the form BASELINE's
        # north_star names -- with the library's three element kernels either side (fused: dist.PeakExchange, round 6) or with
        # ~20 torch tensor operations around them (allreduce: dist.reduce_global_peak, ~0.1 ms of a 3.9 ms step during which the
        # chip idles: 63.7 k vs 65.5 k surfaces/s with one rank) -- or one all_gather of 16 B per surface and rank + a local
        # reduction.  On the main stream, behind the step's find_peak kernel.  --overlap-peak-exchange: on a side stream behind an event
        # recorded after this step's find_peak kernel, on alternating caf_peak buffers; the main stream waits for a
        # buffer's previous exchange before the kernels write it again.
        overlap = coll and not rehearse and args.overlap_peak_exchange
        method = torch_method(args) if (rehearse or overlap) else args.peak_reduce   # (gloo moves CPU tensors; the side stream is torch's)
        px = PeakExchange(eng, nsurf, freqs, dev) if (coll and method == "fused") else None
        peaks = [case.peak, torch.empty_like(case.peak)] if coll else [case.peak]
        side = torch.cuda.Stream(device=dev) if overlap else None
        ev_ready = [torch.cuda.Event(), torch.cuda.Event()]
        ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        used = [False, False]
        nstep = [0]

        def step():
            if not coll:
                case.launch()
                return None
            k = nstep[0] & 1
            nstep[0] += 1
            pk = peaks[k]
            if rehearse:
                case.launch(peak=pk)
                pk_c = pk.cpu()
                pk_ci = pk_c.view(torch.int64)
                return reduce_global_peak(pk_c[:, 0], pk_ci[:, 3], pk_ci[:, 2], method=method, always_collective=True)
            if px is not None:
                case.launch(peak=pk)
                return PeakExchange.peaks_of(px(pk, always_collective=True))   # (views of the exchange's output records: no kernel)
            if not overlap:
                case.launch(peak=pk)
                pki = pk.view(torch.int64)
                return reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=method, always_collective=True)
            main_stream = torch.cuda.current_stream()
            if used[k]:
                main_stream.wait_event(ev_done[k])
            case.launch(peak=pk)
            ev_ready[k].record(main_stream)
            with torch.cuda.stream(side):
                side.wait_event(ev_ready[k])
                pki = pk.view(torch.int64)
                out = reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=method, always_collective=True)
                ev_done[k].record(side)
            used[k] = True
            return out

        def sync_all():
            torch.cuda.synchronize()
            if coll:
                dist.barrier()
                torch.cuda.synchronize()

        def allreduce_max_time(seconds: float) -> float:
            t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearse else dev)
            if coll:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        wd.enter("warmup")
        out = None
        for i in range(args.warmup):
            _test_stall(rank, "warmup", i)
            out = step()
        sync_all()

        # ---- correctness gate on the warmed-up result (cheap; outside the timed region) ----
        wd.enter("check")
        if not args.no_check:
            if args.warmup == 0:
                out = step()
            torch.cuda.synchronize()
            if not coll:
                g_idx, g_freq, _ = case.host_peaks()
            else:
                gmax, grow, gidx = out
                g_idx = gidx.cpu().numpy()
                g_freq = freqs[grow.cpu().numpy()]
            try:
                case.check(g_idx, g_freq, 0.5 if F == 400 else abs(freqs[1] - freqs[0]))
            except AssertionError as e:   # (the launcher reads this marker: no other path's number replaces a wrong answer)
                print(f"bench.py: CORRECTNESS GATE FAILED on rank {rank}: {e}", file=sys.stderr, flush=True)
                raise

        # ---- timed region: exactly K steps between barriers --------------------------------
        wd.enter("timed")
        K = args.steps
        sync_all()
        plan.timing_begin()  # HIP events around the dominant kernel, on the launch stream
        t0 = time.perf_counter()
        for i in range(K):
            _test_stall(rank, "timed", i)
            step()
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kern_ms_total, launches = plan.timing_end()
        el = allreduce_max_time(el)

    except Exception as e:   # a failure before the headline: the one-process path as a fresh child, if this run may fall back
        if not fb.enabled or isinstance(e, AssertionError):   # (a WRONG ANSWER -- the correctness gate -- is never papered over)
            raise
        import traceback
        traceback.print_exc()
        os._exit(fb.run(f"rank {rank} failed before the headline: {type(e).__name__}: {e}"))
    kern_ms = kern_ms_total / max(1, launches)
    # every rank's dominant-kernel time, so that a reader sees that every rank worked
    wd.enter("blocks")
    rank_kernel_ms = [kern_ms]
    if coll:
        tk = torch.tensor([kern_ms], dtype=torch.float64, device="cpu" if rehearse else dev)
        parts = [torch.zeros_like(tk) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, tk)
        rank_kernel_ms = [float(t.item()) for t in parts]

    res = None
    if rank == 0:
        res = assemble_line(args, F=F, n_samp=n_samp, world=world, n_gpus_seen=n_gpus_seen, nsurf=nsurf, rows=rows, K=K, el=el,
                            kern_ms=kern_ms, launches=launches, kernel_name=plan.kernel_name, kernel_path=plan.path,
                            devname=devname, cu=cu, ndev=ndev,
                            peak_exchange=(f"{method}, {'overlapped on a side stream' if overlap else 'on the main stream'}"
                                           if coll else None), rank_devices=rank_devices,
                            rccl_world={"world_size": dist.get_world_size(), "backend": dist.get_backend()} if coll else None,
                            rank_kernel_ms=rank_kernel_ms)
        if emul:
            res["config"]["parallelism"] = (f"EMULATED rank 0 of {emul}: one GPU launching that rank's per-step shape "
                                            f"({nsurf} surfaces x rows [0, {rows})), no collective, not a scaling figure")

    # From here on the headline exists: a hang in what follows still costs the run its status (3), but rank 0 prints the line
    # it measured, with the reason, before it leaves.
    import copy
    headline_only = copy.deepcopy(res)   # (the main thread keeps filling res["extra"]: the watchdog prints this snapshot)

    def print_headline_with_error():
        if rank == 0 and not wd.printed:
            wd.printed = True
            headline_only["extra"]["error"] = ("a phase after the timed region did not finish within its limit; the headline "
                                               "was measured before it")
            headline_only["extra"]["phase_seconds"] = wd.phase_seconds()
            emit_result(headline_only)

    # ---- further timed blocks of K steps: the spread of the step time, and enough GPU time for a sampler to see ----
    wd.enter("blocks", on_expiry=print_headline_with_error)
    blocks_ms = []
    for _ in range(max(0, args.blocks)):
        sync_all()
        tb = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
            torch.cuda.synchronize()
        blocks_ms.append(allreduce_max_time(time.perf_counter() - tb) / K * 1e3)
    if rank == 0:
        res["extra"]["headline_blocks"] = block_stats(blocks_ms)
        res["extra"]["headline_blocks"]["how"] = (f"{len(blocks_ms)} further blocks of {K} steps after the reported one, each between "
                                                  "barriers (max over ranks); `value` comes from the reported block only")

    # ---- live VALU issue ceiling of the shipped instruction stream (n = 4096 shapes) ----
    wd.enter("ceiling", on_expiry=print_headline_with_error)
    if rank == 0 and world == 1 and n_samp == N_SAMP and not args.no_ceiling:
        try:
            ceil_ms = issue_ceiling(torch, dev, case, {"CAF_STORE_MODE": "33"})
        except Exception as e:  # measurement aid only: never fail the bench line for it
            ceil_ms = None
            print(f"bench.py: VALU ceiling not measured: {e}", file=sys.stderr)
        if ceil_ms:
            res["roofline"]["secondary"] = secondary_entry(
                "valu_f64" if args.dtype == "c128" else "valu_packed_f32", ceil_ms, res["roofline"]["kernel_ms"],
                res["roofline"]["algorithmic_bytes_per_launch"],
                "math-only ablation of the product kernel (libcaf_hip_measure.so, CAF_STORE_MODE=33), same batch")
    case.close()

    # ---- the other BASELINE configs, same process, after the headline (N = 1 only) ------
    if rank == 0 and world == 1 and not args.no_extra and F == 400 and n_samp == N_SAMP and args.dtype == "c128":
        wd.enter("extras", on_expiry=print_headline_with_error)
        n1_extras(args, eng, torch, dev, local_rank, freqs, K, res["extra"])

    # ---- N > 1: the two multi-GPU decompositions of the other configs, all ranks take part ------
    if coll and not args.no_extra and F == 400 and n_samp == N_SAMP and args.dtype == "c128":
        torch.cuda.empty_cache()
        # A rank that fails alone (a device fault, an allocation) would leave the others inside a collective for
        # ever: past the limit rank 0 prints the headline with extra.error and every rank leaves with status 3.
        wd.enter("multi_extras", on_expiry=print_headline_with_error)
        try:
            ex = multi_gpu_extras(args, eng, torch, dist, dev, rank, world, rehearse, freqs)
        except Exception as e:
            ex = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["extra"].update(ex)

    if rank == 0:
        # the CPU comparator runs on rank 0 at ANY world size, after every timed region (the other ranks wait at the last barrier)
        wd.enter("cpu_baseline", on_expiry=print_headline_with_error)
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_seconds, args.cpu_threads)
        wd.leave()
        res["extra"]["phase_seconds"] = wd.phase_seconds()
        with wd.line_lock:   # from here on the main thread owns the line: a watchdog that fires now finds `printed` set
            if not wd.printed:
                wd.printed = True
                emit_result(res)
    wd.enter("finish")
    eng.close()
    if coll:
        dist.barrier()
        dist.destroy_process_group()
    wd.leave()


if __name__ == "__main__":
    main()
