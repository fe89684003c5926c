#!/usr/bin/env python3
"""bench.py -- CAF surfaces/sec on MI355X (BASELINE.json metric), one rank per GPU.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process (before it imports torch or touches HIP)
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`
as a CHILD process, relays rank 0's JSON line and exits with the child's status.  Launched by
torchrun directly (RANK/WORLD_SIZE set) it runs as one rank.

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

Rank 0 prints ONE JSON line (contract in the task statement) with
  `roofline`     dominant kernel, HBM bound, algorithmic bytes / HIP-event kernel time; for the
                 complex128 headline also `secondary` = the FP64-VALU issue ceiling of the
                 shipped instruction stream, measured live with the math-only ablation of the
                 measurement library (no LDS traffic, no loads, no stores);
  `cpu_baseline` the C restatement of caf_rust timed on this box's host cores (model and
                 core counts stated);
  `extra`        (N = 1) the other BASELINE configs measured in the same process after the
                 headline: configs[2] complex64, configs[3] 4096 x 65536 complex64 (whole surface
                 and the 512-row shard one of 8 GPUs gets), configs[4] streaming -- each with its own
                 `frac`, live `secondary` issue ceiling (`frac_of_ceiling`) and profiled `traffic`
                 (`traffic_over_algorithmic`) -- and `host_api`: the literal drop-in calls
                 (caf_surface_c128 with host pointers, peaks only / with the 26 MB surface,
                 apply_freq_shift) timed from C by tests/cpp/host_api_time next to the PCIe floor;
                 (N > 1) `configs3_c64_sharded`: every rank its row shard of ONE 4096 x 65536
                 surface + the RCCL peak reduction, and `configs4_stream_surface_parallel`: whole
                 surfaces round-robin over the ranks (the second multi-GPU decomposition), no
                 collective on the data path.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FS = 48000
N_SAMP = 4096
EXTRAS_LIMIT_S = 240           # N > 1 only: see main()
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy


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
    ap.add_argument("--peak-reduce", choices=["allreduce", "allgather"], default="allreduce",
                    help="N>1: RCCL all-reduce(max) + all-reduce(min key) (BASELINE north_star), or one all_gather")
    ap.add_argument("--overlap-peak-exchange", action="store_true",
                    help="N>1: run step k's peak exchange on a side stream under step k+1's kernels (alternating caf_peak "
                         "buffers): ~2 %% more surfaces/s with ONE rank under RCCL, but no multi-GPU box has run it yet, so the "
                         "default keeps the exchange on the main stream")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all --gpus N devices through the C ABI's caf_multi_surface_* (row shards of one "
                         "surface on per-device host threads, global peak joined on the host or by in-process RCCL): times "
                         "BASELINE configs[3] (4096 x 65536 complex64) through it; no torchrun, no torch.distributed")
    ap.add_argument("--in-process-devices", default=None,
                    help="--in-process: comma-separated device ids instead of 0..N-1 (ids may repeat, e.g. 0,0 on a one-GPU box)")
    ap.add_argument("--blocks", type=int, default=10, help="further timed blocks of --steps launches after the reported one "
                    "(extra.headline_blocks: median / min / max of the step time)")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="G",
                    help="ONE GPU, no collective: launch exactly what rank 0 of a G-GPU run launches per step (batch*G surfaces x "
                         "rows [0, F/G)); for profiling the per-rank launch shapes (profiles/r04_rankshape_*), never a scaling figure")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work: launch + rendezvous + peak reduction + JSON relay on fabricated shard peaks "
                         "(gloo); what the CPU test suite runs")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------ the line --
# Rank 0 prints ONE JSON line on stdout -- and nothing else may reach stdout: libraries under this process (RCCL prints a
# version banner on fd 1 at communicator creation, the HIP runtime prints diagnostics) write to the process's fd 1
# directly.  So the real stdout is put aside at start-up, fd 1 is pointed at stderr for everything that runs in between,
# and emit_line() writes the one line to the saved descriptor.
_REAL_STDOUT = None


def guard_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, data)
    else:
        os.write(_REAL_STDOUT, data)


# ----------------------------------------------------------------------------- launcher --
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """--gpus N > 1 outside torchrun: run the N ranks as a child process tree.  Nothing in THIS
    process has imported torch or touched HIP (a process that initialised the GPU must never be
    replaced or fork GPU users on this pool)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()),
           *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode  # stdout is inherited: rank 0's JSON line is ours


# ------------------------------------------------------------------------------ helpers --
def algorithmic_bytes(n_surfaces: int, rows_local: int, n: int, dtype: str) -> int:
    """inputs once + outputs once.  Per surface and row shard:
    needle+haystack 2*n*csize, surface rows*2n*rsize, row peaks rows*(8+rsize);
    freq list rows*8 once per launch."""
    csize, rsize = (16, 8) if dtype == "c128" else (8, 4)
    per_surface = 2 * n * csize + rows_local * (2 * n * rsize + 8 + rsize)
    return n_surfaces * per_surface + rows_local * 8


_KERNEL_HEADERS = {  # the headers a row kernel's code comes from (everything else in csrc/ cannot change it)
    "k_seq_rows": ("cplx.hpp", "kernels_fused4096.hpp", "kernels_seq4096.hpp"),
    "k_duo_rows": ("cplx.hpp", "kernels_fused4096.hpp", "kernels_seq4096.hpp", "kernels_duo4096.hpp"),
    "k_chain_rows": ("cplx.hpp", "kernels_fused4096.hpp", "kernels_seq4096.hpp", "kernels_chain.hpp"),
}


def kernel_source_hash(kernel_name: str = "") -> str:
    """sha256 over the sources of one kernel (the csrc/*.hpp it is written in; every __global__ function
    lives in a header, caf_api.hip is host code; unknown kernels: all headers): ties a
    profiles/*/traffic.json to the code it measured."""
    files = None
    for key, names in _KERNEL_HEADERS.items():
        if key in kernel_name:
            files = [ROOT / "caf_cookoff_amd" / "csrc" / n for n in names]
    if files is None:
        files = sorted((ROOT / "caf_cookoff_amd" / "csrc").glob("*.hpp"))
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def profiled_traffic(kernel_name: str, nsurf: int, dtype: str, abytes=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command in separate passes, corrected as
    MI355X_MICROARCH.md prescribes; tools/profile_pack.py).  Collected offline, so it is only
    reported when the profile's kernel-source hash equals the running code's; otherwise null."""
    here = kernel_source_hash(kernel_name)
    best, stale = None, None
    for f in sorted((ROOT / "profiles").glob("*/traffic.json")):
        try:
            t = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if t.get("kernel") and t["kernel"] in kernel_name and t.get("surfaces_per_launch") == nsurf and \
                t.get("dtype") == ("f64" if dtype == "c128" else "f32") and \
                (abytes is None or t.get("algorithmic_bytes_per_launch") in (None, abytes)):  # same rows per launch too
            if t.get("source_hash") == here:
                best = (t["traffic_bytes_per_launch"], str(f.relative_to(ROOT)))
            else:
                stale = str(f.relative_to(ROOT))
    return best, stale


def host_cpu_info():
    model = "unknown"
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return {"model": model, "nproc_online": os.cpu_count() or 1, "nproc_usable": usable}


def cpu_baseline(seconds: float, max_threads: int):
    """C restatement of caf_rust (oracle/caf_oracle.c, one task per row like CafRustFFTThreadpool,
    mod.rs:391-461) on the reference's own chirp_0 bench input, timed on this host's cores as SURVEY.md
    section 8(d) prescribes: both flavours -- 3 FFTs per row exactly like xcor_rustfft.rs:58-61 (`value`),
    and the hoisted 2-FFT variant (haystack spectrum once per surface) -- at 1 thread, at the box's CPU
    share (<= --cpu-threads) and on every usable core (mod.rs:405 uses num_cpus::get()); every figure is
    the MEDIAN of >= 20 timed runs (fewer only if the time cap is hit; the count is reported)."""
    import statistics
    from oracle import caf_oracle as O
    info = host_cpu_info()
    threads = max(1, min(info["nproc_usable"], max_threads))
    co = O.COracle()
    nd, hs = O.load_pair(O.default_data_dir(), "chirp_0_raw.c64", O.KATS[0][1])
    fr = O.bench_shifts()
    _, ridx, rval = co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=False, nthreads=threads)  # warm-up
    assert co.find_peak(fr, ridx, rval) == (69.0, 202)
    t_start = time.perf_counter()
    runs_total = [0]

    def median_ms(hoist, nthreads, share):
        """median of >= 20 runs (at least 3 when this figure's share of the time cap runs out)"""
        co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=hoist, nthreads=nthreads)
        ts, t0 = [], time.perf_counter()
        while len(ts) < 20 or (nthreads > 1 and len(ts) < 40 and time.perf_counter() - t0 < 0.15):
            t1 = time.perf_counter()
            co.caf_surface(nd, hs, fr, FS, want_surface=True, hoist=hoist, nthreads=nthreads)
            ts.append(time.perf_counter() - t1)
            if len(ts) >= 3 and time.perf_counter() - t0 > share:
                break
        runs_total[0] += len(ts)
        return statistics.median(ts) * 1e3, len(ts)

    figs = {}
    plan = [("threads_%d" % threads, threads, 0.10)]
    if info["nproc_usable"] > threads:
        plan.append(("all_usable_cores_%d" % info["nproc_usable"], info["nproc_usable"], 0.10))
    plan.append(("threads_1", 1, 0.30))
    for name, nt, share in plan:
        for flavour, hoist in (("3fft_per_row", False), ("2fft_hoisted", True)):
            ms, runs = median_ms(hoist, nt, share * seconds)
            figs.setdefault(name, {})[flavour] = {"ms_per_surface": ms, "surfaces_per_s": 1e3 / ms, "runs": runs}
    head = figs["threads_%d" % threads]["3fft_per_row"]
    el = time.perf_counter() - t_start
    return {
        "value": head["surfaces_per_s"], "unit": "surfaces/s", "cores": threads, "kind": "port",
        "sample": f"median of {head['runs']} x (400x8192 c128, chirp_0 pair, 3 FFTs/row, {threads} threads); "
                  f"{runs_total[0]} runs of all flavours in {el:.1f}s",
        "ms_per_surface": head["ms_per_surface"],
        "single_thread_ms_per_surface": figs["threads_1"]["3fft_per_row"]["ms_per_surface"],
        "flavours": figs,
        "host_cpu": info["model"], "host_nproc": info["nproc_online"], "host_nproc_usable": info["nproc_usable"],
        "published_reference_ms": {"rust RustFFT 1 thread (R9-3900X)": 177, "rust RustFFT threadpool (R9-3900X)": 28},
        "corresponds_to": {"value": "README.md's 'RustFFT + threadpool' row (CafRustFFTThreadpool, mod.rs:391-461: one pool task "
                                    "per row, 3 FFTs per row): flavours.threads_%d.3fft_per_row" % threads,
                           "single_thread_ms_per_surface": "README.md's single-thread 'RustFFT' row (CafRustFFT, mod.rs:121-166): "
                                                           "flavours.threads_1.3fft_per_row",
                           "all_usable_cores": "what mod.rs:405's ThreadPool::new(num_cpus::get()) would use on this host; slower than "
                                               "16 threads here because 400 short row tasks do not amortise that many thread starts"},
    }


class Case:
    """One (n, freq list, dtype, row shard, batch) workload with its device buffers."""

    def __init__(self, eng, torch, dev, n, freqs, dtype, batch, lo, hi, seed0=1000, want_surface=True):
        import numpy as np
        from caf_cookoff_amd.synth import make_batch
        self.torch, self.n, self.dtype, self.batch, self.freqs = torch, n, dtype, batch, freqs
        self.rows = hi - lo
        cdt = np.complex128 if dtype == "c128" else np.complex64
        rdt = torch.float64 if dtype == "c128" else torch.float32
        nd_h, hs_h, self.lags, self.fos = make_batch(batch, n, FS, seed0=seed0, dtype=cdt)
        self.nd = torch.from_numpy(nd_h).to(dev)
        self.hs = torch.from_numpy(hs_h).to(dev)
        self.plan = eng.plan(n, freqs, FS, dtype=dtype, row_begin=lo, row_end=hi)
        self.surf = torch.empty((batch, self.rows, 2 * n), dtype=rdt, device=dev) if want_surface else None
        self.ridx = torch.empty((batch, self.rows), dtype=torch.int64, device=dev)
        self.rval = torch.empty((batch, self.rows), dtype=rdt, device=dev)
        self.peak = torch.empty((batch, 4), dtype=torch.float64, device=dev)  # caf_peak records (32 B)
        self.peak_i = self.peak.view(torch.int64)

    def launch(self, plan=None, peak=None):
        (plan or self.plan).surface_dev(self.nd.data_ptr(), self.hs.data_ptr(), self.batch,
                                        self.surf.data_ptr() if self.surf is not None else None,
                                        self.ridx.data_ptr(), self.rval.data_ptr(),
                                        (self.peak if peak is None else peak).data_ptr())

    def host_peaks(self):
        import numpy as np
        pk = self.peak.cpu().numpy().view([("val", "<f8"), ("freq", "<f8"), ("idx", "<u8"), ("row", "<i8")])[:, 0]
        return pk["idx"].astype(np.int64), pk["freq"], pk["row"]

    def check(self, g_idx, g_freq, tol_hz):
        import numpy as np
        for b in range(self.batch):
            want_f = self.freqs[np.argmin(np.abs(self.freqs - self.fos[b]))]
            assert int(g_idx[b]) == self.lags[b], f"surface {b}: tau {g_idx[b]} != {self.lags[b]}"
            assert abs(float(g_freq[b]) - want_f) <= tol_hz + 1e-9, f"surface {b}: f {g_freq[b]} vs {self.fos[b]}"

    def timed(self, steps, warmup):
        """-> (seconds per step, kernel ms per launch, launches)."""
        torch = self.torch
        for _ in range(warmup):
            self.launch()
        torch.cuda.synchronize()
        self.plan.timing_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.launch()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms, launches = self.plan.timing_end()
        return el / steps, ms / max(1, launches), launches

    def close(self):
        self.plan.close()
        self.surf = self.ridx = self.rval = self.peak = self.peak_i = self.nd = self.hs = None


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source", "kernel",
                 "kernel_ms", "launches_timed", "algorithmic_bytes_per_launch", "frac_of_achievable_6.29TBs", "whole_step_frac")
MULTI_EXTRA_KEYS = ("rank_kernel_ms", "rccl_world", "headline_blocks", "configs3_c64_sharded", "configs4_stream_surface_parallel")


def assemble_line(args, *, F, n_samp, world, n_gpus_seen, nsurf, rows, K, el, kern_ms, launches, kernel_name, kernel_path,
                  devname, cu, ndev, peak_exchange):
    """The ONE place the bench line is put together: the measured run at any N and the --plumbing-only rehearsal (fabricated
    measurements) go through it, so that every line carries the same keys (tests/test_bench_launch.py compares them).
    `cpu_baseline` and `extra` are filled in by the caller after the timed region; both keys always exist."""
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
    return {
        "metric": "CAF surfaces/sec (400 freqs x 8192 samp, c128)"
                  if (F == 400 and args.dtype == "c128" and n_samp == N_SAMP)
                  else f"CAF surfaces/sec ({F} freqs x {2 * n_samp} samp, {args.dtype})",
        "value": value, "unit": "surfaces/s", "n_gpus": n_gpus_seen, "steps": K, "warmup": args.warmup,
        "ms_per_step": el / K * 1e3 if K else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if args.dtype == "c128" else "f32", "data": "synthetic",
        "config": {"workload": f"{F}x{2 * n_samp} {'complex128' if args.dtype == 'c128' else 'complex64'} "
                               f"filterbank CAF (BASELINE configs[{cfg_idx}]), n={n_samp}, fs=48000",
                   "surfaces_per_step": nsurf, "batch_per_gpu": args.batch,
                   "rows_per_gpu": rows, "parallelism": f"doppler-row-shard x{world}" if world > 1 else "single",
                   "peak_exchange": peak_exchange,
                   "kernel_path": kernel_path, "device": devname, "cus": cu,
                   "devices_visible_per_rank": ndev, "kernel_source_hash": kernel_source_hash(kernel_name)},
        "roofline": roof,
        "cpu_baseline": None,
        "extra": {},
    }


def block_stats(ms_list):
    import statistics
    return {"blocks": len(ms_list), "ms_per_step_median": statistics.median(ms_list) if ms_list else None,
            "ms_per_step_min": min(ms_list) if ms_list else None, "ms_per_step_max": max(ms_list) if ms_list else None}


def roofline_entry(abytes, kern_ms):
    achieved = abytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}


def issue_ceiling(torch, dev, case, env, steps=10):
    """Issue ceiling of a shipped row kernel, measured live: the measurement library's arithmetic-only
    ablation of the SAME kernel (wrong results, timing only) on the same batch and buffers.
      n = 4096 kernels   CAF_STORE_MODE=33: k_seq_rows / k_duo_rows with LDS traffic, barriers, global loads
                         and stores removed -> what remains is the VALU instruction stream of the row
      chain kernels      CAF_CHAIN_ABL=31 (configs[3]): no global memory, no workgroup barriers; the LDS
                         exchanges of the chain stay (without them the values would have to live in registers
                         and the kernel spills: DESIGN.md section 5)
    -> kernel ms per launch, or None."""
    import caf_cookoff_amd as caf
    if not caf.MEASURE_LIB_PATH.exists():
        return None
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        meng = caf.Engine(dev.index or 0, lib=caf.MEASURE_LIB_PATH)
        meng.set_stream(torch.cuda.current_stream().cuda_stream)
        plan = meng.plan(case.n, case.freqs, FS, dtype=case.dtype, row_begin=case.plan.row_begin,
                         row_end=case.plan.row_begin + case.rows)
        for _ in range(3):
            case.launch(plan)
        torch.cuda.synchronize()
        plan.timing_begin()
        for _ in range(steps):
            case.launch(plan)
        ms, launches = plan.timing_end()
        plan.close()
        meng.close()
        return ms / max(1, launches)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def secondary_entry(bound, ceil_ms, kern_ms, abytes, how):
    return {"bound": bound, "ceiling_ms": ceil_ms, "frac_of_ceiling": ceil_ms / kern_ms,
            "ceiling_frac_of_hbm": abytes / (ceil_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "how": how}


def traffic_fields(kernel_name, nsurf, dtype, abytes):
    """`traffic` (HBM bytes per launch from the committed PMC passes, only if their kernel-source hash equals
    the running code's), where it came from, and its ratio to the algorithmic bytes."""
    traffic, stale = profiled_traffic(kernel_name, nsurf, dtype, abytes)
    return {"traffic": traffic[0] if traffic else None,
            "traffic_over_algorithmic": traffic[0] / abytes if traffic else None,
            "traffic_source": (traffic[1] + " (rocprofv3 PMC passes of this command, collected offline; "
                               "kernel-source hash matches)") if traffic else
                              (f"none: {stale} was measured on other kernel sources" if stale else None)}


def host_api_times(reps=200):
    """The literal drop-in calls timed from C (tests/cpp/host_api_time.cpp: caf_surface_c128 with host pointers,
    peaks only / with the 26 MB surface into pageable, pinned and registered memory, caf_find_peak,
    caf_apply_freq_shift_c128, caf_xcor_c128) next to the PCIe floor of this box.  Runs as a child process."""
    exe = ROOT / "tests" / "cpp" / "host_api_time"
    try:
        if not exe.exists():
            subprocess.run(["make", "-C", str(exe.parent), "host_api_time"], check=True, capture_output=True, timeout=300)
        r = subprocess.run([str(exe), str(reps)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout).strip()[-300:]}
        out = json.loads(r.stdout)
        out["how"] = ("tests/cpp/host_api_time (C, child process): medians; with_surface = caf_surface_c128 into a "
                      "reused pageable buffer, in_place = into caf_host_alloc memory, pcie_floor = one pinned "
                      "hipMemcpy D2H of the same 26 214 400 bytes; PCIe-inclusive, never `value`")
        return out
    except Exception as e:  # reported, never fatal for the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def stream_run(plan, nd, hs, lags, total, nslots, batch, split, three_kernels=False, native=True, passes=5):
    """`total` surfaces through a caf_stream.  native: the whole loop is one caf_stream_run call (fill the
    slot's pinned buffers, replay its graph, retire the oldest slot -- in C++, as a compiled host would);
    otherwise the same loop step by step from Python (submit / wait per slot), which adds ~10 us of
    interpreter time to every step.  -> (surfaces/s, us per surface, 'ok/total')."""
    import numpy as np
    import caf_cookoff_amd as caf
    pool_n = len(lags)
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, split=split, three_kernels=three_kernels)
    best, ok = None, 0
    if native:
        reps = (total + pool_n - 1) // pool_n
        a = np.tile(nd, (reps, 1))[:total]
        b = np.tile(hs, (reps, 1))[:total]
        want = np.tile(np.asarray(lags), reps)[:total]
        st.run(a, b)  # warms the graphs up
        times = []
        for rep in range(passes):
            t0 = time.perf_counter()
            peaks, _, _ = st.run(a, b)
            dt = time.perf_counter() - t0
            times.append(dt)
        import statistics
        best = statistics.median(times)  # (the reported figure is the MEDIAN pass; min / max beside it)
        stream_run.last_spread = {"passes": passes, "value_min": total / max(times), "value_max": total / min(times)}
        ok = int(np.sum(peaks["idx"] == want))
        hs_ = st.run_stats()  # host-thread time of the LAST pass, per surface
        stream_run.last_host_us = {k.replace("_s", "_us_per_surface"): v / total * 1e6 for k, v in hs_.items()}
        stream_run.last_host_us["wall_us_per_surface_last_pass"] = dt / total * 1e6
        st.close()
        return total / best, best / total * 1e6, f"{ok}/{total}"
    bufs = [st.buffers(s) for s in range(nslots)]
    steps = max(nslots + 1, total // batch)
    for rep in range(2):
        ok = 0
        t0 = time.perf_counter()
        inflight = []
        for step in range(steps):
            slot = step % nslots
            if len(inflight) == nslots:
                s0, step0 = inflight.pop(0)
                peaks, _, _ = st.wait(s0, want_rows=False)
                ok += sum(int(peaks[j]["idx"]) == lags[(step0 * batch + j) % pool_n] for j in range(batch))
            a, b = bufs[slot]
            for j in range(batch):
                k = (step * batch + j) % pool_n
                a[j], b[j] = nd[k], hs[k]
            st.submit(slot)
            inflight.append((slot, step))
        for s0, step0 in inflight:
            peaks, _, _ = st.wait(s0, want_rows=False)
            ok += sum(int(peaks[j]["idx"]) == lags[(step0 * batch + j) % pool_n] for j in range(batch))
        best = time.perf_counter() - t0
    st.close()
    nsurf = steps * batch
    return nsurf / best, best / nsurf * 1e6, f"{ok}/{nsurf}"


def stream_case(eng, torch, freqs, total=1000):
    """BASELINE configs[4]: `total` back-to-back 400x8192 complex128 surfaces from host memory, double-
    buffered across pinned slots, one hipGraph replay per slot; surfaces stay on the device, (tau, f) + row
    peaks come back.  Sustained surfaces/s over the whole run, H2D and D2H included.  A single-surface
    chain is ONE kernel node (k_seq_surface: needle staging, haystack spectrum, rows and find_peak as roles
    of one launch) while at most two surfaces are in flight, TWO nodes {staging + spectrum | rows +
    find_peak} from three on; the host reads completion from a pinned sequence word.  Reported forms:
      single_2slots / _3slots / _4slots   one surface per graph replay, native loop (caf_stream_run)
      split4_2slots                   four independent single-surface chains per replay
      batched4_2slots / batched8_4slots   one batched chain of four / eight surfaces per replay (coarser granularity:
                                      60-62 k surfaces/s with eight per replay and four slots, tools/stream_batch_sweep.py)
      single_2slots_three_kernels     round-2a form {spectrum, rows, find_peak} as three nodes (for comparison)
      single_2slots_python_loop       submit / wait driven from Python, step by step (for comparison)
    `value` = the FIXED form batched8_4slots (eight surfaces per graph replay, four slots): the fastest form AND the one
    whose rate does not depend on how the runtime happens to map the slot streams onto its hardware queues
    (tools/stream_form_stability.py, profiles/r03_stream/form_stability.txt: 55.4-58.6 k surfaces/s over six
    creations, against 38.7-55.0 k for single_4slots, whose single-surface chains serialise when two slots share a
    queue); the other forms are reported beside it, never selected from."""
    from caf_cookoff_amd.synth import make_batch
    plan = eng.plan(N_SAMP, freqs, FS)
    nd, hs, lags, _ = make_batch(64, N_SAMP, FS, seed0=5000)
    forms = {}
    for name, nslots, batch, split, three, native in (
            ("single_2slots", 2, 1, False, False, True), ("single_3slots", 3, 1, False, False, True),
            ("single_4slots", 4, 1, False, False, True),
            ("split4_2slots", 2, 4, True, False, True), ("batched4_2slots", 2, 4, False, False, True),
            ("batched8_4slots", 4, 8, False, False, True),
            ("single_2slots_three_kernels", 2, 1, False, True, True),
            ("single_2slots_python_loop", 2, 1, False, False, False)):
        v, us, okc = stream_run(plan, nd, hs, lags, total, nslots, batch, split, three, native)
        forms[name] = {"value": v, "us_per_surface": us, "tau_correct": okc}
        if native:
            forms[name]["host_thread"] = dict(getattr(stream_run, "last_host_us", {}))
            forms[name].update(getattr(stream_run, "last_spread", {}))
    plan.close()
    abytes = algorithmic_bytes(1, 400, N_SAMP, "c128")
    best = "batched8_4slots"
    return {"workload": f"{total} back-to-back 400x8192 complex128 surfaces from host memory, hipGraph replay per slot, "
                        "stage-in of inputs and stage-out of peaks included, surfaces left on the device (BASELINE configs[4]); "
                        "the kernels read and write mapped pinned host memory in place (no hipMemcpyAsync / copy-engine nodes)",
            "value": forms[best]["value"], "unit": "surfaces/s", "form": best, "forms": forms,
            "value_is": f"median of {forms[best].get('passes')} passes over the {total} pairs (one more pass before them warms up)",
            "value_min": forms[best].get("value_min"), "value_max": forms[best].get("value_max"),
            "algorithmic_bytes_per_surface": abytes, "frac": abytes * forms[best]["value"] / 1e9 / HBM_PEAK_GBS}


def multi_stream_case(freqs, devices, total=1000):
    """caf_multi_stream_* (surface-parallel decomposition inside ONE process): one context + plan + stream per
    entry of `devices`, each on its own host thread, whole surfaces round-robin.  At N = 1 the bench runs it with
    two contexts on the one GPU -- a functional leg (results checked), not a scaling claim."""
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    nd, hs, lags, _ = make_batch(64, N_SAMP, FS, seed0=5000)
    reps = (total + 63) // 64
    a, b = np.tile(nd, (reps, 1))[:total], np.tile(hs, (reps, 1))[:total]
    want = np.tile(np.asarray(lags), reps)[:total]
    import statistics
    ms = caf.MultiStream(devices, N_SAMP, freqs, FS, nslots=3, want_surface=True)
    ms.run(a, b)
    times = []
    for rep in range(5):
        t0 = time.perf_counter()
        peaks, _, _ = ms.run(a, b)
        times.append(time.perf_counter() - t0)
    stored = ms.surface_ptr(0, 0) != 0
    ms.close()
    best = statistics.median(times)
    return {"value": total / best, "unit": "surfaces/s", "workers": len(devices), "slots_per_worker": 3,
            "surfaces_stored": stored, "value_is": "median of 5 passes", "value_min": total / max(times),
            "value_max": total / min(times), "tau_correct": f"{int(np.sum(peaks['idx'] == want))}/{total}"}


def in_process_config3(devices, steps, warmup, forms=("host_join",), check=True):
    """BASELINE configs[3] (ONE 4096 x 65536 complex64 surface) through the C ABI's caf_multi_surface_*: ONE process, worker r
    of G = len(devices) computes the Doppler rows [r*4096/G, (r+1)*4096/G) on devices[r] on its own host thread and keeps them
    in its HBM (CAF_MULTI_SURFACE_ON_DEVICE: the surface stays sharded); inputs are host pointers
    (2 x 256 KiB staged per call), row peaks and the global (tau, f) come back to the host.  Forms of the join:
      host_join   the G shard records reduced on the host
      rccl_join   ncclAllReduce(max) + ncclAllReduce(min key) inside the process over xGMI (distinct devices only)
    -> {form: {value surfaces/s, ms_per_surface, per-worker row-kernel ms, join seconds, ...}}."""
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    f3 = np.arange(4096) * 0.05 - 102.4
    nd, hs, lags, fos = make_batch(1, 32768, FS, seed0=3000, dtype=np.complex64)
    want_f = f3[np.argmin(np.abs(f3 - fos[0]))]
    out = {}
    for form in forms:
        ms = caf.MultiSurface(devices, 32768, f3, FS, dtype="c64", rccl=(form == "rccl_join"), surface_on_device=True)
        try:
            for _ in range(warmup):
                ms.run(nd[0], hs[0], want_surface=False)
            ms.timing_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                _, ridx, rval, pk = ms.run(nd[0], hs[0], want_surface=False)
            el = time.perf_counter() - t0
            kms, nl = ms.timing_end()
            stats, shard = ms.run_stats()
            info = [ms.worker_info(w) for w in range(len(devices))]
            ok = int(pk["idx"]) == lags[0] and abs(float(pk["freq"]) - want_f) <= 0.05 + 1e-9
            if check:
                assert ok, f"in-process configs[3] ({form}): peak ({pk['freq']}, {pk['idx']}) vs plant ({fos[0]}, {lags[0]})"
            per_worker = [float(k) / max(1, int(c)) for k, c in zip(kms, nl)]
            ab0 = algorithmic_bytes(1, info[0][2] - info[0][1], 32768, "c64")
            out[form] = {"value": steps / el, "unit": "surfaces/s", "ms_per_surface": el / steps * 1e3, "steps": steps,
                         "devices": list(devices), "rows_per_worker": [i[2] - i[1] for i in info],
                         "worker_kernel_ms": per_worker, "kernel": info[0][3],
                         "worker0_algorithmic_bytes": ab0,
                         "worker0_frac": roofline_entry(ab0, per_worker[0])["frac"] if per_worker[0] > 0 else None,
                         "last_call": {"shards_ms": stats["shards_s"] * 1e3, "join_ms": stats["reduce_s"] * 1e3},
                         "global_peak_correct": bool(ok), "surface": "kept on the devices (one slab of rows per worker)"}
        finally:
            ms.close()
    return out


def cpu_baseline_config3(seconds: float, max_threads: int):
    """The C restatement of caf_rust on a BOUNDED sample of configs[3]: 64 of the 4096 rows of a 65536-point surface (f64
    arithmetic: the port has no f32 path), 3 FFTs per row like the reference, thread-per-row; scaled to surfaces/s."""
    import numpy as np
    from oracle import caf_oracle as O
    from caf_cookoff_amd.synth import make_batch
    info = host_cpu_info()
    threads = max(1, min(info["nproc_usable"], max_threads))
    co = O.COracle()
    nd, hs, _, _ = make_batch(1, 32768, FS, seed0=3000)
    f3 = (np.arange(4096) * 0.05 - 102.4)[::64]
    co.caf_surface(nd[0], hs[0], f3[:8], FS, want_surface=True, hoist=False, nthreads=threads)
    ts, t0 = [], time.perf_counter()
    while len(ts) < 3 or (time.perf_counter() - t0 < seconds and len(ts) < 20):
        t1 = time.perf_counter()
        co.caf_surface(nd[0], hs[0], f3, FS, want_surface=True, hoist=False, nthreads=threads)
        ts.append(time.perf_counter() - t1)
    import statistics
    t64 = statistics.median(ts)
    return {"value": 1.0 / (t64 * 4096 / len(f3)), "unit": "surfaces/s", "cores": threads, "kind": "port",
            "sample": f"median of {len(ts)} x ({len(f3)} of the 4096 rows of one 4096x65536 surface, f64, 3 FFTs/row, {threads} threads), "
                      f"scaled by 4096/{len(f3)}",
            "ms_per_sample": t64 * 1e3, "host_cpu": info["model"], "host_nproc": info["nproc_online"],
            "host_nproc_usable": info["nproc_usable"]}


def in_process_main(args):
    """`bench.py --gpus N --in-process`: the row-shard multi-GPU path as a compiled host would drive it -- one process, the
    C ABI only (no torch.distributed).  Prints ONE JSON line in the contract's format for BASELINE configs[3]."""
    import caf_cookoff_amd as caf
    lib = caf.load()
    devices = [int(x) for x in args.in_process_devices.split(",")] if args.in_process_devices else list(range(args.gpus))
    ndev = lib.caf_device_count()
    if ndev <= 0:
        sys.exit("bench.py needs a GPU (no CPU fallback exists)")
    if max(devices) >= ndev:
        sys.exit(f"bench.py --in-process: device {max(devices)} wanted but only {ndev} device(s) are visible")
    distinct = len(set(devices)) == len(devices)
    forms = ("host_join", "rccl_join") if distinct else ("host_join",)
    K = args.steps
    res_forms = in_process_config3(devices, steps=K, warmup=max(1, args.warmup), forms=forms, check=not args.no_check)
    head = res_forms["host_join"]
    eng = caf.Engine(devices[0])
    cu, devname = eng.device_info()
    eng.close()
    kms = head["worker_kernel_ms"]
    ab = head["worker0_algorithmic_bytes"]
    roof = roofline_entry(ab, kms[0]) if kms[0] > 0 else {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}
    roof.update(traffic_fields(head["kernel"], 1, "c64", ab))
    if roof["traffic"] is None and roof["traffic_source"] is None:
        roof["traffic_source"] = f"none: no PMC profile of a {head['rows_per_worker'][0]}-row launch is committed under profiles/"
    roof.update({"kernel": head["kernel"], "kernel_ms": kms[0], "launches_timed": K, "algorithmic_bytes_per_launch": ab,
                 "frac_of_achievable_6.29TBs": roof["achieved"] / HBM_ACHIEVABLE_GBS if roof["achieved"] else None,
                 "whole_step_frac": ab / (head["ms_per_surface"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "of": "worker 0's row shard (every worker launches the same kernel on its own rows)"})
    res = {"metric": "CAF surfaces/sec (4096 freqs x 65536 samp, c64)", "value": head["value"], "unit": "surfaces/s",
           "n_gpus": len(set(devices)), "steps": K, "warmup": max(1, args.warmup), "ms_per_step": head["ms_per_surface"],
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "ONE 4096x65536 complex64 filterbank CAF per step (BASELINE configs[3]), Doppler rows sharded over "
                                  f"{len(devices)} worker(s) inside ONE process through caf_multi_surface_* (host pointers in, peaks out, "
                                  "surface rows kept in each worker's HBM), n=32768, fs=48000",
                      "devices": devices, "rows_per_worker": head["rows_per_worker"], "parallelism": f"doppler-row-shard x{len(devices)} (in-process)",
                      "peak_exchange": "host join of the shard records (value); in-process RCCL join beside it in extra",
                      "kernel_path": "chain", "device": devname, "cus": cu, "devices_visible": ndev,
                      "kernel_source_hash": kernel_source_hash(head["kernel"])},
           "roofline": roof,
           "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_config3(args.cpu_seconds, args.cpu_threads),
           "extra": {"forms": res_forms, "worker_kernel_ms": kms}}
    emit_line(res)
    return 0


def multi_gpu_extras(args, eng, torch, dist, dev, rank, world, rehearse, freqs):
    """N > 1 (every rank calls this; rank 0 keeps the result).
    configs3_c64_sharded: BASELINE configs[3] as north_star words it -- ONE 4096 x 65536 complex64 surface, rank r
      computes Doppler rows [r*4096/N, (r+1)*4096/N) and the global (tau, f) comes from the RCCL peak reduction;
      value = surfaces/s of the whole job (K steps between barriers, max over ranks).
    configs4_stream_surface_parallel: BASELINE configs[4] in the second decomposition -- 1000 host-resident pairs,
      rank r streams pairs r, r + N, ... (caf_multi_stream_share) through its own caf_stream; no collective on the
      data path; elapsed = all-reduce(MAX) over the ranks, value = 1000 / elapsed."""
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.dist import reduce_global_peak
    from caf_cookoff_amd.synth import make_batch
    cdev = "cpu" if rehearse else dev

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(x):
        t = torch.tensor([x], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def sync_all():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    out = {}
    # ---- configs[3], Doppler-row shards + peak reduction
    f3 = np.arange(4096) * 0.05 - 102.4
    lo, hi = caf.shard_range(4096, rank, world)
    c = Case(eng, torch, dev, 32768, f3, "c64", 1, lo, hi, seed0=3000)   # same seed on every rank: same pair

    def step3():
        c.launch()
        pk = c.peak.cpu() if rehearse else c.peak
        pki = pk.view(torch.int64)
        return reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=args.peak_reduce, always_collective=True)

    for _ in range(2):
        g = step3()
    sync_all()
    K3 = 10
    c.plan.timing_begin()
    t0 = time.perf_counter()
    for _ in range(K3):
        g = step3()
    sync_all()
    el = all_max(time.perf_counter() - t0)
    kms, nl = c.plan.timing_end()
    gmax, grow, gidx = g
    want_f = f3[np.argmin(np.abs(f3 - c.fos[0]))]
    ok3 = int(gidx.cpu()[0]) == c.lags[0] and abs(float(f3[int(grow.cpu()[0])]) - want_f) <= 0.05 + 1e-9
    ab = algorithmic_bytes(1, hi - lo, 32768, "c64")
    out["configs3_c64_sharded"] = {
        "workload": f"ONE 4096x65536 complex64 surface, Doppler rows sharded over {world} ranks "
                    f"({hi - lo} rows on rank 0) + RCCL peak reduction ({args.peak_reduce}) (BASELINE configs[3])",
        "value": K3 / el, "unit": "surfaces/s", "ms_per_surface": el / K3 * 1e3, "steps": K3,
        "rank0_kernel_ms": kms / max(1, nl), "rank0_kernel": c.plan.kernel_name, "global_peak_correct": bool(ok3),
        "rank0_algorithmic_bytes": ab, "rank0_frac": roofline_entry(ab, kms / max(1, nl))["frac"]}
    c.close()
    torch.cuda.empty_cache()
    # ---- configs[4], whole surfaces round-robin over the ranks
    total = 1000
    nd, hs, lags, _ = make_batch(64, N_SAMP, FS, seed0=5000)
    first, stride, items = caf.multi_stream_share(total, world, rank)
    mine = (first + stride * np.arange(items)) % 64      # pair k of the run is pool entry k % 64
    a, b, want = nd[mine], hs[mine], np.asarray(lags)[mine]
    plan = eng.plan(N_SAMP, freqs, FS)
    st = caf.Stream(plan, batch=8, nslots=4, want_surface=True)   # the fixed form of extra.configs4_stream
    st.run(a[:32], b[:32])   # warm the graphs
    best = None
    for rep in range(2):
        sync_all()
        t0 = time.perf_counter()
        peaks, _, _ = st.run(a, b)
        el4 = all_max(time.perf_counter() - t0)
        best = el4 if best is None else min(best, el4)
    okc = all_sum(int(np.sum(peaks["idx"] == want)))
    st.close()
    plan.close()
    out["configs4_stream_surface_parallel"] = {
        "workload": f"{total} back-to-back 400x8192 complex128 surfaces from host memory, whole surfaces round-robin over "
                    f"{world} ranks ({items} on rank 0), one caf_stream (eight surfaces per replay, 4 slots) per rank, no collective on the data path "
                    "(BASELINE configs[4], surface-parallel decomposition)",
        "value": total / best, "unit": "surfaces/s", "elapsed_ms_max_over_ranks": best * 1e3,
        "tau_correct": f"{okc}/{total}"}
    return out


# -------------------------------------------------------------------------- plumbing only --
def plumbing_only(args):
    """The N>1 control path without a GPU: rendezvous over gloo, reduce fabricated shard peaks with
    the product's reduce_global_peak, rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    from caf_cookoff_amd.dist import reduce_global_peak
    from caf_cookoff_amd.shifts import shard_range
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    F, nsurf = args.nfreq, 4 * world
    lo, hi = shard_range(F, rank, world)
    # surface b peaks on global row (37*b) % F; ranks that do not own the row report a lower local peak
    want_row = torch.tensor([(37 * b) % F for b in range(nsurf)], dtype=torch.int64)
    mine = (want_row >= lo) & (want_row < hi)
    val = torch.where(mine, torch.full((nsurf,), 9.0, dtype=torch.float64), torch.full((nsurf,), 1.0 + rank, dtype=torch.float64))
    row = torch.where(mine, want_row, torch.full((nsurf,), lo, dtype=torch.int64))
    idx = torch.where(mine, torch.arange(nsurf) + 100, torch.zeros(nsurf, dtype=torch.int64))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gmax, grow, gidx = reduce_global_peak(val, row, idx, method=args.peak_reduce)
    el = time.perf_counter() - t0
    assert torch.equal(grow, want_row) and torch.equal(gidx, torch.arange(nsurf) + 100) and bool((gmax == 9.0).all())
    n_seen = dist.get_world_size() if world > 1 else 1
    # the N > 1 extras' control path on fabricated numbers: ONE surface whose rows are sharded (peak on global row
    # 2500 of 4096, so exactly one rank owns it) and 1000 pairs shared out round-robin (counts summed over ranks)
    import caf_cookoff_amd as caf
    lo3, hi3 = shard_range(4096, rank, world)
    own = lo3 <= 2500 < hi3
    v3 = torch.tensor([7.0 if own else 0.5], dtype=torch.float64)
    r3 = torch.tensor([2500 if own else lo3], dtype=torch.int64)
    i3 = torch.tensor([123 if own else 9], dtype=torch.int64)
    g3 = reduce_global_peak(v3, r3, i3, method=args.peak_reduce)
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
    if world > 1:
        dist.all_gather(allk, kms)
    backend = dist.get_backend() if world > 1 else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        res = assemble_line(args, F=F, n_samp=args.n, world=world, n_gpus_seen=n_seen, nsurf=args.batch * world, rows=hi - lo,
                            K=args.steps, el=el, kern_ms=None, launches=0,
                            kernel_name="caf::k_seq_rows<double, 15, caf::SeqIo<double> >", kernel_path="fused4096",
                            devname="none (plumbing only)", cu=0, ndev=0,
                            peak_exchange=f"{args.peak_reduce}, on the main stream" if world > 1 else None)
        res["value"] = None
        res["plumbing_only"] = True
        if world > 1:
            extra.update({"rank_kernel_ms": [float(t.item()) for t in allk], "rccl_world": {"world_size": n_seen, "backend": backend},
                          "headline_blocks": block_stats([el / max(1, args.steps) * 1e3] * 2)})
        else:
            extra = {"headline_blocks": block_stats([el / max(1, args.steps) * 1e3] * 2)}
        res["extra"] = extra
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(min(args.cpu_seconds, 1.0), args.cpu_threads)
        emit_line(res)
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
    from caf_cookoff_amd.dist import reduce_global_peak

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # every rank must see at least N devices (one process per GPU on ONE node)
        t = torch.tensor([ndev], dtype=torch.int64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not rehearse and int(t.item()) < world:
            sys.exit(f"bench.py: a rank sees only {int(t.item())} device(s) for a {world}-GPU run")
    n_gpus_seen = dist.get_world_size() if coll else 1

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

    # find_peak across the row shards (dist.reduce_global_peak with --peak-reduce: RCCL all-reduce(max) + all-reduce(min
    # key), the form BASELINE's north_star names, or one all_gather of 16 B per surface and rank + a local reduction).
    # Default: on the main stream, behind the step's find_peak kernel (~0.1 ms of a 3.9 ms step during which the chip
    # idles: 63.7 k vs 65.5 k surfaces/s with one rank).  --overlap-peak-exchange: on a side stream behind an event
    # recorded after this step's find_peak kernel, on alternating caf_peak buffers; the main stream waits for a
    # buffer's previous exchange before the kernels write it again.
    overlap = coll and not rehearse and args.overlap_peak_exchange
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
            return reduce_global_peak(pk_c[:, 0], pk_ci[:, 3], pk_ci[:, 2], method=args.peak_reduce, always_collective=True)
        if not overlap:
            case.launch(peak=pk)
            pki = pk.view(torch.int64)
            return reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=args.peak_reduce, always_collective=True)
        main = torch.cuda.current_stream()
        if used[k]:
            main.wait_event(ev_done[k])
        case.launch(peak=pk)
        ev_ready[k].record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_ready[k])
            pki = pk.view(torch.int64)
            out = reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=args.peak_reduce, always_collective=True)
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

    out = None
    for _ in range(args.warmup):
        out = step()
    sync_all()

    # ---- correctness gate on the warmed-up result (cheap; outside the timed region) ----
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
        case.check(g_idx, g_freq, 0.5 if F == 400 else abs(freqs[1] - freqs[0]))

    # ---- timed region: exactly K steps between barriers --------------------------------
    K = args.steps
    sync_all()
    plan.timing_begin()  # HIP events around the dominant kernel, on the launch stream
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if coll:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kern_ms_total, launches = plan.timing_end()
    el = allreduce_max_time(el)

    kern_ms = kern_ms_total / max(1, launches)
    # ---- ten further timed blocks of K steps: the spread of the step time, and enough GPU time for a sampler to see ----
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
    # every rank's dominant-kernel time, so that a reader sees that every rank worked
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
                            peak_exchange=(f"{args.peak_reduce}, {'overlapped on a side stream' if overlap else 'on the main stream'}"
                                           if coll else None))
        if emul:
            res["config"]["parallelism"] = (f"EMULATED rank 0 of {emul}: one GPU launching that rank's per-step shape "
                                            f"({nsurf} surfaces x rows [0, {rows})), no collective, not a scaling figure")
        res["extra"]["headline_blocks"] = block_stats(blocks_ms)
        res["extra"]["headline_blocks"]["how"] = (f"{len(blocks_ms)} further blocks of {K} steps after the reported one, each between "
                                                  "barriers (max over ranks); `value` comes from the reported block only")
        if coll:
            res["extra"]["rank_kernel_ms"] = rank_kernel_ms
            res["extra"]["rccl_world"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend()}
    # ---- live VALU issue ceiling of the shipped instruction stream (n = 4096 shapes) ----
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
        extra = res["extra"]
        torch.cuda.empty_cache()

        def plan_case(name, n, freqs_x, dtype, batch, lo_x, hi_x, steps, warmup, cfg, ceiling=None):
            c = Case(eng, torch, dev, n, freqs_x, dtype, batch, lo_x, hi_x, seed0=3000)
            try:
                sec, kms, nl = c.timed(steps, warmup)
                if not args.no_check:
                    g_idx, g_freq, _ = c.host_peaks()
                    if lo_x == 0 and hi_x == len(freqs_x):
                        c.check(g_idx, g_freq, abs(freqs_x[1] - freqs_x[0]))
                ab = algorithmic_bytes(batch, hi_x - lo_x, n, dtype)
                e = roofline_entry(ab, kms)
                extra[name] = {"workload": cfg, "value": batch / sec, "unit": "surfaces/s" if hi_x - lo_x == len(freqs_x)
                               else "row-shards/s", "ms_per_step": sec * 1e3, "steps": steps,
                               "kernel": c.plan.kernel_name, "kernel_path": c.plan.path, "kernel_ms": kms,
                               "algorithmic_bytes": ab, "achieved_GBs": e["achieved"], "frac": e["frac"]}
                extra[name].update(traffic_fields(c.plan.kernel_name, batch, dtype, ab))
                if ceiling and not args.no_ceiling:
                    bound, env, how = ceiling
                    try:
                        cms = issue_ceiling(torch, dev, c, env, steps=min(10, steps))
                        if cms:
                            extra[name]["secondary"] = secondary_entry(bound, cms, kms, ab, how)
                    except Exception as ex:
                        extra[name]["secondary_error"] = f"{type(ex).__name__}: {ex}"
            finally:
                c.close()
                torch.cuda.empty_cache()

        try:
            plan_case("configs2_c64", N_SAMP, freqs, "c64", args.batch, 0, 400, max(5, min(K, 30)), 3,
                      "400x8192 complex64 filterbank CAF (BASELINE configs[2]), batch %d" % args.batch,
                      ceiling=("valu_packed_f32", {"CAF_STORE_MODE": "33"},
                               "math-only ablation of k_duo_rows<float> (libcaf_hip_measure.so, CAF_STORE_MODE=33: the "
                               "product kernel body over a null memory policy), same batch"))
            f3 = np.arange(4096) * 0.05 - 102.4   # 0.05 Hz grid
            plan_case("configs3_c64_full", 32768, f3, "c64", 1, 0, 4096, 5, 2,
                      "4096x65536 complex64 filterbank CAF, all rows on ONE GPU (BASELINE configs[3] shape)",
                      ceiling=("valu_plus_lds_exchanges", {"CAF_CHAIN_ABL": "31"},
                               "k_chain_rows<float, 14, 4> without global memory and workgroup barriers "
                               "(libcaf_hip_measure.so, CAF_CHAIN_ABL=31), same rows"))
            lo3, hi3 = caf.shard_range(4096, 3, 8)
            plan_case("configs3_c64_shard", 32768, f3, "c64", 1, lo3, hi3, 10, 2,
                      "rows [1536,2048) of 4096x65536 complex64: the shard rank 3 of 8 GPUs computes (BASELINE configs[3])")
            extra["configs4_stream"] = stream_case(eng, torch, freqs, total=1000)
            extra["configs4_stream"]["multi_ctx2_same_gpu"] = multi_stream_case(freqs, [local_rank, local_rank], total=1000)
        except Exception as e:
            extra["error"] = f"{type(e).__name__}: {e}"
        try:
            extra["in_process_multi"] = in_process_config3([local_rank], steps=10, warmup=2, forms=("host_join", "rccl_join"))
            extra["in_process_multi"]["two_contexts_same_gpu"] = in_process_config3(
                [local_rank, local_rank], steps=10, warmup=2, forms=("host_join",))["host_join"]
        except Exception as e:
            extra["in_process_multi"] = {"error": f"{type(e).__name__}: {e}"}
        extra["host_api"] = host_api_times()

    # ---- N > 1: the two multi-GPU decompositions of the other configs, all ranks take part ------
    if coll and not args.no_extra and F == 400 and n_samp == N_SAMP and args.dtype == "c128":
        torch.cuda.empty_cache()
        # A rank that fails alone (a device fault, an allocation) would leave the others inside a collective for
        # ever, and the headline measured above would never be printed: after EXTRAS_LIMIT_S rank 0 prints the line
        # without the extras and every rank leaves.
        import threading
        line_lock = threading.Lock()
        printed = [False]

        def give_up():
            # (every rank leaves with a non-zero status: a hang in the extras is a failure of the run, but the headline
            #  measured before them is still reported; the process is ended, never restarted or replaced)
            with line_lock:
                if rank == 0 and not printed[0]:
                    printed[0] = True
                    res["extra"]["error"] = (f"the multi-GPU extras did not finish within {EXTRAS_LIMIT_S} s; "
                                             "the headline above was measured before them")
                    emit_line(res)
                os._exit(3)

        watchdog = threading.Timer(EXTRAS_LIMIT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            ex = multi_gpu_extras(args, eng, torch, dist, dev, rank, world, rehearse, freqs)
        except Exception as e:
            ex = {"error": f"{type(e).__name__}: {e}"}
        with line_lock:   # from here on the main thread owns the line: a timer that fires now finds `printed` set
            watchdog.cancel()
            printed[0] = True
        if rank == 0:
            res["extra"].update(ex)

    if rank == 0:
        # the CPU comparator runs on rank 0 at ANY world size, after every timed region (the other ranks wait at the last barrier)
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_seconds, args.cpu_threads)
        emit_line(res)
    eng.close()
    if coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
