"""bench_common.py -- what bench.py and bench_extras.py share: the constants of the contract, the algorithmic-byte count of
SURVEY.md section 8(d), the roofline / traffic objects, the one-line output discipline, the device-resident workload (`Case`)
and the phase watchdog that keeps a run from hanging silently."""
from __future__ import annotations

import hashlib
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

FS = 48000
N_SAMP = 4096
EXTRAS_LIMIT_S = 240           # N > 1 only: see main()
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy


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


def emit_line(obj, echo=True):
    """the ONE stdout line -- and, last thing on stderr, a copy of it behind a prefix: whoever looks at the END of a capture
    that puts stderr after stdout (the driver's record keeps 8 KB of such a tail) finds the line there too, not only the
    end of the long detail record"""
    text = json.dumps(obj, allow_nan=False)
    data = (text + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, data)
    else:
        os.write(_REAL_STDOUT, data)
    if echo:
        os.write(2, ("bench.py line: " + text + "\n").encode())


# The contract line is SMALL (round 5's grew to 21 KB and the driver could no longer recover it: BENCH_r05.json parsed = null).
# Everything a run measures goes into the full record; the full record goes to bench_detail.json next to bench.py (and, as one
# line, to stderr); the LAST -- and only -- stdout line is the compact form below: the contract's keys, `roofline`,
# `cpu_baseline`, and one scalar pair per other config under `extra`.  tests/test_bench_launch.py holds it to LINE_LIMIT.
LINE_LIMIT = 4096
DETAIL_FILE = "bench_detail.json"
_CONFIG_KEYS = ("workload", "surfaces_per_step", "batch_per_gpu", "rows_per_gpu", "parallelism", "peak_exchange", "kernel_path",
                "device", "cus", "devices_visible_per_rank", "rank_devices", "rank_device_list", "rccl_world", "rccl_world_size",
                "rccl_backend", "rank_kernel_ms", "rank_kernel_ms_min", "rank_kernel_ms_max", "ranks_with_kernel_time",
                "rank_kernel_ms_spread", "rank_kernel_ms_flag", "kernel_source_hash", "fallback_from", "child_rc", "roofline_of")
_ROOFLINE_LINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source", "kernel",
                       "kernel_ms", "launches_timed", "algorithmic_bytes_per_launch", "frac_of_achievable_6.29TBs", "whole_step_frac")
_CPU_LINE_KEYS = ("value", "unit", "cores", "kind", "ms_per_surface", "single_thread_ms_per_surface", "host_cpu", "sample")


def _sig(x, digits=6):
    """floats of the compact line: six significant digits (the full record keeps every digit)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{digits}g}")


def _pick(d, keys, digits=6):
    """scalars of `d` under `keys`; byte counts stay exact"""
    return {k: (d[k] if k in ("traffic", "algorithmic_bytes_per_launch") else _sig(d[k], digits))
            for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def compact_extra(extra):
    """`extra` of the line: one scalar pair per other config; everything else stays in the detail file."""
    out = {}
    if not isinstance(extra, dict):
        return out
    hb = extra.get("headline_blocks")
    if isinstance(hb, dict):
        out["headline_blocks"] = _pick(hb, ("blocks", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max"), 5)
    picks = {"configs2_c64": ("value", "frac"), "configs3_c64_full": ("ms_per_step", "frac", "traffic_over_algorithmic"),
             "configs3_c64_shard": ("ms_per_step", "frac"), "configs4_stream": ("value", "frac", "memcpy_nodes_value"),
             "configs3_c64_sharded": ("value", "ms_per_surface", "rank0_frac", "global_peak_correct"),
             "configs4_stream_surface_parallel": ("value", "tau_correct"),
             "compiled_host_bench": ("batch_resident_surfaces_per_s", "literal_loop_ms_per_surface"),
             "host_api": ("peaks_only_us", "with_surface_ms")}
    for name, keys in picks.items():
        v = extra.get(name)
        if isinstance(v, dict):
            out[name] = {"error": str(v["error"])[:120]} if "error" in v else _pick(v, keys, 5)
            if name.startswith("configs3_c64_") and "ms_per_step" in out[name]:
                out[name]["ms"] = out[name].pop("ms_per_step")
    iph = extra.get("in_process_headline")
    if isinstance(iph, dict):
        head = iph.get("rccl_join") if isinstance(iph.get("rccl_join"), dict) and "value" in iph["rccl_join"] else iph.get("host_join")
        out["in_process_headline"] = ({"value": _sig(head["value"], 5), "join": "rccl" if head is iph.get("rccl_join") else "host"}
                                      if isinstance(head, dict) and "value" in head else {"error": str(iph.get("error", "no figure"))[:120]})
    c3 = extra.get("configs3_single_call")
    if isinstance(c3, dict):
        f = c3.get("rccl_join") if isinstance(c3.get("rccl_join"), dict) else c3.get("host_join")
        out["configs3_single_call"] = _pick(f, ("value", "ms_per_surface", "worker0_frac"), 5) if isinstance(f, dict) else \
            {"error": str(c3.get("error", "no figure"))[:120]}
    for k in ("error", "plumbing"):
        if k in extra:
            out[k] = str(extra[k])[:200] if k == "error" else extra[k]
    ps = extra.get("phase_seconds")
    if isinstance(ps, dict):
        out["phase_seconds"] = {k: round(v, 2) for k, v in ps.items()}
    return out


def compact_line(res):
    """the contract line of a full record (never larger than LINE_LIMIT: see shrink_to_limit)"""
    line = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    cfg = res.get("config") or {}
    c = {}
    for k in _CONFIG_KEYS:
        if k in cfg:
            v = cfg[k]
            if k == "rank_devices" and isinstance(v, list):
                v = [{kk: d[kk] for kk in ("rank", "worker", "device") if kk in d} for d in v]
            elif k == "rank_kernel_ms" and isinstance(v, list):
                v = [_sig(x, 5) for x in v]
            elif k == "fallback_from" and isinstance(v, dict):
                v = {"path": v.get("path"), "rc": v.get("rc"), "stderr_tail": str(v.get("stderr_tail", ""))[-300:]}
            elif isinstance(v, str):
                v = v[:200]
            c[k] = _sig(v)
    line["config"] = c
    roof = res.get("roofline") or {}
    r = _pick(roof, _ROOFLINE_LINE_KEYS)
    if isinstance(r.get("traffic_source"), str):
        r["traffic_source"] = r["traffic_source"].split(" (")[0][:120]
    if isinstance(roof.get("secondary"), dict):
        r["secondary"] = _pick(roof["secondary"], ("bound", "ceiling_ms", "frac_of_ceiling", "ceiling_frac_of_hbm"), 5)
    line["roofline"] = r
    cb = res.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, _CPU_LINE_KEYS) if isinstance(cb, dict) else None
    if line["cpu_baseline"] and isinstance(line["cpu_baseline"].get("sample"), str):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
    line["extra"] = compact_extra(res.get("extra"))
    for k in ("plumbing_only",):
        if k in res:
            line[k] = res[k]
    return line


def shrink_to_limit(line, limit=LINE_LIMIT):
    """A line that would still exceed the limit loses `extra` entries, largest first (never a contract key): the driver
    must be able to read the line whatever a future run adds to it."""
    def size(o):
        return len(json.dumps(o, allow_nan=False)) + 1
    dropped = []
    while size(line) > limit and any(k not in ("detail_file", "dropped") for k in line["extra"]):
        k = max((k for k in line["extra"] if k not in ("detail_file", "dropped")), key=lambda k: size(line["extra"][k]))
        del line["extra"][k]
        dropped.append(k)
        line["extra"]["dropped"] = dropped
    return line


def write_detail(res):
    """the full record -> bench_detail.json next to bench.py (CAF_BENCH_DETAIL overrides; /tmp if the tree is read-only) and,
    as ONE line, to stderr.  -> the path written, or None."""
    text = json.dumps(res, allow_nan=False, default=str)
    os.write(2, ("bench.py detail: " + text + "\n").encode())
    if (ROOT / "gpurun_out").is_dir():   # (on a gpurun box: what is written there travels back with the call)
        try:
            (ROOT / "gpurun_out" / DETAIL_FILE).write_text(text + "\n")
        except OSError:
            pass
    for cand in (os.environ.get("CAF_BENCH_DETAIL"), str(ROOT / DETAIL_FILE), "/tmp/" + DETAIL_FILE):
        if not cand:
            continue
        try:
            Path(cand).write_text(text + "\n")
            return cand
        except OSError:
            continue
    return None


def sanitize(o):
    """NaN / inf have no JSON spelling: null them (the strict parser the tests use rejects the Python spellings)"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k): sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [sanitize(v) for v in o]
    return o


def emit_result(res):
    """detail file + stderr copy of the full record, then the ONE stdout line."""
    res = sanitize(res)
    path = write_detail(res)
    line = compact_line(res)
    line["extra"]["detail_file"] = path
    emit_line(shrink_to_limit(line))


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
                best = (t["traffic_bytes_per_launch"], str(f.relative_to(ROOT)), t)
            else:
                stale = str(f.relative_to(ROOT))
    return best, stale


def traffic_detail(kernel_name: str, nsurf: int, dtype: str, abytes=None):
    """What the committed profile of this launch shape says beyond the byte count: the L2's memory-side request counters
    (profiles/*/traffic.json `fabric`), whether an HBM / Infinity-Cache split exists (`hbm_bytes_per_launch`, `hbm_split`) and,
    where tools/traffic_components.py has run, which buffers the bytes consist of -- each only under the source-hash rule."""
    best, _ = profiled_traffic(kernel_name, nsurf, dtype, abytes)
    if not best:
        return None
    t, src = best[2], Path(best[1])
    out = {"source": str(src), "hbm_bytes_per_launch": t.get("hbm_bytes_per_launch"), "hbm_split": t.get("hbm_split"),
           "fabric": t.get("fabric")}
    comp = ROOT / src.parent / "traffic_components.json"
    if comp.exists():
        try:
            c = json.loads(comp.read_text())
            if c.get("source_hash") == kernel_source_hash(kernel_name):
                out["components"] = c.get("components")
                out["components_source"] = str(comp.relative_to(ROOT))
        except (OSError, ValueError):
            pass
    return out


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


def block_stats(ms_list):
    import statistics
    return {"blocks": len(ms_list), "ms_per_step_median": statistics.median(ms_list) if ms_list else None,
            "ms_per_step_min": min(ms_list) if ms_list else None, "ms_per_step_max": max(ms_list) if ms_list else None}


def roofline_entry(abytes, kern_ms):
    achieved = abytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}


def secondary_entry(bound, ceil_ms, kern_ms, abytes, how):
    return {"bound": bound, "ceiling_ms": ceil_ms, "frac_of_ceiling": ceil_ms / kern_ms,
            "ceiling_frac_of_hbm": abytes / (ceil_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "how": how}


def traffic_fields(kernel_name, nsurf, dtype, abytes):
    """`traffic` (HBM bytes per launch from the committed PMC passes, only if their kernel-source hash equals
    the running code's), where it came from, and its ratio to the algorithmic bytes."""
    traffic, stale = profiled_traffic(kernel_name, nsurf, dtype, abytes)
    return {"traffic": traffic[0] if traffic else None,
            "traffic_over_algorithmic": traffic[0] / abytes if traffic else None,
            "traffic_is": "bytes that leave L2 (misses + write-backs: 2 x FETCH_SIZE + WRITE_SIZE); whether the Infinity Cache or an HBM "
                          "channel serves them is not exposed by rocprofv3 on gfx950 (no MALL / UMC counter)",
            "traffic_source": (traffic[1] + " (rocprofv3 PMC passes of this command, collected offline; "
                               "kernel-source hash matches)") if traffic else
                              (f"none: {stale} was measured on other kernel sources" if stale else None)}


# --------------------------------------------------------------------------- phase watchdog --
class PhaseWatchdog:
    """A run that hangs must end by itself, loudly, and must never print a line it did not measure (the join that cannot be
    skipped, mod.rs:452-457, has no timeout in the reference; a collective over 8 GPUs needs one).  One daemon thread per
    process watches the phase the main thread says it is in -- rendezvous, set-up, warm-up, the correctness gate, the timed
    region, the further blocks, the extras, the CPU comparator -- each with its own limit.  When a phase overruns, the process
    writes ONE stderr line naming its rank, its device and the phase and leaves with status 3 through os._exit (a thread
    blocked inside a collective cannot be unwound; the process ends, it is never restarted or replaced).  `on_expiry` (per
    phase) runs first, under `line_lock`: the extras phases use it to print the headline that was measured BEFORE the hang.
    Limits (seconds) can be overridden for tests: CAF_BENCH_PHASE_LIMITS="timed=5,warmup=5"."""

    # (a healthy default run spends < 1 s in each of the first five, ~8 s in the blocks, ~10 s in the N = 1 extras; the limits
    #  are sized so that a rank stuck in any ONE phase leaves well inside a 600-second outer limit)
    # (rendezvous: 240 s -- on a fresh node the ranks' first `import torch` can take minutes and they do not finish it together)
    LIMITS = {"rendezvous": 240.0, "setup": 180.0, "warmup": 120.0, "check": 90.0, "timed": 150.0, "blocks": 180.0,
              "ceiling": 120.0, "extras": 420.0, "multi_extras": float(EXTRAS_LIMIT_S), "cpu_baseline": 120.0, "finish": 90.0,
              "in_process_timed": 300.0, "in_process_rccl": 240.0}

    def __init__(self, rank: int, device):
        self.rank, self.device = rank, device
        self.limits = dict(self.LIMITS)
        for item in filter(None, os.environ.get("CAF_BENCH_PHASE_LIMITS", "").split(",")):
            k, _, v = item.partition("=")
            self.limits[k.strip()] = float(v)
        self.line_lock = threading.Lock()
        self.printed = False       # the one JSON line has been written (by whoever holds line_lock)
        self._lock = threading.Lock()
        self._phase, self._deadline, self._limit, self._cb = None, None, None, None
        self.history = []          # (phase, seconds) of every finished phase: goes into extra.phase_seconds
        self._t_enter = None
        threading.Thread(target=self._watch, name="bench-watchdog", daemon=True).start()

    def enter(self, phase: str, on_expiry=None):
        now = time.monotonic()
        with self._lock:
            if self._phase is not None:
                self.history.append((self._phase, now - self._t_enter))
            self._limit = self.limits.get(phase, 300.0)
            self._phase, self._deadline, self._cb, self._t_enter = phase, now + self._limit, on_expiry, now

    def leave(self):
        now = time.monotonic()
        with self._lock:
            if self._phase is not None:
                self.history.append((self._phase, now - self._t_enter))
            self._phase, self._deadline, self._cb = None, None, None

    def phase_seconds(self):
        out = {}
        for ph, s in self.history:
            out[ph] = round(out.get(ph, 0.0) + s, 3)
        return out

    # phases before the headline exists: an overrun there may be answered by `rescue` (bench.RankFallback.run: the one-process
    # path as a fresh child on rank 0, the other ranks wait for its verdict) instead of leaving empty-handed
    PRE_HEADLINE = ("rendezvous", "setup", "warmup", "check", "timed")
    rescue = None   # callable(reason) -> exit status; set by bench.py for collective runs under an external launcher

    def _watch(self):
        while True:
            time.sleep(0.2)
            with self._lock:
                ph, dl, lim, cb = self._phase, self._deadline, self._limit, self._cb
            if ph is None or time.monotonic() <= dl:
                continue
            status = 3
            try:
                if cb is not None:
                    with self.line_lock:
                        cb()
            finally:
                what = f"rank {self.rank} (device {self.device}) did not finish phase '{ph}' within {lim:g} s"
                if self.rescue is not None and ph in self.PRE_HEADLINE and not self.printed:
                    os.write(2, f"bench.py: {what}\n".encode())
                    try:
                        status = int(self.rescue(what))
                    except Exception as e:   # the rescue is best effort: the run still ends here, loudly
                        os.write(2, f"bench.py: fallback failed: {type(e).__name__}: {e}\n".encode())
                        status = 3
                    os.write(2, f"bench.py: rank {self.rank} leaving with status {status} after the fallback\n".encode())
                else:
                    os.write(2, (f"bench.py: {what}; leaving with status 3" + ("" if self.printed else " and without a result line")
                                 + "\n").encode())
                os._exit(status)


def under_rocprofiler() -> bool:
    """rocprofv3 preloads its tool library into the profiled program, and that library initialises the GPU before the program's
    first line runs: from such a process no other GPU program may be started (tools/profile_run.sh, bench.self_launch).
    Detected by the preload itself -- LD_PRELOAD naming a rocprofiler library, or the variables through which the profiler's
    launcher hands its tool library to the runtime -- not by any ROCPROF* variable a shell or a site profile may carry
    (ROCPROFILER_LOG_LEVEL and the like say nothing about a preload)."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in pre:
        return True
    return any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
