"""bench_extras.py -- everything bench.py measures BESIDE the headline: the CPU comparator (the C restatement of caf_rust, timed on
this box's host cores), the live issue ceilings, the other BASELINE configs of the N = 1 line (configs[2], [3], [4], the literal
host-pointer calls), the in-process multi-device measurements and the two multi-GPU extras of an N > 1 line.  bench.py keeps the
contract: arguments, launch, the timed region, the one JSON line."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from bench_common import (FS, HBM_PEAK_GBS, N_SAMP, ROOT, Case, algorithmic_bytes, host_cpu_info, roofline_entry, secondary_entry,
                          traffic_detail, traffic_fields)


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


def issue_ceiling(torch, dev, case, env, steps=10):
    """Issue ceiling of a shipped row kernel, measured live: the measurement library's arithmetic-only
    ablation of the SAME kernel (wrong results, timing only) on the same batch and buffers.
      n = 4096 kernels   CAF_STORE_MODE=33: k_seq_rows / k_duo_rows with LDS traffic, barriers, global loads
                         and stores removed -> what remains is the VALU instruction stream of the row
      chain kernels      CAF_CHAIN_ABL=31 (configs[3]): no global memory, no workgroup barriers; the LDS
                         exchanges of the chain stay (without them the values would have to live in registers
                         and the kernel spills: HISTORY.md section 5)
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


def compiled_host_bench(batch=256, calls=10):
    """examples/caf_bench.cpp (built as tests/cpp/caf_bench): the reference's bench loop (benches/caf_bench.rs:150-168) from a
    COMPILED host over the C ABI -- the literal one-call-per-iteration loop, the peaks-only call, and the loop as one
    caf_multi_surface_run_batch call per `batch` pairs (join: the in-library RCCL exchange when more than one GPU is visible,
    the host join on one -- loading the system librccl and creating a communicator costs a cold child process ~5 s of a
    default run, and `in_process_headline` already takes the one-rank RCCL join).  A child process without Python; it checks
    its own answers.  Reported, never `value` (its batched figure from HBM is the same measurement as `value`, one level down)."""
    exe = ROOT / "tests" / "cpp" / "caf_bench"
    try:
        if not exe.exists():
            subprocess.run(["make", "-C", str(exe.parent), "caf_bench"], check=True, capture_output=True, timeout=300)
        r = subprocess.run([str(exe), str(ROOT / "tests" / "golden" / "data"), str(batch), str(calls)], capture_output=True, text=True,
                           timeout=300, cwd=str(ROOT))
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout).strip()[-300:]}
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out["how"] = "tests/cpp/caf_bench (C++, child process, system HIP + librccl): examples/caf_bench.cpp"
        return out
    except Exception as e:  # reported, never fatal for the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def stream_run(plan, nd, hs, lags, total, nslots, batch, split, three_kernels=False, native=True, passes=5, memcpy_nodes=False):
    """`total` surfaces through a caf_stream.  native: the whole loop is one caf_stream_run call (fill the
    slot's pinned buffers, replay its graph, retire the oldest slot -- in C++, as a compiled host would);
    otherwise the same loop step by step from Python (submit / wait per slot), which adds ~10 us of
    interpreter time to every step.  -> (surfaces/s, us per surface, 'ok/total')."""
    import numpy as np
    import caf_cookoff_amd as caf
    pool_n = len(lags)
    st = caf.Stream(plan, batch=batch, nslots=nslots, want_surface=True, split=split, three_kernels=three_kernels,
                    memcpy_nodes=memcpy_nodes)
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


STREAM_FIXED_FORM = "batched20_2slots"
STREAM_FORMS = (   # name, slots, surfaces per replay, split, three kernels, native loop, hipMemcpyAsync nodes
    ("single_2slots", 2, 1, False, False, True, False), ("single_3slots", 3, 1, False, False, True, False),
    ("single_4slots", 4, 1, False, False, True, False),
    ("split4_2slots", 2, 4, True, False, True, False), ("batched4_2slots", 2, 4, False, False, True, False),
    ("batched8_4slots", 4, 8, False, False, True, False),
    ("batched20_2slots", 2, 20, False, False, True, False),
    ("batched32_2slots", 2, 32, False, False, True, False),
    ("batched20_2slots_memcpy_nodes", 2, 20, False, False, True, True),
    ("batched8_4slots_memcpy_nodes", 4, 8, False, False, True, True),
    ("batched1_2slots_memcpy_nodes", 2, 1, False, False, True, True),
    ("single_2slots_three_kernels", 2, 1, False, True, True, False),
    ("single_2slots_python_loop", 2, 1, False, False, False, False))
STREAM_DEFAULT_FORMS = (STREAM_FIXED_FORM, STREAM_FIXED_FORM + "_memcpy_nodes")


def stream_case(eng, torch, freqs, total=1000, sweeps=False):
    """BASELINE configs[4]: `total` back-to-back 400x8192 complex128 surfaces from host memory, double-
    buffered across pinned slots, one hipGraph replay per slot; surfaces stay on the device, (tau, f) + row
    peaks come back.  Sustained surfaces/s over the whole run, H2D and D2H included.
    A default run measures TWO forms:
      batched20_2slots               `value`: the FIXED form -- one batched chain of 20 surfaces per graph replay on two slots
                                     (the plain double buffer of the config text; 1000 surfaces = exactly 50 replays), the
                                     kernels read and write mapped pinned host memory in place
      batched20_2slots_memcpy_nodes  `memcpy_nodes_value`: BASELINE configs[4] TO THE LETTER -- the same form with the inputs
                                     crossing PCIe as hipMemcpyAsync (copy-engine) nodes of the slot's graph into device buffers
                                     and the results coming back as hipMemcpyAsync nodes (CAF_STREAM_MEMCPY_NODES)
    `--sweeps` adds the comparison forms (never selected from): single_2slots / _3slots / _4slots (one surface per replay),
    split4_2slots (four independent single-surface chains per replay), batched4_2slots / batched8_4slots / batched32_2slots,
    batched8_4slots_memcpy_nodes / batched1_2slots_memcpy_nodes, single_2slots_three_kernels (round-2a form), single_2slots_python_loop
    (submit / wait driven from Python), and `value_long_stream`: 32 per replay on two slots over 8 x `total` surfaces, the
    flattest form for streams much longer than the config's 1000.
    How the fixed form was chosen -- ON THIS WORKLOAD, 1000 surfaces, which 20 divides exactly (profiles/README.md
    says so too): 8 surfaces are 3 200 rows = 6.25 rounds of the 512 persistent row workgroups (a seventh, quarter-full round per
    launch), 20 are 16.7; and the row launch of a replay of 16 surfaces or more leaves 32 workgroup slots free, so that the other
    slot's staging + spectrum launch runs beside it instead of in its tail (HISTORY.md R5.6).  tools/stream_sweep.py 1000,
    profiles/r05_stream/stream_1000.txt (median of 9 passes, each form visited twice): 20 x 2 slots 64.5-64.7 k, 25 x 2
    64.0-64.3 k, 32 x 2 62.1-62.2 k (its last replay is three quarters padding), 8 x 4 61.6-62.5 k; for long streams 32 x 2 is the
    flattest form (profiles/r05_stream/form_stability.txt: 64.2-64.6 k over six creations)."""
    from caf_cookoff_amd.synth import make_batch
    plan = eng.plan(N_SAMP, freqs, FS)
    nd, hs, lags, _ = make_batch(64, N_SAMP, FS, seed0=5000)
    forms = {}
    for name, nslots, batch, split, three, native, mc in STREAM_FORMS:
        if not sweeps and name not in STREAM_DEFAULT_FORMS:
            continue
        v, us, okc = stream_run(plan, nd, hs, lags, total, nslots, batch, split, three, native, memcpy_nodes=mc)
        forms[name] = {"value": v, "us_per_surface": us, "tau_correct": okc}
        if native:
            forms[name]["host_thread"] = dict(getattr(stream_run, "last_host_us", {}))
            forms[name].update(getattr(stream_run, "last_spread", {}))
    long_stream = None
    if sweeps:
        v, us, okc = stream_run(plan, nd, hs, lags, 8 * total, 2, 32, False, False, True, passes=3)
        long_stream = {"value": v, "form": "batched32_2slots", "surfaces": 8 * total, "tau_correct": okc}
    plan.close()
    abytes = algorithmic_bytes(1, 400, N_SAMP, "c128")
    best = STREAM_FIXED_FORM
    return {"workload": f"{total} back-to-back 400x8192 complex128 surfaces from host memory, hipGraph replay per slot, "
                        "stage-in of inputs and stage-out of peaks included, surfaces left on the device (BASELINE configs[4]); "
                        "`value`: the kernels read and write mapped pinned host memory in place; `memcpy_nodes_value`: the literal "
                        "form of the config text (hipMemcpyAsync nodes both ways)",
            "value": forms[best]["value"], "unit": "surfaces/s", "form": best, "forms": forms,
            "form_selected_on": f"this workload ({total} surfaces: 20 divides it exactly); value_long_stream (--sweeps) is the steady-state form",
            "value_is": f"median of {forms[best].get('passes')} passes over the {total} pairs (one more pass before them warms up)",
            "value_min": forms[best].get("value_min"), "value_max": forms[best].get("value_max"),
            "memcpy_nodes_value": forms[best + "_memcpy_nodes"]["value"],
            "value_long_stream": long_stream,
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


def in_process_config3(devices, steps, warmup, forms=("host_join",), check=True, timeout_s=None):
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
            if timeout_s:
                ms.set_timeout(timeout_s)
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
    # one surface per step here: with 512 rows per rank at N = 8 the row launch takes 0.2 ms, so ~0.08 ms of tensor operations
    # around the two all-reduces would be 40 % of the step -- the fused exchange (three library kernels) matters most here
    fused = args.peak_reduce == "fused" and not rehearse
    method = "fused" if fused else ("allreduce" if args.peak_reduce == "fused" else args.peak_reduce)

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

    from caf_cookoff_amd.dist import PeakExchange
    px = PeakExchange(eng, 1, f3, dev) if fused else None

    def step3():
        c.launch()
        if px is not None:
            return PeakExchange.peaks_of(px(c.peak, always_collective=True))
        pk = c.peak.cpu() if rehearse else c.peak
        pki = pk.view(torch.int64)
        return reduce_global_peak(pk[:, 0], pki[:, 3], pki[:, 2], method=method, always_collective=True)

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
                    f"({hi - lo} rows on rank 0) + RCCL peak reduction ({method}) (BASELINE configs[3])",
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
    st = caf.Stream(plan, batch=20, nslots=2, want_surface=True)   # the fixed form of extra.configs4_stream
    st.run(a[:40], b[:40])   # warm the graphs
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
                    f"{world} ranks ({items} on rank 0), one caf_stream (20 surfaces per replay, 2 slots) per rank, no collective on the data path "
                    "(BASELINE configs[4], surface-parallel decomposition)",
        "value": total / best, "unit": "surfaces/s", "elapsed_ms_max_over_ranks": best * 1e3,
        "tau_correct": f"{okc}/{total}"}
    return out


def in_process_headline(devices, batch_per_gpu, steps, warmup, dtype="c128", forms=("host_join",), blocks=0, check=True,
                        timeout_s=None, with_upload=False):
    """BASELINE configs[1] (configs[2] with dtype c64) as a compiled host reaches it: ONE process, the C ABI only.
    caf_multi_surface_run_batch over G = len(devices) workers: a step covers B = batch_per_gpu * G surfaces (weak scaling: the
    per-GPU work is constant), worker r computes its Doppler rows [r*F/G, (r+1)*F/G) of EVERY surface in ONE launch of the row
    kernel and keeps its slab [B][rows][2n] in its own HBM; the B global peaks are joined
      host_join   on the host from the G shard records per surface (caf_multi_surface_reduce)
      rccl_join   by ONE grouped ncclAllReduce(max) over the B shard values + ONE ncclAllReduce(min key) per call, inside the
                  library, on the workers' streams (distinct devices only).
    The inputs are uploaded (replicated on every worker) BEFORE the timed region and re-run from HBM (needles = haystacks =
    NULL), as the contract prescribes; `with_upload` (--sweeps) is the same call with the host-to-device copy of all B pairs
    inside it (PCIe-inclusive: reported, never `value`).  `timeout_s`: the library's own deadline for every call
    (caf_multi_surface_set_timeout: a device that does not answer fails the call with CAF_ERR_TIMEOUT instead of hanging it).
    -> {form: {value surfaces/s, ms_per_step, per-worker row-kernel ms, join seconds, blocks, ...}}."""
    import statistics
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    G = len(devices)
    B = batch_per_gpu * G
    freqs = caf.bench_shifts()
    cdt = np.complex128 if dtype == "c128" else np.complex64
    nd, hs, lags, fos = make_batch(B, N_SAMP, FS, seed0=1000, dtype=cdt)
    want_f = np.array([freqs[np.argmin(np.abs(freqs - f))] for f in fos])
    out = {}
    for form in forms:
        ms = caf.MultiSurface(devices, N_SAMP, freqs, FS, dtype=dtype, rccl=(form == "rccl_join"), surface_on_device=True)
        try:
            if timeout_s:
                ms.set_timeout(timeout_s)
            ms.run_batch(nd, hs, want_rows=False)                       # first call: allocations + upload
            with_upload_s = None
            if with_upload:
                t0 = time.perf_counter()
                ms.run_batch(nd, hs, want_rows=False)                   # steady-state call WITH the upload of all B pairs
                with_upload_s = time.perf_counter() - t0
            pk = None
            for _ in range(max(1, warmup)):
                _, _, pk = ms.run_batch(batch=B, want_rows=False)
            ok = bool(np.array_equal(pk["idx"], np.asarray(lags)) and np.all(np.abs(pk["freq"] - want_f) <= 0.5 + 1e-9))
            if check:
                assert ok, f"in-process headline ({form}): the planted (tau, f) of the batch were not all found"
            ms.timing_begin()
            t0 = time.perf_counter()
            for _ in range(steps):
                ms.run_batch(batch=B, want_rows=False)
            el = time.perf_counter() - t0
            kms, nl = ms.timing_end()
            stats, _ = ms.run_stats()
            blocks_ms = []
            for _ in range(max(0, blocks)):
                tb = time.perf_counter()
                for _ in range(steps):
                    ms.run_batch(batch=B, want_rows=False)
                blocks_ms.append((time.perf_counter() - tb) / steps * 1e3)
            info = [ms.worker_info(w) for w in range(G)]
            per_worker = [float(k) / max(1, int(c)) for k, c in zip(kms, nl)]
            rows0 = info[0][2] - info[0][1]
            ab0 = algorithmic_bytes(B, rows0, N_SAMP, dtype)
            out[form] = {"value": B * steps / el, "unit": "surfaces/s", "ms_per_step": el / steps * 1e3, "steps": steps,
                         "elapsed_s": el, "surfaces_per_step": B, "devices": list(devices),
                         "rows_per_worker": [i[2] - i[1] for i in info], "worker_devices": [i[0] for i in info],
                         "worker_kernel_ms": per_worker, "launches_timed": int(nl[0]), "kernel": info[0][3],
                         "worker0_algorithmic_bytes": ab0,
                         "worker0_frac": roofline_entry(ab0, per_worker[0])["frac"] if per_worker[0] > 0 else None,
                         "last_call": {"shards_ms": stats["shards_s"] * 1e3, "join_ms": stats["reduce_s"] * 1e3},
                         "with_upload": {"ms_per_step": with_upload_s * 1e3, "value": B / with_upload_s,
                                         "what": f"the same call with the host-to-device copy of all {B} pairs to every worker inside it "
                                                 "(PCIe-inclusive; never `value`)"} if with_upload_s else None,
                         "blocks_ms": blocks_ms,
                         "blocks_median_ms": statistics.median(blocks_ms) if blocks_ms else None,
                         "planted_peaks_found": ok,
                         "surfaces": "kept on the devices (one slab [B][rows][2n] per worker)"}
        finally:
            ms.close()
    return out


def n1_extras(args, eng, torch, dev, local_rank, freqs, K, extra):
    """The other BASELINE configs, same process, after the headline (N = 1 only): fills `extra` in place."""
    import numpy as np
    import caf_cookoff_amd as caf
    torch.cuda.empty_cache()
    legs, cur = {}, [None, time.perf_counter()]

    def leg(name):
        """wall seconds of every leg of the extras (full record: extra.extras_leg_seconds)"""
        now = time.perf_counter()
        if cur[0] is not None:
            legs[cur[0]] = round(legs.get(cur[0], 0.0) + now - cur[1], 3)
        cur[0], cur[1] = name, now

    def plan_case(name, n, freqs_x, dtype, batch, lo_x, hi_x, steps, warmup, cfg, ceiling=None):
        leg(name)
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
            det = traffic_detail(c.plan.kernel_name, batch, dtype, ab)
            if det:
                extra[name]["traffic_detail"] = det
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
        plan_case("configs3_c64_full", 32768, f3, "c64", 1, 0, 4096, 10, 3,
                  "4096x65536 complex64 filterbank CAF, all rows on ONE GPU (BASELINE configs[3] shape)",
                  ceiling=("valu_plus_lds_exchanges", {"CAF_CHAIN_ABL": "31"},
                           "k_chain_rows<float, 14, 4> without global memory and workgroup barriers "
                           "(libcaf_hip_measure.so, CAF_CHAIN_ABL=31), same rows"))
        lo3, hi3 = caf.shard_range(4096, 3, 8)
        plan_case("configs3_c64_shard", 32768, f3, "c64", 1, lo3, hi3, 10, 2,
                  "rows [1536,2048) of 4096x65536 complex64: the shard rank 3 of 8 GPUs computes (BASELINE configs[3])")
        leg("configs4_stream")
        extra["configs4_stream"] = stream_case(eng, torch, freqs, total=1000, sweeps=args.sweeps)
        if args.sweeps:
            extra["configs4_stream"]["multi_ctx2_same_gpu"] = multi_stream_case(freqs, [local_rank, local_rank], total=1000)
    except Exception as e:
        extra["error"] = f"{type(e).__name__}: {e}"
    # (child processes, timed from C / C++: before the in-process leg below, whose 6.7 GB of slab allocations and frees would
    #  otherwise sit between them and the streaming measurements they have always followed)
    leg("host_api")
    extra["host_api"] = host_api_times()
    leg("compiled_host_bench")
    extra["compiled_host_bench"] = compiled_host_bench(batch=args.batch)
    if args.sweeps:
        leg("in_process_multi")
        try:
            extra["in_process_multi"] = in_process_config3([local_rank], steps=10, warmup=2, forms=("host_join", "rccl_join"))
            extra["in_process_multi"]["two_contexts_same_gpu"] = in_process_config3(
                [local_rank, local_rank], steps=10, warmup=2, forms=("host_join",))["host_join"]
        except Exception as e:
            extra["in_process_multi"] = {"error": f"{type(e).__name__}: {e}"}
    leg("in_process_headline")
    try:
        # the headline itself through the C ABI's batched row-shard call (what `bench.py --in-process` times), RCCL join with one
        # rank (--sweeps: the host join and the call with the upload inside beside it); if RCCL cannot be used on this box (not
        # loadable, no communicator, the join runs into the deadline) the host join is measured instead and the record says why
        import caf_cookoff_amd as caf_
        kw = dict(steps=max(5, min(K, 30)), warmup=5, check=not args.no_check, timeout_s=60.0, with_upload=args.sweeps)
        try:
            extra["in_process_headline"] = in_process_headline([local_rank], args.batch, forms=("rccl_join", "host_join") if args.sweeps
                                                               else ("rccl_join",), **kw)
        except caf_.CafError as e:
            if e.code not in (caf_._lib.CAF_ERR_RCCL, caf_._lib.CAF_ERR_TIMEOUT):
                raise
            extra["in_process_headline"] = in_process_headline([local_rank], args.batch, forms=("host_join",), **kw)
            extra["in_process_headline"]["rccl_join"] = {"error": str(e)}
    except Exception as e:
        extra["in_process_headline"] = {"error": f"{type(e).__name__}: {e}"}
    leg(None)
    extra["extras_leg_seconds"] = legs
