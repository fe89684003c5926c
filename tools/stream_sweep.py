#!/usr/bin/env python3
"""tools/stream_sweep.py -- every streaming (BASELINE configs[4]) measurement of rounds 2-5 as ONE parametrised tool
(it replaces fourteen one-off tools/stream_*.py scripts; the evidence they produced is under profiles/r0*_stream/).

  stream_sweep.py rates  [--preset NAME | FORM ...] [--count N] [--rounds R] [--mode alternate|each|recreate]
                         [--dtype c128|c64] [--python-loop] [--measure] [--reserve V ...]
      surfaces/s of caf_stream_run (or, with --python-loop, of submit / wait driven from Python with the per-step host
      times) for streaming forms.  FORM = batch:slots[:flags], flags any of  s split chains, 1 one kernel node, 2 two nodes,
      3 three nodes (round-2a chain), m hipMemcpyAsync nodes.
        --mode alternate  every form's stream is created first, then R rounds visit them in rotation (median [min .. max])
        --mode each       per form: create, warm, R passes, close; the list of forms is visited twice (A/B/A/B)
        --mode recreate   per form R creations in rotation, each the median of 5 passes: is a form's rate stable across
                          caf_stream objects (the runtime maps HIP streams onto a few hardware queues as it sees fit)?
        --reserve V ...   measurement library: CAF_STREAM_RESERVE = V at capture time (workgroup slots the persistent row
                          launch leaves free for the next slot's staging + spectrum launch), every form at every V
      presets (the runs behind the committed evidence):
        1000       replay sizes for exactly 1000 surfaces     (profiles/r05_stream/stream_1000.txt; how the fixed form was chosen)
        stability  forms created repeatedly in one process    (profiles/r05_stream/form_stability.txt)
        native     one / two / three kernel nodes, split      (profiles/r03_stream)
        batch      batch x slots grid of batched chains
        slots      single-surface chains vs slot count        (try GPU_MAX_HW_QUEUES=8 in the environment)
        fixed      the fixed form by itself, for rocprofv3 --kernel-trace --stats
        probe      --python-loop over batched / single / split forms: where a step spends host time
        reserve    --measure --reserve 0 8 16 32 64 0 16 32 on the 8 x 4 form
  stream_sweep.py soak [count] [rounds]
      the forms whose cross-workgroup hand-offs are hand-made (k_seq_surface, one and two nodes): every row peak (index and
      value bits) and every caf_peak of every surface of every round must equal round 0's; exit status 1 otherwise.
  stream_sweep.py trace <kernel_trace.csv>            kernel durations + a 24-kernel steady-state timeline of a rocprofv3 trace
  stream_sweep.py busy  <kernel_trace.csv> [substr]   how busy the row kernels keep the GPU in the second half of a trace:
                                                      union of their intervals, overlap of two or more, the gaps and what runs in them
"""
import argparse
import csv
import os
import statistics
import sys
import time
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

PRESETS = {
    "1000": dict(forms="8:4 20:2 25:2 32:2 40:2 50:2 32:2:m 40:2:m 50:2:m 8:4:m", count=1000, rounds=9, mode="each"),
    "stability": dict(forms="1:4 8:4 16:4 24:2 32:2 32:3 48:2 64:2", count=2048, rounds=5, mode="recreate"),
    "native": dict(forms="1:2 1:3 1:4 1:2:2 1:3:2 1:4:2 1:2:3 1:3:3 4:2:s 4:2:s2 4:2:s3 4:2 16:2", count=2000, rounds=5, mode="alternate"),
    "batch": dict(forms="1:4 4:3 8:2 8:3 8:4 16:2 16:3 16:4 32:2 32:3 64:2 64:3", count=4096, rounds=5, mode="alternate"),
    "slots": dict(forms="1:2 1:3 1:4 1:5 1:6 1:8", count=4096, rounds=5, mode="each"),
    "fixed": dict(forms="20:2", count=2000, rounds=5, mode="alternate"),
    "probe": dict(forms="1:2 1:3 2:2:s 4:2:s 4:3:s 8:2:s 4:2 16:2", count=1000, rounds=2, mode="each", python_loop=True),
    "reserve": dict(forms="8:4", count=2048, rounds=7, mode="each", measure=True, reserve=[0, 8, 16, 32, 64, 0, 16, 32]),
}


def parse_form(text):
    parts = text.split(":")
    flags = parts[2] if len(parts) > 2 else ""
    return dict(batch=int(parts[0]), nslots=int(parts[1]), split="s" in flags, one_kernel="1" in flags, two_kernels="2" in flags,
                three_kernels="3" in flags, memcpy_nodes="m" in flags)


def form_name(f):
    tags = [t for t, on in (("split", f["split"]), ("one node", f["one_kernel"]), ("two nodes", f["two_kernels"]),
                            ("three nodes", f["three_kernels"]), ("memcpy nodes", f["memcpy_nodes"])) if on]
    return f"{f['batch']:3d} per replay x {f['nslots']} slots" + (" (" + ", ".join(tags) + ")" if tags else "")


def python_loop(st, f, nd, hs, lags, total):
    """submit / wait driven from Python, step by step -> (surfaces/s, per-step host times, correct, steps)"""
    batch, nslots, pool = f["batch"], f["nslots"], len(lags)
    bufs = [st.buffers(s) for s in range(nslots)]
    steps = max(nslots + 1, total // batch)
    for _ in range(2):
        tf = ts = tw = 0.0
        ok, infl = 0, []
        t0 = time.perf_counter()
        for step in range(steps):
            slot = step % nslots
            if len(infl) == nslots:
                a = time.perf_counter()
                s0, st0 = infl.pop(0)
                peaks, _, _ = st.wait(s0, want_rows=False)
                tw += time.perf_counter() - a
                ok += all(int(peaks[j]["idx"]) == lags[(st0 * batch + j) % pool] for j in range(batch))
            a = time.perf_counter()
            for j in range(batch):
                k = (step * batch + j) % pool
                bufs[slot][0][j], bufs[slot][1][j] = nd[k], hs[k]
            b = time.perf_counter()
            st.submit(slot)
            tf, ts = tf + b - a, ts + time.perf_counter() - b
            infl.append((slot, step))
        for s0, st0 in infl:
            peaks, _, _ = st.wait(s0, want_rows=False)
            ok += all(int(peaks[j]["idx"]) == lags[(st0 * batch + j) % pool] for j in range(batch))
        dt = time.perf_counter() - t0
    return steps * batch / dt, (tf / steps * 1e6, ts / steps * 1e6, tw / steps * 1e6, dt / steps * 1e6), ok, steps


def rates(args):
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    cfg = dict(PRESETS[args.preset]) if args.preset else {}
    forms = [parse_form(t) for t in (args.forms or cfg.get("forms", "20:2").split())]
    count = args.count or cfg.get("count", 1000)
    rounds = args.rounds or cfg.get("rounds", 5)
    mode = args.mode or cfg.get("mode", "alternate")
    use_python_loop = args.python_loop or cfg.get("python_loop", False)
    measure = args.measure or cfg.get("measure", False) or bool(args.reserve)
    reserve = args.reserve or cfg.get("reserve") or [None]
    cdt = np.complex128 if args.dtype == "c128" else np.complex64
    eng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH) if measure else caf.Engine(0)
    fr = caf.bench_shifts()
    nd64, hs64, lags64, _ = make_batch(64, 4096, 48000, seed0=5000, dtype=cdt)
    reps = (count + 63) // 64
    nd, hs = np.tile(nd64, (reps, 1))[:count], np.tile(hs64, (reps, 1))[:count]
    want = np.tile(np.asarray(lags64), reps)[:count]
    print(f"# {count} surfaces per pass, {rounds} rounds, mode {mode}, {args.dtype}, {'measurement' if measure else 'product'} library; "
          f"GPU_MAX_HW_QUEUES = {os.environ.get('GPU_MAX_HW_QUEUES', '(default)')}", flush=True)

    def make(plan, f):
        return caf.Stream(plan, want_surface=True, **f)

    def passes(st, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            pk, _, _ = st.run(nd, hs)
            ts.append(time.perf_counter() - t0)
        return ts, int(np.sum(pk["idx"] == want))

    def report(f, ts, ok, tag="", st=None):
        line = (f"{form_name(f)}{tag}: median {count / statistics.median(ts):8.0f} surfaces/s [{count / max(ts):.0f} .. {count / min(ts):.0f}] "
                f"tau ok {ok}/{count}")
        if st is not None:
            h = st.run_stats()
            line += ("; host thread per surface: fill %.2f us, launch %.2f, wait %.2f, collect %.2f"
                     % tuple(h[k] / count * 1e6 for k in ("fill_s", "launch_s", "wait_s", "collect_s")))
        print(line, flush=True)

    for rv in reserve:
        tag = ""
        if rv is not None:
            os.environ["CAF_STREAM_RESERVE"] = str(rv)
            tag = f" reserve {rv:3d}"
        plan = eng.plan(4096, fr, 48000, dtype=args.dtype)
        if use_python_loop:
            for f in forms:
                st = make(plan, f)
                v, (tf, ts_, tw, tt), ok, steps = python_loop(st, f, nd64, hs64, lags64, count)
                print(f"{form_name(f)}{tag} [python loop]: {v:8.0f} surfaces/s; per step: fill {tf:.1f} us, submit {ts_:.1f}, wait {tw:.1f}, "
                      f"total {tt:.1f}; tau ok {ok}/{steps}", flush=True)
                st.close()
        elif mode == "alternate":
            streams = [make(plan, f) for f in forms]
            for st in streams:
                st.run(nd[:256], hs[:256])
            ts, oks = [[] for _ in forms], [0] * len(forms)
            for _ in range(rounds):
                for i, st in enumerate(streams):
                    t, oks[i] = passes(st, 1)
                    ts[i] += t
            for f, t, ok, st in zip(forms, ts, oks, streams):
                report(f, t, ok, tag, st)
                st.close()
        elif mode == "each":
            for visit in range(1 if rv is not None else 2):
                for f in forms:
                    st = make(plan, f)
                    st.run(nd, hs)
                    t, ok = passes(st, rounds)
                    report(f, t, ok, tag)
                    st.close()
        else:  # recreate
            med = {i: [] for i in range(len(forms))}
            for _ in range(rounds):
                for i, f in enumerate(forms):
                    st = make(plan, f)
                    st.run(nd[:128], hs[:128])
                    t, _ = passes(st, 5)
                    med[i].append(count / statistics.median(t))
                    st.close()
            for i, f in enumerate(forms):
                v = med[i]
                print(f"{form_name(f)}{tag}: per-creation medians (k surfaces/s): " + " ".join(f"{x / 1e3:5.1f}" for x in v)
                      + f"   min {min(v) / 1e3:.1f}  max {max(v) / 1e3:.1f}", flush=True)
        plan.close()
    return 0


def soak(args):
    import numpy as np
    import caf_cookoff_amd as caf
    from caf_cookoff_amd.synth import make_batch
    count, rounds = args.count, args.rounds
    eng = caf.Engine(0)
    nd16, hs16, lags16, _ = make_batch(16, 4096, 48000, seed0=7000)
    reps = (count + 15) // 16
    nd, hs = np.tile(nd16, (reps, 1))[:count], np.tile(hs16, (reps, 1))[:count]
    lags = np.tile(np.asarray(lags16), reps)[:count]
    bad = 0
    for dtype in ("c128", "c64"):
        plan = eng.plan(4096, caf.bench_shifts(), 48000, dtype=dtype)
        ref = None
        for name, kw in (("one node, 2 slots", dict(batch=1, nslots=2, one_kernel=True)),
                         ("one node, 3 slots", dict(batch=1, nslots=3, one_kernel=True)),
                         ("two nodes, 3 slots", dict(batch=1, nslots=3, two_kernels=True)),
                         ("two nodes, 4 chains x 2 slots", dict(batch=4, nslots=2, split=True, two_kernels=True)),
                         ("one node, 4 chains x 2 slots", dict(batch=4, nslots=2, split=True, one_kernel=True))):
            st = caf.Stream(plan, want_surface=True, **kw)
            t0 = time.perf_counter()
            nbad = 0
            for rnd in range(rounds):
                peaks, ridx, rval = st.run(nd, hs, want_rows=True)
                if ref is None:
                    ref = (peaks.copy(), ridx.copy(), rval.copy())
                    assert np.array_equal(peaks["idx"], lags), "reference round: wrong lags"
                    continue
                badmask = np.any(ridx != ref[1], axis=1) | np.any(rval != ref[2], axis=1) | (peaks != ref[0])
                nbad += int(np.sum(badmask))
                nsl = kw["nslots"] * kw["batch"]
                for k in np.nonzero(badmask)[0][:3]:
                    dr = np.nonzero((ridx[k] != ref[1][k]) | (rval[k] != ref[2][k]))[0]
                    prev = k - nsl  # the surface that used the same pinned words one replay earlier
                    stale = (prev >= 0 and len(dr) and np.array_equal(ridx[k][dr], ref[1][prev][dr])
                             and np.array_equal(rval[k][dr], ref[2][prev][dr]))
                    print(f"    round {rnd} surface {k}: {len(dr)} rows differ (first {dr[:6]}), peak record "
                          f"{'differs' if peaks[k] != ref[0][k] else 'equal'}; differing rows equal the previous occupant's values: "
                          f"{bool(stale)}; got {ridx[k][dr[:2]]}/{rval[k][dr[:2]]} want {ref[1][k][dr[:2]]}/{ref[2][k][dr[:2]]}", flush=True)
            dt = time.perf_counter() - t0
            print(f"{dtype} {name:30s}: {rounds * count} surfaces in {dt:.1f} s ({rounds * count / dt:.0f}/s), "
                  f"{nbad} surfaces differ from the first round", flush=True)
            bad += nbad
            st.close()
        plan.close()
    print("SOAK", "FAILED" if bad else "ok")
    return 1 if bad else 0


def _trace_rows(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def trace(args):
    rows = _trace_rows(args.csv)
    dur = defaultdict(list)
    for r in rows:
        dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in dur.items():
        v.sort()
        print(f"{k:62s} n={len(v):5d} median {v[len(v) // 2] / 1e3:7.2f} us  min {v[0] / 1e3:7.2f}  max {v[-1] / 1e3:7.2f}")
    mid = rows[len(rows) // 2: len(rows) // 2 + 24]   # 24 consecutive kernels in steady state
    t0 = int(mid[0]["Start_Timestamp"])
    for r in mid:
        print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.2f} us .. +{(int(r['End_Timestamp']) - t0) / 1e3:8.2f} us  q{r.get('Queue_Id', '?'):>3s}  "
              f"{r['Kernel_Name'].split('(')[0][:50]}")
    return 0


def busy(args):
    rows = _trace_rows(args.csv)
    rows = rows[len(rows) // 2:]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if args.key in r["Kernel_Name"]]
    t0, t1 = iv[0][0], max(e for _, e in iv)
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    busy1 = busy2 = 0
    depth, last, gaps = 0, t0, []
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth == 0 and t > last:
            gaps.append((last, t))
        depth += d
        last = t
    span = t1 - t0
    durs = sorted(e - s for s, e in iv)
    print(f"{len(iv)} row kernels over {span / 1e3:.0f} us: median duration {durs[len(durs) // 2] / 1e3:.1f} us (min {durs[0] / 1e3:.1f}, "
          f"max {durs[-1] / 1e3:.1f}); sum of durations {sum(durs) / 1e3:.0f} us")
    print(f"at least one row kernel running {busy1 / span * 100:.1f} % of the span, two or more {busy2 / span * 100:.1f} %, "
          f"none {100 - busy1 / span * 100:.1f} %")
    if gaps:
        g = sorted(b - a for a, b in gaps)
        print(f"{len(gaps)} gaps without a row kernel: median {g[len(g) // 2] / 1e3:.1f} us, max {g[-1] / 1e3:.1f} us, total {sum(g) / 1e3:.0f} us")
        a, b = max(gaps, key=lambda x: x[1] - x[0])
        print("kernels overlapping the longest gap:")
        for r in rows:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if e > a - 20000 and s < b + 20000:
                print(f"   {(s - a) / 1e3:+8.1f} .. {(e - a) / 1e3:+8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][:60]}")
    return 0


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("rates")
    r.add_argument("forms", nargs="*", help="batch:slots[:flags]")
    r.add_argument("--preset", choices=sorted(PRESETS))
    r.add_argument("--count", type=int)
    r.add_argument("--rounds", type=int)
    r.add_argument("--mode", choices=["alternate", "each", "recreate"])
    r.add_argument("--dtype", choices=["c128", "c64"], default="c128")
    r.add_argument("--python-loop", action="store_true")
    r.add_argument("--measure", action="store_true")
    r.add_argument("--reserve", type=int, nargs="+")
    r.set_defaults(fn=rates)
    s = sub.add_parser("soak")
    s.add_argument("count", type=int, nargs="?", default=4096)
    s.add_argument("rounds", type=int, nargs="?", default=20)
    s.set_defaults(fn=soak)
    t = sub.add_parser("trace")
    t.add_argument("csv")
    t.set_defaults(fn=trace)
    b = sub.add_parser("busy")
    b.add_argument("csv")
    b.add_argument("key", nargs="?", default="k_seq_rows")
    b.set_defaults(fn=busy)
    args = ap.parse_args()
    sys.exit(args.fn(args))


if __name__ == "__main__":
    main()
