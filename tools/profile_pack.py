#!/usr/bin/env python3
"""Pack a gpurun_out/prof_* directory (rocprofv3 csv output of bench.py passes) into a
profiles/<name>/ directory: kernel_stats.csv, pmc_summary.txt, traffic.json."""
import csv
import glob
import json
import shutil
import subprocess
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
kernel = sys.argv[3] if len(sys.argv) > 3 else "k_seq_rows"
dst.mkdir(parents=True, exist_ok=True)
for f in glob.glob(str(src / "stats" / "*" / "*kernel_stats.csv")):
    shutil.copy(f, dst / "kernel_stats.csv")
if (src / "stats_bench.json").exists():
    shutil.copy(src / "stats_bench.json", dst / "bench_under_rocprof.json")
# steady-state timing of the dominant kernel from the per-dispatch trace: the --stats average includes the warm-up launches
# (cold caches, first-touch of the output pages), so the median of the launches AFTER the warm-up is stored beside it
bench_line = json.loads((src / "stats_bench.json").read_text()) if (src / "stats_bench.json").exists() else {}
durs = []
for f in glob.glob(str(src / "stats" / "*" / "*kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
if durs:
    import statistics
    warm = min(int(bench_line.get("warmup", 0)), max(0, len(durs) - 1))
    steady = durs[warm:]
    timing = {"kernel": kernel, "calls": len(durs), "warmup_calls_dropped": warm,
              "steady_state_median_ms": statistics.median(steady), "steady_state_min_ms": min(steady),
              "steady_state_max_ms": max(steady), "average_all_calls_ms": sum(durs) / len(durs),
              "note": "rocprofv3 --kernel-trace per-dispatch durations of the same command; kernel_stats.csv holds the "
                      "tool's own average over ALL calls (warm-up included)"}
    (dst / "kernel_timing.json").write_text(json.dumps(timing, indent=1) + "\n")
    print(json.dumps(timing))
files = sorted(glob.glob(str(src / "pmc_*" / "*" / "*counter_collection.csv")))
out = subprocess.run([sys.executable, str(Path(__file__).parent / "pmc_summary.py"), *files], capture_output=True,
                     text=True).stdout.replace(str(src) + "/", "")
(dst / "pmc_summary.txt").write_text(out)


def mean_counter(pattern, counter):
    vals = []
    for f in glob.glob(str(src / pattern / "*" / "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


fetch, write = mean_counter("pmc_fetch", "FETCH_SIZE"), mean_counter("pmc_write", "WRITE_SIZE")
bench = json.loads((src / "stats_bench.json").read_text()) if (src / "stats_bench.json").exists() else {}
if fetch is not None and write is not None:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    # half of the bytes of 16-B-per-lane streaming reads -> doubled; WRITE_SIZE is exact for
    # 16-B-per-lane stores.
    traffic = (2.0 * fetch + write) * 1024.0
    info = {"kernel": kernel, "source_hash": bench.get("config", {}).get("kernel_source_hash"),
            "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "fetch_correction": 2.0,
            "traffic_bytes_per_launch": traffic,
            "surfaces_per_launch": bench.get("config", {}).get("surfaces_per_step"),
            "dtype": bench.get("dtype"), "algorithmic_bytes_per_launch":
                bench.get("roofline", {}).get("algorithmic_bytes_per_launch")}
    # The L2's memory-side (fabric) request counters by size and destination, when the pmc_ea_* passes were taken.  They say how
    # much leaves L2 and that it is all bound for the memory side (DRAM, as opposed to GMI / IO); whether the Infinity Cache
    # or an HBM channel then serves a request is invisible from here: rocprofv3 lists no MALL and no UMC counter on gfx950.
    ea = {c: mean_counter(pat, c) for pat, cs in (("pmc_ea_rd", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_BUBBLE_sum")),
                                                  ("pmc_ea_wr", ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum")),
                                                  ("pmc_l2", ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))) for c in cs}
    if ea.get("TCC_EA0_RDREQ_sum") is not None and ea.get("TCC_EA0_WRREQ_sum") is not None:
        rd, rd32, rdd, bub = (ea[k] or 0.0 for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_BUBBLE_sum"))
        wr, wr64, wrd = (ea[k] or 0.0 for k in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum"))
        info["fabric"] = {
            "counters": ea,
            # rocprofv3's own derived-metric expression for read bytes: 128-B "bubble" requests, 64-B and 32-B requests
            "read_bytes_by_request_size": bub * 128 + (rd - bub - rd32) * 64 + rd32 * 32,
            "write_bytes_by_request_size": wr64 * 64 + (wr - wr64) * 32,
            "read_requests_bound_for_dram": rdd / rd if rd else None,
            "write_requests_bound_for_dram": wrd / wr if wr else None,
            "l2_hit_rate": (ea["TCC_HIT_sum"] / (ea["TCC_HIT_sum"] + ea["TCC_MISS_sum"])) if ea.get("TCC_HIT_sum") is not None and
                           (ea["TCC_HIT_sum"] + ea["TCC_MISS_sum"]) else None}
    info["hbm_bytes_per_launch"] = None
    info["hbm_split"] = ("not exposed: rocprofv3 --list-avail on gfx950 (profiles/r05_counters/) offers the blocks SQ, SQC, TA, TD, TCP, TCC, "
                         "TCA, SPI, GRBM, CPC, CPF only -- no MALL / Infinity-Cache hit counter and no UMC (HBM channel) counter; "
                         "TCC_EA0_*_DRAM count requests BOUND for the memory side, which the Infinity Cache serves or passes on "
                         "invisibly.  traffic_bytes_per_launch is therefore fabric traffic (L2 misses and write-backs); see "
                         "traffic_components.json, where one exists, for which buffers it consists of")
    (dst / "traffic.json").write_text(json.dumps(info, indent=1) + "\n")
    print(json.dumps(info))
