#!/usr/bin/env python3
"""How busy do the row kernels keep the GPU in a streaming run?  From a rocprofv3 --kernel-trace CSV: the second half of the
run (steady state): wall span, the union of the row-kernel intervals (time in which at least one row kernel runs), the time in
which two or more overlap, the idle gaps between row kernels and what runs in them.  usage: stream_busy.py <kernel_trace.csv> [row-kernel substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "k_seq_rows"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"]]
t0, t1 = iv[0][0], max(e for _, e in iv)
ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
busy1 = busy2 = 0
depth, last = 0, t0
gaps = []
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    if depth == 0 and t > last: gaps.append((last, t))
    depth += d
    last = t
span = t1 - t0
durs = sorted(e - s for s, e in iv)
print(f"{len(iv)} row kernels over {span / 1e3:.0f} us: median duration {durs[len(durs) // 2] / 1e3:.1f} us (min {durs[0] / 1e3:.1f}, max {durs[-1] / 1e3:.1f}); sum of durations {sum(durs) / 1e3:.0f} us")
print(f"at least one row kernel running {busy1 / span * 100:.1f} % of the span, two or more {busy2 / span * 100:.1f} %, none {100 - busy1 / span * 100:.1f} %")
if gaps:
    g = sorted(b - a for a, b in gaps)
    print(f"{len(gaps)} gaps without a row kernel: median {g[len(g) // 2] / 1e3:.1f} us, max {g[-1] / 1e3:.1f} us, total {sum(g) / 1e3:.0f} us")
    a, b = max(gaps, key=lambda x: x[1] - x[0])
    print("kernels overlapping the longest gap:")
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > a - 20000 and s < b + 20000:
            print(f"   {(s - a) / 1e3:+8.1f} .. {(e - a) / 1e3:+8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][:60]}")
