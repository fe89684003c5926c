#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of a streaming run: kernel durations and the gaps between
consecutive kernels of one slot's chain.  usage: stream_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"{k:62s} n={len(v):5d} median {v[len(v) // 2] / 1e3:7.2f} us  min {v[0] / 1e3:7.2f}  max {v[-1] / 1e3:7.2f}")
# timeline of 24 consecutive kernels in steady state
mid = rows[len(rows) // 2: len(rows) // 2 + 24]
t0 = int(mid[0]["Start_Timestamp"])
for r in mid:
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.2f} us .. +{(int(r['End_Timestamp']) - t0) / 1e3:8.2f} us  q{r.get('Queue_Id', '?'):>3s}  "
          f"{r['Kernel_Name'].split('(')[0][:50]}")
