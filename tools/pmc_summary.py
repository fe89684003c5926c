#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per
kernel (over dispatches), plus mean kernel duration from the same file."""
import csv
import sys
from collections import defaultdict


def main(paths):
    for path in paths:
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        print(f"== {path}")
        for k in acc:
            n = max(len(v) for v in acc[k].values())
            print(f"  {k}  dispatches={n}  mean_ns={sum(dur[k]) / len(dur[k]):.0f}")
            for c, v in acc[k].items():
                print(f"      {c:28s} mean={sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1:])
