#!/usr/bin/env python3
"""tools/traffic_components.py SRC DST: turn the passes of tools/traffic_components.sh into DST/traffic_components.json --
the fabric bytes (L2 misses + write-backs: 2 x FETCH_SIZE + WRITE_SIZE, KiB, corrected as MI355X_MICROARCH.md prescribes) of
the configs[3] row kernel with nothing cut and with one buffer's accesses cut at a time; a buffer's share = full - cut."""
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
ROOT = Path(__file__).resolve().parent.parent
KERNEL = "k_chain_rows"
MASKS = {0: "nothing cut", 4: "no slab round trip", 16: "no surface stores", 2: "no haystack-spectrum loads", 8: "no needle loads",
         30: "no global memory at all"}


def mean(mask, counter):
    vals, durs = [], []
    for f in glob.glob(str(src / f"abl{mask}_{counter}" / "*" / "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    return (sum(vals) / len(vals), sum(durs) / len(durs)) if vals else (None, None)


rows = {}
for m, what in MASKS.items():
    (f, tf), (w, tw) = mean(m, "FETCH_SIZE"), mean(m, "WRITE_SIZE")
    if f is None or w is None:
        continue
    rows[m] = {"what": what, "read_bytes": 2.0 * f * 1024, "write_bytes": w * 1024, "fabric_bytes": (2.0 * f + w) * 1024,
               "kernel_ms_under_the_profiler": (tf + tw) / 2}
full = rows.get(0)
out = {"kernel": "caf::k_chain_rows<float, 14, 4, 1, 0>", "shape": "4096 x 65536 complex64 (BASELINE configs[3]), one launch",
       "algorithmic_bytes_per_launch": 1074348032, "masks": rows}
if full:
    comp = {}
    for m, key in ((4, "slab_round_trip"), (16, "surface_stores"), (2, "haystack_spectrum_loads"), (8, "needle_loads")):
        if m in rows:
            comp[key] = {"read_bytes": full["read_bytes"] - rows[m]["read_bytes"], "write_bytes": full["write_bytes"] - rows[m]["write_bytes"],
                         "fabric_bytes": full["fabric_bytes"] - rows[m]["fabric_bytes"]}
    if 30 in rows:
        comp["left_without_any_global_access (register spills to scratch, tables)"] = {
            "read_bytes": rows[30]["read_bytes"], "write_bytes": rows[30]["write_bytes"], "fabric_bytes": rows[30]["fabric_bytes"]}
    out["components"] = comp
    out["fabric_bytes_per_launch"] = full["fabric_bytes"]
    out["over_algorithmic"] = full["fabric_bytes"] / 1074348032
h = hashlib.sha256()
for n in ("cplx.hpp", "kernels_fused4096.hpp", "kernels_seq4096.hpp", "kernels_chain.hpp"):
    h.update(n.encode())
    h.update((ROOT / "caf_cookoff_amd" / "csrc" / n).read_bytes())
out["source_hash"] = h.hexdigest()[:16]
out["how"] = ("tools/traffic_components.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/chain_time.py 32768 4096 c64 1 "
              "on libcaf_hip_measure.so with CAF_CHAIN_ABL = the mask; FETCH_SIZE doubled (16-B-per-lane reads, MI355X_MICROARCH.md)")
dst.mkdir(parents=True, exist_ok=True)
(dst / "traffic_components.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
