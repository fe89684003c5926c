#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch loads/stores counted between consecutive s_barrier
instructions of build/caf_api.gfx950.s.  usage: spill_map.py <mangled-name-prefix>"""
import re
import sys
from pathlib import Path

text = (Path(__file__).resolve().parent.parent / "caf_cookoff_amd/csrc/build/caf_api.gfx950.s").read_text().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith(sys.argv[1]))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
cnt, last = {}, 0
for i, l in enumerate(text[start:end]):
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op == "s_barrier":
        print(f"lines {last:6d}-{i:6d}: {cnt}")
        cnt, last = {}, i
    elif op.startswith("scratch_"):
        cnt[op] = cnt.get(op, 0) + 1
print(f"lines {last:6d}-{end - start:6d}: {cnt}")
