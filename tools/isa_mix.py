#!/usr/bin/env python3
"""Instruction mix of a kernel in build/caf_api.gfx950.s (make -C caf_cookoff_amd/csrc asm).
usage: tools/isa_mix.py <mangled-name-substring> [...]"""
import collections
import re
import sys

path = "caf_cookoff_amd/csrc/build/caf_api.gfx950.s"
text = open(path).read().split("\n")
for want in sys.argv[1:]:
    start = next(i for i, l in enumerate(text) if l.startswith("_ZN") and want in l.split(":")[0] and l.split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    c = collections.Counter()
    for line in text[start + 1:end]:
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_pk"):
            c["v_pk"] += 1
        elif op.startswith("v_") and op.endswith("_f64"):
            c["v_f64"] += 1
        elif op.startswith("v_") and op.endswith("_f32"):
            c["v_f32"] += 1
        elif op.startswith("v_"):
            c["v_other"] += 1
            c["  " + op] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith(("buffer_", "global_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith("s_"):
            c["s_other"] += 1
    print(text[start].split(":")[0])
    valu = c["v_pk"] + c["v_f64"] + c["v_f32"] + c["v_other"]
    print(f"  static VALU {valu}: " + ", ".join(f"{k} {c[k]}" for k in ("v_f64", "v_f32", "v_pk", "v_other", "ds", "vmem", "s_waitcnt", "s_other")))
    print("  other VALU:", ", ".join(f"{k.strip()} {v}" for k, v in sorted(((k, v) for k, v in c.items() if k.startswith("  ")), key=lambda kv: -kv[1])[:10]))
