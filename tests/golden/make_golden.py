#!/usr/bin/env python3
"""Generate tests/golden/golden.npz from the numpy restatement in oracle/.

Vectors:
  * bench configs (chirp_0 and chirp_4, 400 shifts -100..99.5 Hz, fs 48000):
    per-row (idx, val), global (freq, idx), full rows {0,199,200,338,339,399},
    every 97th element of the flattened surface
  * the ten KAT configs: per-row (idx, val) + expected global answer
  * apply_freq_shift / xcor input+output vectors at N in {8, 64, 4096}
The numpy restatement is itself pinned by the ten reference KATs
(tests/test_oracle_kats.py); the reference pins no surface values.
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import caf_oracle as O  # noqa: E402

FS = 48000
FULL_ROWS = [0, 199, 200, 338, 339, 399]
STRIDE = 97


def main():
    dd = HERE / "data"
    out = {}
    manifest = {"fs": FS, "full_rows": FULL_ROWS, "stride": STRIDE, "bench": {}, "kats": {}}
    hay = {k: hf for k, hf, _, _ in O.KATS}
    shifts = O.bench_shifts()
    out["bench_freqs"] = shifts
    for k in (0, 4):
        nd, hs = O.load_pair(dd, f"chirp_{k}_raw.c64", hay[k])
        surf, ridx, rval = O.np_caf_surface(nd, hs, shifts, FS)
        bf, bi = O.np_find_peak(shifts, ridx, rval)
        out[f"bench{k}_row_idx"] = ridx
        out[f"bench{k}_row_val"] = rval
        out[f"bench{k}_rows"] = surf[FULL_ROWS]
        out[f"bench{k}_strided"] = surf.reshape(-1)[::STRIDE].copy()
        order = np.argsort(-rval, kind="stable")
        manifest["bench"][str(k)] = {
            "needle": f"chirp_{k}_raw.c64", "haystack": hay[k], "nfreq": len(shifts),
            "best_freq": bf, "best_idx": bi, "peak": float(rval.max()),
            "runner_up_row": int(order[1]), "runner_up_val": float(rval[order[1]]),
            "surface_max": float(surf.max()),
        }
    for k, hf, (s, e, st), exp in O.KATS:
        nd, hs = O.load_pair(dd, f"chirp_{k}_raw.c64", hf)
        fr = O.gen_float_shifts(s, e, st)
        _, ridx, rval = O.np_caf_surface(nd, hs, fr, FS, want_surface=False)
        bf, bi = O.np_find_peak(fr, ridx, rval)
        assert (bf, bi) == exp, (k, bf, bi, exp)
        out[f"kat{k}_row_idx"] = ridx
        out[f"kat{k}_row_val"] = rval
        manifest["kats"][str(k)] = {"needle": f"chirp_{k}_raw.c64", "haystack": hf,
                                    "shifts": [s, e, st], "nfreq": len(fr),
                                    "best_freq": exp[0], "best_idx": exp[1]}
    rng = np.random.default_rng(20261003)
    nd0, hs0 = O.load_pair(dd, "chirp_0_raw.c64", hay[0])
    for n in (8, 64, 4096):
        if n == 4096:
            a, b = nd0, hs0
        else:
            a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        out[f"vec{n}_a"] = a
        out[f"vec{n}_b"] = b
        out[f"vec{n}_shift_77p77"] = O.np_apply_freq_shift(a, 77.77, FS)  # caf_bench.rs:172-173
        out[f"vec{n}_shift_m12p5"] = O.np_apply_freq_shift(a, -12.5, FS)
        out[f"vec{n}_xcor"] = O.np_xcor(a, b)
    np.savez_compressed(HERE / "golden.npz", **out)
    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1) + "\n")
    print("wrote", HERE / "golden.npz", (HERE / "golden.npz").stat().st_size, "bytes")
    print(json.dumps(manifest["bench"], indent=1))


if __name__ == "__main__":
    main()
