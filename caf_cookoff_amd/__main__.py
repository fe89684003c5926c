"""Command-line demo: the reference's caf_rust/src/main.rs:10-32 with its TODO done
("take in two c64 files as arguments", main.rs:1-2).

    python -m caf_cookoff_amd NEEDLE.c64 HAYSTACK.c64 [--start -100 --end 100 --step 0.5]
                              [--fs 48000] [--refine 0.05] [--dump-surf PATH --view rust|go|python]
"""
from __future__ import annotations

import argparse
import sys

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="caf_cookoff_amd", description=__doc__.splitlines()[0])
    ap.add_argument("needle")
    ap.add_argument("haystack")
    ap.add_argument("--start", type=float, default=-100.0)
    ap.add_argument("--end", type=float, default=100.0)
    ap.add_argument("--step", type=float, default=0.5)
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--refine", type=float, default=0.0, help="fine Doppler step for a second pass around the peak")
    ap.add_argument("--dump-surf", default=None, help="write the surface as raw little-endian f64 rows "
                    "(caf_go dump_surf / numpy.fromfile compatible)")
    ap.add_argument("--view", choices=["rust", "go", "python"], default="rust")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    from . import Engine, gen_float_shifts, load_files
    needle, haystack = load_files(a.needle, a.haystack)  # haystack.resize(needle.len()), main.rs:15
    shifts = gen_float_shifts(a.start, a.end, a.step)
    eng = Engine(a.device)
    want = a.dump_surf is not None
    surf, ridx, rval, peak = eng.surface_arrays(needle, haystack, shifts, a.fs, want_surface=want)
    freq, idx = float(peak.freq), int(peak.idx)
    if a.refine > 0:
        _, (freq, idx), _ = eng.refine_peak(needle, haystack, a.fs, shifts, a.refine)
    # main.rs:29-31
    print(f"Frequency offset: {freq:.1f}Hz" if a.refine <= 0 else f"Frequency offset: {freq:g}Hz")
    print(f"Time offset: {idx} samples ({idx / (a.fs / 1000.0):.3f}ms)")
    if want:
        out = surf if a.view == "rust" else eng.surface_view(surf, a.view)
        np.ascontiguousarray(out, dtype="<f8").tofile(a.dump_surf)
        print(f"wrote ({out.shape[0]}x{out.shape[1]}) surf to {a.dump_surf}", file=sys.stderr)
    eng.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
