#!/usr/bin/env python3
"""Generate tests/golden/py_amb_surf.npz by RUNNING the reference's own Python CAF.

Build container only (/root/reference does not exist on the GPU box; only the .npz travels).
caf_python/caf.py imports `numba`, which is not installed here; its decorators only wrap
functions the plain `amb_surf` (caf.py:89-117) does not call, so a pass-through stand-in module
named `numba` is put on sys.path for the import.  Nothing
of the reference's source is copied: the fixture holds inputs' file names and OUTPUT values.

What is stored (the (400 x 4096) float64 output is 13 MB, so a sample of it):
  tau, freq            the reference's own answer on its __main__ pair (caf.py:126-130,144-146)
  rows                 full rows FULL_ROWS of the surface
  strided              every STRIDE-th element of the flattened surface
  row_max, row_argmax  per-row maxima
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF_DIR = Path("/root/reference/caf_python")
FULL_ROWS = [0, 199, 365, 366, 367, 399]
STRIDE = 97


def main():
    if not (REF_DIR / "caf.py").exists():
        sys.exit("reference not present; the fixture is committed, nothing to do")
    with tempfile.TemporaryDirectory() as tmp:
        (Path(tmp) / "numba.py").write_text(
            "def _passthrough(*a, **k):\n"
            "    if len(a) == 1 and callable(a[0]) and not k:\n"
            "        return a[0]\n"
            "    return lambda f: f\n"
            "jit = njit = _passthrough\n")
        sys.path.insert(0, tmp)
        sys.path.insert(0, str(REF_DIR))
        import caf as ref_caf  # the reference's module, imported in place
    needle_name, hay_name = "chirp_4_raw.c64", "chirp_4_T+70samp_F+82.89Hz.c64"  # caf.py:126-127
    needle = np.fromfile(HERE / "data" / needle_name, dtype=np.complex64)
    haystack = np.fromfile(HERE / "data" / hay_name, dtype=np.complex64)[0:4096]  # caf.py:130
    samp_rate = 48e3
    freq_offsets = np.arange(-100, 100, 0.5)  # caf.py:133
    surf = ref_caf.amb_surf(needle, haystack, freq_offsets, samp_rate)
    fmax, tmax = np.unravel_index(surf.argmax(), surf.shape)  # caf.py:144-146
    tau_max = len(needle) // 2 - tmax
    freq_max = freq_offsets[fmax]
    print(surf.shape, surf.dtype, "->", tau_max, freq_max, "peak", surf.max())
    assert (tau_max, freq_max) == (70, 83.0)
    np.savez_compressed(
        HERE / "py_amb_surf.npz", needle=np.array(needle_name), haystack=np.array(hay_name),
        freqs=freq_offsets, samp_rate=np.array(samp_rate), tau=np.array(tau_max), freq=np.array(freq_max),
        full_rows=np.array(FULL_ROWS), rows=surf[FULL_ROWS], stride=np.array(STRIDE),
        strided=surf.reshape(-1)[::STRIDE].copy(), row_max=surf.max(axis=1), row_argmax=surf.argmax(axis=1),
        shape=np.array(surf.shape))


if __name__ == "__main__":
    main()
