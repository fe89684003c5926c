#!/usr/bin/env python3
"""kernels_r32.hpp (measurement build, CAF_R32=1) against the product chain kernel and the numpy oracle on
BASELINE configs[3]'s shape (n = 32768 complex64), a few rows."""
import os
import sys
from pathlib import Path

import numpy as np

os.environ["CAF_R32"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import caf_cookoff_amd as caf  # noqa: E402
from caf_cookoff_amd.synth import make_pair  # noqa: E402
from oracle import caf_oracle as O  # noqa: E402

n = 32768
s0, s1, lag, fo = make_pair(n=n, seed=5, lag=777, foffset=-31.5, dtype=np.complex64)
fr = np.array([-40.0, -32.0, -31.5, -31.0, 0.0, 31.5, 977.25, 12.0, 13.0])
meng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
plan = meng.plan(n, fr, 48000, dtype="c64")
print("kernel:", plan.kernel_name)
plan.close()
surf, ridx, rval, pk = meng.surface_arrays(s0, s1, fr, 48000, dtype="c64")
eng = caf.Engine(0)
ref, ridx0, rval0, pk0 = eng.surface_arrays(s0, s1, fr, 48000, dtype="c64")
osurf, oidx, oval = O.np_caf_surface(s0.astype(np.complex128), s1.astype(np.complex128), fr, 48000)
print("vs oracle: r32 %.3e   chain %.3e   (of max)" % (np.max(np.abs(surf - osurf)) / osurf.max(), np.max(np.abs(ref - osurf)) / osurf.max()))
print("row argmax r32", ridx, "oracle", oidx, "peak", (pk.freq, pk.idx), "want", (-31.5, lag))
assert np.max(np.abs(surf - osurf)) <= 1e-3 * osurf.max() and (pk.freq, pk.idx) == (-31.5, lag)
assert np.array_equal(ridx, ridx0)
print("r32 ok")
