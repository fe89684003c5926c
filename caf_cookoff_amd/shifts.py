"""Shift-list constructors of the reference's callers (host logic, no GPU)."""
from __future__ import annotations

import numpy as np


def gen_float_shifts(start: float, end: float, step: float) -> np.ndarray:
    """caf_rust/tests/test.rs:335-352: integer milli-Hz range, end-exclusive;
    ``as i32`` / ``as usize`` truncate toward zero; values are ``m as f64 / 1e3``."""
    s = int(start * 1000.0)
    e = int(end * 1000.0)
    st = int(step * 1000.0)
    if st <= 0:
        raise ValueError("step_by(0) panics in the reference")
    return np.array([m / 1e3 for m in range(s, e, st)], dtype=np.float64)


def bench_shifts() -> np.ndarray:
    """caf_rust/benches/caf_bench.rs:31-35 == main.rs:18-22: -100.0 .. 99.5 Hz, 400 rows."""
    return np.array([m / 1e3 for m in range(-100000, 100000, 500)], dtype=np.float64)


def shard_range(nfreq: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous row shard of rank `rank`: [r*F/G, (r+1)*F/G).
    Contiguity keeps 'first row wins' == min over global row positions."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * nfreq) // world, ((rank + 1) * nfreq) // world
