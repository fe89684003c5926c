"""The reference's sample-file format (caf_rust/src/utils.rs)."""
from __future__ import annotations

from pathlib import Path

import numpy as np


def read_file_c64(filename) -> np.ndarray:
    """utils.rs:10-35: packed little-endian f32 I/Q pairs -> Complex64 (f64 pairs).
    A trailing partial sample panics in the reference (slice out of range)."""
    raw = Path(filename).read_bytes()
    if len(raw) % 8:
        raise ValueError("file length is not a multiple of 8 bytes (utils.rs:21 would panic)")
    return np.frombuffer(raw, dtype="<c8").astype(np.complex128)


def read_file_c64_f32(filename) -> np.ndarray:
    """Same file kept in its native complex64 (for the CAF_C64 path)."""
    raw = Path(filename).read_bytes()
    if len(raw) % 8:
        raise ValueError("file length is not a multiple of 8 bytes")
    return np.frombuffer(raw, dtype="<c8").copy()


def write_file_binary(samples, filename) -> None:
    """utils.rs:39-63: complex128 little-endian, numpy.fromfile(dtype=complex128) compatible."""
    np.asarray(samples, dtype="<c16").tofile(filename)


def load_files(needle_filename, haystack_filename):
    """tests/test.rs:319-331: haystack.resize(needle.len(), 0)."""
    needle = read_file_c64(needle_filename)
    hay = read_file_c64(haystack_filename)
    n = len(needle)
    if len(hay) >= n:
        hay = hay[:n].copy()
    else:
        hay = np.concatenate([hay, np.zeros(n - len(hay), dtype=np.complex128)])
    return needle, hay
