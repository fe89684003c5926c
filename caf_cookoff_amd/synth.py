"""Seeded synthetic s0/s1 pairs shaped like the reference's fixtures.

Own generator (not a copy of utils/generate.py): same recipe at a high level --
band-limited complex Gaussian noise under a Hann taper, swept in frequency, then a
search capture = the chirp delayed by a known lag, offset by a known Doppler, plus
weak noise (generate.py:22-39,55-66) -- but with numpy's Generator API, an FFT-domain
band limit instead of firwin/filtfilt, and any power-of-two length.
"""
from __future__ import annotations

import numpy as np


def make_pair(n: int = 4096, fs: float = 48000.0, seed: int = 0, lag: int | None = None,
              foffset: float | None = None, rel_bw: float = 0.02, sweep_hz: float = 5e3,
              noise: float = 1e-5, dtype=np.complex128):
    """Returns (s0, s1, lag, foffset): s1[k+lag] ~ s0[k]*e^{j2pi f k/fs}, both length n."""
    rng = np.random.default_rng(seed)
    if lag is None:
        lag = int(rng.integers(7, max(8, min(256, n // 4))))
    if foffset is None:
        foffset = float(rng.uniform(-100.0, 100.0))
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    X = np.fft.fft(x)
    f = np.fft.fftfreq(n)
    X[np.abs(f) > 0.5 * rel_bw * 4] = 0.0  # crude low-pass
    x = np.fft.ifft(X) * np.hanning(n)
    x = x / np.max(np.abs(x)) * 0.25
    t = np.arange(n) / fs
    shape = np.linspace(-1.0, 1.0, n) ** 2
    s0 = (x * np.exp(2j * np.pi * np.cumsum(shape * sweep_hz) / fs)).astype(np.complex64)
    cap = np.concatenate([np.zeros(lag, dtype=np.complex128), s0.astype(np.complex128)])[:n]
    cap = cap * np.exp(2j * np.pi * foffset * t)
    cap = cap + noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    s1 = cap.astype(np.complex64)
    return s0.astype(dtype), s1.astype(dtype), lag, foffset


def make_batch(batch: int, n: int = 4096, fs: float = 48000.0, seed0: int = 0, dtype=np.complex128):
    """`batch` distinct pairs -> (needles[batch,n], haystacks[batch,n], lags, foffsets)."""
    nd = np.empty((batch, n), dtype=dtype)
    hs = np.empty((batch, n), dtype=dtype)
    lags, fos = [], []
    for b in range(batch):
        s0, s1, lag, fo = make_pair(n, fs, seed0 + b, dtype=dtype)
        nd[b], hs[b] = s0, s1
        lags.append(lag)
        fos.append(fo)
    return nd, hs, lags, fos
