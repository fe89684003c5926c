"""Shared helpers of the -m gpu tests (imported by tests/test_gpu_*.py; the fixtures `eng`, `meng` and `pinned_copies` live in
conftest.py).  Test infrastructure only."""
import mmap
from pathlib import Path

import numpy as np

from conftest import DATA

ROOT = Path(__file__).resolve().parent.parent
FS = 48000
TOL64 = 1e-6   # of the surface maximum, complex128 (BASELINE north_star: 1e-6 relative for f64)
TOL32 = 1e-3   # complex64 (BASELINE configs[2])
PAGE = mmap.PAGESIZE


def _kats():
    from oracle import caf_oracle as O
    return O.KATS


def _pair(oracle, k):
    return oracle.load_pair(DATA, f"chirp_{k}_raw.c64", oracle.KATS[k][1])


def _plan_arrays(plan, eng, nd, hs, dtype):
    import torch
    import caf_cookoff_amd as caf
    cdt, tdt = (np.complex128, torch.float64) if dtype == "c128" else (np.complex64, torch.float32)
    d_nd = torch.from_numpy(nd.astype(cdt)[None]).cuda()
    d_hs = torch.from_numpy(hs.astype(cdt)[None]).cuda()
    surf = torch.empty((1, plan.rows, 8192), dtype=tdt, device="cuda")
    ridx = torch.empty((1, plan.rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((1, plan.rows), dtype=tdt, device="cuda")
    peak = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(d_nd.data_ptr(), d_hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                     peak.data_ptr())
    eng.synchronize()
    pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
    return surf[0].cpu().numpy(), ridx[0].cpu().numpy(), rval[0].cpu().numpy().astype(np.float64), pk


def _plan_arrays_n(plan, eng, nd, hs, dtype, n):
    import torch
    import caf_cookoff_amd as caf
    tdt = torch.float64 if dtype == "c128" else torch.float32
    d_nd, d_hs = torch.from_numpy(nd[None]).cuda(), torch.from_numpy(hs[None]).cuda()
    surf = torch.empty((1, plan.rows, 2 * n), dtype=tdt, device="cuda")
    ridx = torch.empty((1, plan.rows), dtype=torch.int64, device="cuda")
    rval = torch.empty((1, plan.rows), dtype=tdt, device="cuda")
    peak = torch.empty((1, 4), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    plan.surface_dev(d_nd.data_ptr(), d_hs.data_ptr(), 1, surf.data_ptr(), ridx.data_ptr(), rval.data_ptr(),
                     peak.data_ptr())
    eng.synchronize()
    pk = peak.cpu().numpy().view(caf.Stream.PEAK_DTYPE)[0, 0]
    return surf[0].cpu().numpy().astype(np.float64), ridx[0].cpu().numpy(), rval[0].cpu().numpy(), pk


def _mmap_array(shape, dtype, fill=0.0):
    """A caller-owned buffer for caf_host_register that is its own anonymous mapping: page-aligned, a whole number of pages
    long, and returned to the kernel (munmap, which tears down every GPU mapping of the range) when the array dies -- not a
    heap block that goes back to malloc and is handed out again, still mapped, as somebody else's copy destination.
    caf_host_register takes whole pages only (include/caf_hip.h): the shape must fill a page multiple."""
    count = int(np.prod(shape))
    nbytes = max(count * np.dtype(dtype).itemsize, 1)
    m = mmap.mmap(-1, (nbytes + PAGE - 1) // PAGE * PAGE)
    a = np.frombuffer(m, dtype=dtype, count=count).reshape(shape)
    a[...] = fill
    return a


def _guard_rows(row_bytes):
    """Rows of `row_bytes` that make up at least one whole page: a fence of that many rows either side of a registered block of
    rows keeps the block page-aligned inside an _mmap_array (row_bytes must divide the page size or be a multiple of it)."""
    assert PAGE % row_bytes == 0 or row_bytes % PAGE == 0, row_bytes
    return max(1, PAGE // row_bytes)


def _planted(rng, n, fs, f, lag, cdt=np.complex128):
    """haystack = needle delayed by `lag` (lag < 0: advanced, the peak lands at 2n + lag) and shifted by f."""
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n) if n >= 8 else \
        (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    y = np.roll(x, lag) * np.exp(2j * np.pi * f * np.arange(n) / fs)
    if lag >= 0:
        y[:lag] = 0
    else:
        y[lag:] = 0
    y = y + 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(cdt), y.astype(cdt)


def _dev_view(ptr, shape, typestr):
    import torch

    class _Dev:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Dev(), device="cuda")
