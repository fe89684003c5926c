"""Host-side mirror of the reference's operator surface over the C ABI.

Reference interface (caf_rust/src/caf/mod.rs:17-66): a trait ``CafSurface`` with
associated functions ``caf_surface``, ``find_peak``, ``apply_freq_shift`` and the
row record ``CafSurfaceRow``; plus ``xcor_rustfft::Xcor::{new, run, clone}``
(xcor_rustfft.rs:14-93).  ``CafHip`` is the backend an eighth
``impl CafSurface for CafHip`` would be; everything here is marshalling -- the
arithmetic runs in the HIP kernels behind ``libcaf_hip.so``.
"""
from __future__ import annotations

import ctypes
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import CAF_C64, CAF_C128, CafError, CafPeak, check


def _as_c128(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex128)


def _as_c64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex64)


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _uptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


@dataclass
class CafSurfaceRow:
    """mod.rs:17-22."""
    freq: float
    xcor_mag: Optional[np.ndarray]  # |.|^2 over 2n lags (None if the surface was not requested)
    xcor_peak_idx: int
    xcor_peak_val: float


class Engine:
    """One ``caf_ctx``: a GPU, a stream, cached plans.  Not thread-safe.
    ``lib`` selects another build of the C-ABI library (``MEASURE_LIB_PATH`` for tools/)."""

    def __init__(self, device: int = 0, lib=None):
        self._h = None
        self._plans = weakref.WeakSet()
        self._host_bufs = weakref.WeakValueDictionary()   # address -> live host_empty() buffer: close() refuses while any of them is alive
        self._registered = {}                             # address -> array registered with host_register (kept alive until unregistered)
        self.lib = _lib.load(lib)
        h = ctypes.c_void_p()
        self._check(self.lib.caf_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def _check(self, rc: int):
        check(rc, self.lib)

    def close(self):
        """Closes every live Plan (and their Streams) first: caf_ctx_destroy frees the plans'
        device buffers, so their Python handles must not outlive it.  Refuses (RuntimeError) while an array of
        :meth:`host_empty` is still alive: caf_ctx_destroy frees that pinned memory, and a later read or write of the
        array would touch freed memory -- drop the arrays (and their views) first."""
        if getattr(self, "_h", None):
            alive = len(self._host_bufs)
            if alive:
                raise RuntimeError(f"Engine.close(): {alive} host_empty() array(s) still alive; delete them (and every "
                                   "view of them) before closing the engine that owns their memory")
            for p in list(self._plans):
                p.close()
            self._check(self.lib.caf_ctx_destroy(self._h))   # (unregisters what is still registered, before the arrays go)
            self._h = None
            self._registered.clear()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------
    def set_stream(self, hip_stream: Optional[int]):
        """Run on the given hipStream_t handle (0 = HIP's null stream = torch's default
        stream); ``None`` goes back to the context's private stream."""
        if hip_stream is None:
            self._check(self.lib.caf_ctx_reset_stream(self._h))
        else:
            self._check(self.lib.caf_ctx_set_stream(self._h, ctypes.c_void_p(int(hip_stream))))

    def synchronize(self):
        self._check(self.lib.caf_ctx_synchronize(self._h))

    def device_info(self) -> Tuple[int, str]:
        cu = ctypes.c_int()
        buf = ctypes.create_string_buffer(128)
        self._check(self.lib.caf_ctx_device_info(self._h, ctypes.byref(cu), buf, 128))
        return cu.value, buf.value.decode()

    # -- host memory the kernels may write in place ---------------------------------
    def host_empty(self, shape, dtype) -> np.ndarray:
        """``caf_host_alloc``: an uninitialised numpy array in pinned host memory of this context.
        Passed as ``out=`` to :meth:`surface_arrays` the row kernel stores the surface straight into it
        (no device-to-host copy after the kernel).  The memory lives until the array (and every view of
        it) is garbage-collected; the array keeps its Engine alive, and ``close()`` refuses while it exists."""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = ctypes.c_void_p()
        self._check(self.lib.caf_host_alloc(self._h, max(nbytes, 1), ctypes.byref(p)))
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        buf._caf_owner = self  # the buffer keeps the Engine alive: its context owns (and would free) this memory
        self._host_bufs[p.value] = buf
        lib, h, addr = self.lib, self._h, p.value

        def _free():   # (h stays valid: the Engine cannot be closed or collected while the buffer lives)
            lib.caf_host_free(h, ctypes.c_void_p(addr))
        weakref.finalize(buf, _free)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def host_register(self, arr: np.ndarray):
        """``caf_host_register``: page-lock a caller-owned array (~1.2 ms per 26 MB: once per buffer,
        not per call) so that surfaces can be written into it in place; undo with
        :meth:`host_unregister`.  Whole pages only (page-aligned address, size a multiple of the page size: an
        ``mmap`` buffer, not a heap block); the Engine keeps a reference to the array while it is registered."""
        self._check(self.lib.caf_host_register(self._h, ctypes.c_void_p(arr.ctypes.data), arr.nbytes))
        # the registered range must outlive its registration: the Engine holds the array until host_unregister (or close)
        self._registered[arr.ctypes.data] = arr

    def host_unregister(self, arr: np.ndarray):
        self._check(self.lib.caf_host_unregister(self._h, ctypes.c_void_p(arr.ctypes.data)))
        self._registered.pop(arr.ctypes.data, None)

    # -- a1 -----------------------------------------------------------------------
    def apply_freq_shift(self, samples, freq_shift: float, fs: int) -> np.ndarray:
        """mod.rs:46-65."""
        if np.asarray(samples).dtype == np.complex64:
            s = _as_c64(samples)
            out = np.empty_like(s)
            self._check(self.lib.caf_apply_freq_shift_c64(self._h, _fptr(s.view(np.float32)), len(s),
                                                    float(freq_shift), int(fs), _fptr(out.view(np.float32))))
            return out
        s = _as_c128(samples)
        out = np.empty_like(s)
        self._check(self.lib.caf_apply_freq_shift_c128(self._h, _dptr(s.view(np.float64)), len(s),
                                                 float(freq_shift), int(fs), _dptr(out.view(np.float64))))
        return out

    # -- a3 -----------------------------------------------------------------------
    def xcor(self, a, b) -> np.ndarray:
        """xcor_rustfft.rs:51-78.  Length mismatch asserts like :54-55."""
        if len(a) != len(b):
            raise AssertionError("assertion failed: b.len() == self.n")
        if np.asarray(a).dtype == np.complex64 and np.asarray(b).dtype == np.complex64:
            a, b = _as_c64(a), _as_c64(b)
            out = np.empty_like(a)
            self._check(self.lib.caf_xcor_c64(self._h, _fptr(a.view(np.float32)), _fptr(b.view(np.float32)), len(a),
                                        _fptr(out.view(np.float32))))
            return out
        a, b = _as_c128(a), _as_c128(b)
        out = np.empty_like(a)
        self._check(self.lib.caf_xcor_c128(self._h, _dptr(a.view(np.float64)), _dptr(b.view(np.float64)), len(a),
                                     _dptr(out.view(np.float64))))
        return out

    # -- a5 -----------------------------------------------------------------------
    def surface_arrays(self, needle, haystack, freqs_hz, fs: int, want_surface: bool = True,
                       dtype: str = "c128", out: Optional[np.ndarray] = None):
        """caf_surface as flat arrays: (surface[F,2n] | None, row_idx[F], row_val[F], CafPeak).
        ``out``: a C-contiguous [F, 2n] array of the dtype's real type to receive the surface (e.g. from
        :meth:`host_empty`: written in place by the row kernel) instead of a fresh ``np.empty``."""
        if len(needle) != len(haystack):
            # caf_surface itself does not check; Xcor::run's assert trips (xcor_rustfft.rs:54-55)
            raise AssertionError("assertion failed: a.len() == self.n")
        fr = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        F, n = len(fr), len(needle)
        ridx = np.zeros(F, dtype=np.uint64)
        peak = CafPeak()
        if dtype == "c64":
            nd, hs = _as_c64(needle), _as_c64(haystack)
            surf = self._surface_out(out, F, n, np.float32) if want_surface else None
            rval = np.zeros(F, dtype=np.float32)
            self._check(self.lib.caf_surface_c64(self._h, _fptr(nd.view(np.float32)), _fptr(hs.view(np.float32)), n,
                                           _dptr(fr), F, int(fs), _fptr(surf) if want_surface else None,
                                           _uptr(ridx), _fptr(rval), ctypes.byref(peak)))
        elif dtype == "c128":
            nd, hs = _as_c128(needle), _as_c128(haystack)
            surf = self._surface_out(out, F, n, np.float64) if want_surface else None
            rval = np.zeros(F, dtype=np.float64)
            self._check(self.lib.caf_surface_c128(self._h, _dptr(nd.view(np.float64)), _dptr(hs.view(np.float64)), n,
                                            _dptr(fr), F, int(fs), _dptr(surf) if want_surface else None,
                                            _uptr(ridx), _dptr(rval), ctypes.byref(peak)))
        else:
            raise ValueError("dtype must be 'c128' or 'c64'")
        return surf, ridx, rval, peak

    @staticmethod
    def _surface_out(out, F, n, rdt):
        if out is None:
            return np.empty((F, 2 * n), dtype=rdt)
        if out.shape != (F, 2 * n) or out.dtype != rdt or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous [{F}, {2 * n}] array of {np.dtype(rdt).name}")
        return out

    def caf_surface(self, needle, haystack, freqs_hz, fs: int, want_surface: bool = True,
                    dtype: str = "c128") -> List[CafSurfaceRow]:
        """mod.rs:26-27 -> Vec<CafSurfaceRow> in freq-list order (like CafRustFFT /
        the rayon collect, mod.rs:283-309)."""
        surf, ridx, rval, _ = self.surface_arrays(needle, haystack, freqs_hz, fs, want_surface, dtype)
        fr = np.asarray(freqs_hz, dtype=np.float64)
        return [CafSurfaceRow(float(fr[r]), surf[r] if surf is not None else None, int(ridx[r]), float(rval[r]))
                for r in range(len(fr))]

    # -- a7 -----------------------------------------------------------------------
    def find_peak(self, arr: Sequence[CafSurfaceRow]) -> Tuple[float, int]:
        """mod.rs:31-42 -> (freq, xcor_peak_idx); first strictly-greater row wins,
        an empty / all-zero surface gives (0.0, 0)."""
        F = len(arr)
        fr = np.array([r.freq for r in arr], dtype=np.float64)
        ri = np.array([r.xcor_peak_idx for r in arr], dtype=np.uint64)
        rv = np.array([r.xcor_peak_val for r in arr], dtype=np.float64)
        peak = CafPeak()
        self._check(self.lib.caf_find_peak(self._h, _dptr(fr), _uptr(ri), _dptr(rv), F, ctypes.byref(peak)))
        return float(peak.freq), int(peak.idx)

    def peak_exchange_stage(self, stage: int, d_peaks: int, count: int, d_red: int, d_freqs_all: int = 0, nfreq_all: int = 0,
                            d_out: int = 0):
        """``caf_peak_exchange_stage``: one of the three element-sized kernels around the two collectives of the global-peak
        exchange of row shards (device pointers; asynchronous on the engine's stream): see :class:`caf_cookoff_amd.dist.PeakExchange`."""
        self._check(self.lib.caf_peak_exchange_stage(self._h, int(stage), ctypes.c_void_p(d_peaks or None), int(count),
                                                     ctypes.c_void_p(d_red), ctypes.c_void_p(d_freqs_all or None), int(nfreq_all),
                                                     ctypes.c_void_p(d_out or None)))

    # -- views in the Go / Python implementations' conventions -----
    def surface_view(self, surface: np.ndarray, view: str) -> np.ndarray:
        """``view='go'``: caf_go amb_surf (2n lags, |.|, lag = n - k, caf.go:95-116, main.go:35);
        ``view='python'``: caf_python amb_surf (n lags of scipy 'same', |.|, tau = n/2 - i,
        caf.py:15-18,145)."""
        surf = np.ascontiguousarray(surface)
        rows, L = surf.shape
        n = L // 2
        code = {"go": _lib.CAF_VIEW_GO, "python": _lib.CAF_VIEW_PYTHON}[view]
        dt = CAF_C128 if surf.dtype == np.float64 else CAF_C64
        out = np.empty((rows, L if view == "go" else n), dtype=surf.dtype)
        self._check(self.lib.caf_surface_view(self._h, dt, ctypes.c_void_p(surf.ctypes.data), rows, n, code,
                                        ctypes.c_void_p(out.ctypes.data)))
        return out

    # -- coarse -> fine Doppler search -----------------------------
    def refine_peak(self, needle, haystack, fs: int, coarse_freqs, fine_step: float, dtype: str = "c128"):
        """Coarse grid first, then a fine grid of ``fine_step`` spanning one coarse bin either
        side of the coarse peak (the KAT grids of test.rs:174,212 are such fine grids).
        Returns ((coarse_freq, coarse_idx), (fine_freq, fine_idx), fine_freqs)."""
        cf = np.ascontiguousarray(coarse_freqs, dtype=np.float64)
        _, _, _, pk = self.surface_arrays(needle, haystack, cf, fs, want_surface=False, dtype=dtype)
        if pk.row < 0:
            return (0.0, 0), (0.0, 0), np.array([])
        r = int(pk.row)
        lo = cf[r - 1] if r > 0 else cf[r] - (cf[1] - cf[0] if len(cf) > 1 else fine_step)
        hi = cf[r + 1] if r + 1 < len(cf) else cf[r] + (cf[-1] - cf[-2] if len(cf) > 1 else fine_step)
        # the fine grid is built like gen_float_shifts (test.rs:335-352): integer milli-Hz / 1e3,
        # so its values are bit-identical to the reference's grids (32.15, not 32.150000000000006)
        st = int(round(fine_step * 1000.0))
        if st <= 0:
            raise ValueError("fine_step must be >= 0.001 Hz (milli-Hz grid, test.rs:341)")
        k0, k1 = int(round(lo * 1000.0 / st)), int(round(hi * 1000.0 / st))
        ff = np.array([(k * st) / 1e3 for k in range(k0, k1 + 1)], dtype=np.float64)
        _, _, _, pf = self.surface_arrays(needle, haystack, ff, fs, want_surface=False, dtype=dtype)
        return (float(pk.freq), int(pk.idx)), (float(pf.freq), int(pf.idx)), ff

    # -- device-resident plans ----------------------------------------------------
    def plan(self, n: int, freqs_hz, fs: int, dtype: str = "c128", row_begin: int = 0,
             row_end: Optional[int] = None) -> "Plan":
        return Plan(self, n, freqs_hz, fs, dtype, row_begin, row_end)


class Plan:
    """``caf_plan``: (n, freq list, fs, dtype[, row shard]) with device-resident tables.
    All pointers handed to :meth:`surface_dev` are DEVICE addresses (e.g.
    ``torch.Tensor.data_ptr()``)."""

    def __init__(self, eng: Engine, n: int, freqs_hz, fs: int, dtype: str = "c128", row_begin: int = 0,
                 row_end: Optional[int] = None):
        self.eng = eng
        fr = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        if row_end is None:
            row_end = len(fr)
        self.dtype = dtype
        dt = {"c128": CAF_C128, "c64": CAF_C64}[dtype]
        h = ctypes.c_void_p()
        eng._check(eng.lib.caf_plan_create(eng._h, int(n), _dptr(fr), len(fr), int(fs), dt, int(row_begin),
                                      int(row_end), ctypes.byref(h)))
        self._h = h
        self._streams = weakref.WeakSet()
        eng._plans.add(self)
        self.n, self.L = int(n), 2 * int(n)
        self.rows = int(eng.lib.caf_plan_rows(h))
        self.row_begin = int(row_begin)
        self.path = eng.lib.caf_plan_path(h).decode()
        self.kernel_name = eng.lib.caf_plan_kernel_name(h).decode()

    def surface_dev(self, d_needle: int, d_haystack: int, batch: int, d_surface: Optional[int], d_row_idx: int,
                    d_row_val: int, d_peak: int):
        self.eng._check(self.eng.lib.caf_surface_dev(self._h, ctypes.c_void_p(d_needle), ctypes.c_void_p(d_haystack),
                                           int(batch), ctypes.c_void_p(d_surface or 0), ctypes.c_void_p(d_row_idx),
                                           ctypes.c_void_p(d_row_val), ctypes.c_void_p(d_peak)))

    def timing_begin(self):
        self.eng._check(self.eng.lib.caf_plan_timing_begin(self._h))

    def timing_end(self) -> Tuple[float, int]:
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        self.eng._check(self.eng.lib.caf_plan_timing_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, int(n.value)

    def close(self):
        """Closes the plan's Streams first (their graphs hold the plan's device buffers)."""
        if getattr(self, "_h", None) and getattr(self.eng, "_h", None):
            for st in list(self._streams):
                st.close()
            self.eng._check(self.eng.lib.caf_plan_destroy(self._h))
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """``caf_stream``: double-buffered streaming of host-resident pairs (BASELINE configs[4]).
    ``buffers(slot)`` are numpy views of the slot's PINNED staging memory; fill them, then
    ``submit(slot)`` (asynchronous graph replay) and later ``wait(slot)``.  The views keep the
    Stream object alive, but an explicit ``close()`` frees the pinned memory under them."""

    PEAK_DTYPE = np.dtype([("val", "<f8"), ("freq", "<f8"), ("idx", "<u8"), ("row", "<i8")])

    def __init__(self, plan: "Plan", batch: int, nslots: int = 2, want_surface: bool = True, split: bool = False,
                 three_kernels: bool = False, two_kernels: bool = False, one_kernel: bool = False, memcpy_nodes: bool = False):
        """``split``: the slot's graph holds ``batch`` independent single-surface node chains
        (CAF_STREAM_SPLIT) instead of one batched chain.  Single-surface chains of n = 4096 plans are one
        launch or two kernel nodes, chosen by the number of surfaces in flight (include/caf_hip.h);
        ``one_kernel`` / ``two_kernels`` force either, ``three_kernels`` keeps the older
        {spectrum, rows, find_peak} chain (for comparison).  ``memcpy_nodes``: inputs and results cross PCIe as
        hipMemcpyAsync nodes of the graph (BASELINE configs[4] to the letter) instead of kernels reading / writing the
        mapped pinned buffers in place."""
        self.plan, self.batch, self.nslots = plan, int(batch), int(nslots)
        h = ctypes.c_void_p()
        plan.eng._check(plan.eng.lib.caf_stream_create_ex(plan._h, self.batch, self.nslots, int(bool(want_surface)),
                                                          (_lib.CAF_STREAM_SPLIT if split else 0)
                                                          | (_lib.CAF_STREAM_THREE_KERNELS if three_kernels else 0)
                                                          | (_lib.CAF_STREAM_TWO_KERNELS if two_kernels else 0)
                                                          | (_lib.CAF_STREAM_ONE_KERNEL if one_kernel else 0)
                                                          | (_lib.CAF_STREAM_MEMCPY_NODES if memcpy_nodes else 0),
                                                          ctypes.byref(h)))
        self._h = h
        plan._streams.add(self)
        self._cdt = np.complex128 if plan.dtype == "c128" else np.complex64
        self._rdt = np.float64 if plan.dtype == "c128" else np.float32

    def buffers(self, slot: int):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.plan.eng._check(self.plan.eng.lib.caf_stream_host_buffers(self._h, int(slot), ctypes.byref(a), ctypes.byref(b)))
        nbytes = self.batch * self.plan.n * np.dtype(self._cdt).itemsize

        def view(p):
            buf = (ctypes.c_char * nbytes).from_address(p.value)
            buf._caf_owner = self  # the view's base keeps the Stream (and its pinned memory) alive
            return np.frombuffer(buf, dtype=self._cdt).reshape(self.batch, self.plan.n)
        return view(a), view(b)

    def submit(self, slot: int):
        self.plan.eng._check(self.plan.eng.lib.caf_stream_submit(self._h, int(slot)))

    def wait(self, slot: int, want_rows: bool = True):
        peaks = np.zeros(self.batch, dtype=self.PEAK_DTYPE)
        ridx = np.zeros((self.batch, self.plan.rows), dtype=np.uint64) if want_rows else None
        rval = np.zeros((self.batch, self.plan.rows), dtype=self._rdt) if want_rows else None
        self.plan.eng._check(self.plan.eng.lib.caf_stream_wait(
            self._h, int(slot), peaks.ctypes.data_as(ctypes.POINTER(CafPeak)),
            _uptr(ridx) if want_rows else None, ctypes.c_void_p(rval.ctypes.data) if want_rows else None))
        return peaks, ridx, rval

    def run(self, needles, haystacks, want_rows: bool = False):
        """``caf_stream_run``: all ``count`` host-resident pairs through the slots in one native loop.
        needles / haystacks: [count][n] complex arrays of the plan's dtype (C-contiguous)."""
        nd = np.ascontiguousarray(needles, dtype=self._cdt)
        hs = np.ascontiguousarray(haystacks, dtype=self._cdt)
        if nd.ndim != 2 or nd.shape != hs.shape or nd.shape[1] != self.plan.n:
            raise ValueError("needles / haystacks must be [count][n]")
        count = nd.shape[0]
        peaks = np.zeros(count, dtype=self.PEAK_DTYPE)
        ridx = np.zeros((count, self.plan.rows), dtype=np.uint64) if want_rows else None
        rval = np.zeros((count, self.plan.rows), dtype=self._rdt) if want_rows else None
        self.plan.eng._check(self.plan.eng.lib.caf_stream_run(
            self._h, ctypes.c_void_p(nd.ctypes.data), ctypes.c_void_p(hs.ctypes.data), count,
            peaks.ctypes.data_as(ctypes.POINTER(CafPeak)), _uptr(ridx) if want_rows else None,
            ctypes.c_void_p(rval.ctypes.data) if want_rows else None))
        return peaks, ridx, rval

    def run_stats(self) -> dict:
        """Host-thread seconds of the last :meth:`run`: fill (memcpy into the pinned slots), launch, wait, collect."""
        a = (ctypes.c_double * 4)()
        self.plan.eng._check(self.plan.eng.lib.caf_stream_run_stats(self._h, a))
        return {"fill_s": a[0], "launch_s": a[1], "wait_s": a[2], "collect_s": a[3]}

    def surface_ptr(self, slot: int) -> int:
        return int(self.plan.eng.lib.caf_stream_surface(self._h, int(slot)) or 0)

    def close(self):
        if getattr(self, "_h", None) and getattr(self.plan, "_h", None):
            self.plan.eng.lib.caf_stream_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_stream_share(count: int, nworkers: int, worker: int) -> Tuple[int, int, int]:
    """``caf_multi_stream_share``: (first, stride, items) of the pairs worker ``worker`` of ``nworkers``
    handles out of ``count`` (round-robin).  Pure host arithmetic: works without a GPU."""
    a, b, c = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    lib = _lib.load()
    check(lib.caf_multi_stream_share(int(count), int(nworkers), int(worker), ctypes.byref(a), ctypes.byref(b),
                                     ctypes.byref(c)), lib)
    return a.value, b.value, c.value


class MultiStream:
    """``caf_multi_stream``: surface-parallel streaming over several devices (second
    decomposition): one context + plan + stream per entry of ``devices`` (ids may repeat), whole surfaces
    round-robin, results in input order, no collective."""

    PEAK_DTYPE = Stream.PEAK_DTYPE

    def __init__(self, devices: Sequence[int], n: int, freqs_hz, fs: int, dtype: str = "c128", nslots: int = 3,
                 want_surface: bool = False, lib=None):
        self.lib = _lib.load(lib)
        self._h = None
        fr = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        check(self.lib.caf_multi_stream_create(ids, len(devices), int(n), _dptr(fr), len(fr), int(fs),
                                               {"c128": CAF_C128, "c64": CAF_C64}[dtype], int(nslots), int(bool(want_surface)),
                                               ctypes.byref(h)),
              self.lib)
        self._h = h
        self.n, self.rows, self.ndev = int(n), len(fr), len(devices)
        self.devices = [int(d) for d in devices]
        self._cdt = np.complex128 if dtype == "c128" else np.complex64
        self._rdt = np.float64 if dtype == "c128" else np.float32

    def set_timeout(self, seconds: float):
        """``caf_multi_stream_set_timeout``: every later :meth:`run` fails with ``CafError`` (``CAF_ERR_TIMEOUT``) instead of
        waiting longer than ``seconds`` for a device; the object is unusable afterwards.  0 = no deadline (the default)."""
        check(self.lib.caf_multi_stream_set_timeout(self._h, float(seconds)), self.lib)

    def run(self, needles, haystacks, want_rows: bool = False):
        nd = np.ascontiguousarray(needles, dtype=self._cdt)
        hs = np.ascontiguousarray(haystacks, dtype=self._cdt)
        if nd.ndim != 2 or nd.shape != hs.shape or nd.shape[1] != self.n:
            raise ValueError("needles / haystacks must be [count][n]")
        count = nd.shape[0]
        peaks = np.zeros(count, dtype=self.PEAK_DTYPE)
        ridx = np.zeros((count, self.rows), dtype=np.uint64) if want_rows else None
        rval = np.zeros((count, self.rows), dtype=self._rdt) if want_rows else None
        check(self.lib.caf_multi_stream_run(self._h, ctypes.c_void_p(nd.ctypes.data), ctypes.c_void_p(hs.ctypes.data), count,
                                            peaks.ctypes.data_as(ctypes.POINTER(CafPeak)), _uptr(ridx) if want_rows else None,
                                            ctypes.c_void_p(rval.ctypes.data) if want_rows else None), self.lib)
        return peaks, ridx, rval

    def surface_ptr(self, worker: int, slot: int) -> int:
        """``caf_multi_stream_surface``: DEVICE address (on ``devices[worker]``) of the slot's slab
        [8][rows][2n]; 0 if created without surfaces."""
        return int(self.lib.caf_multi_stream_surface(self._h, int(worker), int(slot)) or 0)

    def locate(self, count: int, pair: int) -> Tuple[int, int, int, bool]:
        """``caf_multi_stream_locate``: (worker, slot, index in the slab, still resident) of pair ``pair`` after a
        run over ``count`` pairs."""
        w, s, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        i = ctypes.c_size_t()
        check(self.lib.caf_multi_stream_locate(self._h, int(count), int(pair), ctypes.byref(w), ctypes.byref(s), ctypes.byref(i),
                                               ctypes.byref(r)), self.lib)
        return w.value, s.value, i.value, bool(r.value)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.caf_multi_stream_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_surface_shard(nfreq: int, nworkers: int, worker: int) -> Tuple[int, int]:
    """``caf_multi_surface_shard``: the contiguous rows [begin, end) worker ``worker`` of ``nworkers`` owns.  Host
    arithmetic only (== :func:`caf_cookoff_amd.shifts.shard_range`)."""
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    lib = _lib.load()
    check(lib.caf_multi_surface_shard(int(nfreq), int(nworkers), int(worker), ctypes.byref(a), ctypes.byref(b)), lib)
    return a.value, b.value


def multi_surface_reduce(shard_peaks) -> np.ndarray:
    """``caf_multi_surface_reduce``: find_peak over shard peak records (structured array of ``Stream.PEAK_DTYPE``):
    largest value, then lowest global row; no GPU needed.  -> one record."""
    sp = np.ascontiguousarray(shard_peaks, dtype=Stream.PEAK_DTYPE)
    out = np.zeros(1, dtype=Stream.PEAK_DTYPE)
    lib = _lib.load()
    check(lib.caf_multi_surface_reduce(sp.ctypes.data_as(ctypes.POINTER(CafPeak)), len(sp),
                                       out.ctypes.data_as(ctypes.POINTER(CafPeak))), lib)
    return out[0]


class MultiSurface:
    """``caf_multi_surface``: the Doppler rows of ONE surface sharded over several devices behind one call -- the
    reference's threadpool fan-out and join (mod.rs:391-461) with GPUs as the workers (first
    decomposition).  ``devices`` may repeat an id (several contexts on one GPU); ``rccl=True`` joins the global peak
    with ncclAllReduce(max) + ncclAllReduce(min key) inside the process (distinct devices only) instead of on the host."""

    PEAK_DTYPE = Stream.PEAK_DTYPE

    def __init__(self, devices: Sequence[int], n: int, freqs_hz, fs: int, dtype: str = "c128", rccl: bool = False,
                 surface_on_device: bool = False, lib=None):
        self.lib = _lib.load(lib)
        self._h = None
        self.freqs = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        check(self.lib.caf_multi_surface_create(ids, len(devices), int(n), _dptr(self.freqs), len(self.freqs), int(fs),
                                                {"c128": CAF_C128, "c64": CAF_C64}[dtype],
                                                (_lib.CAF_MULTI_REDUCE_RCCL if rccl else 0)
                                                | (_lib.CAF_MULTI_SURFACE_ON_DEVICE if surface_on_device else 0),
                                                ctypes.byref(h)), self.lib)
        self._h = h
        self.n, self.rows, self.ndev, self.dtype = int(n), len(self.freqs), len(devices), dtype
        self.surface_on_device = bool(surface_on_device)
        self._cdt = np.complex128 if dtype == "c128" else np.complex64
        self._rdt = np.float64 if dtype == "c128" else np.float32
        self._host_bufs = weakref.WeakValueDictionary()  # address -> live host_empty() buffer (close() refuses under them)

    def set_timeout(self, seconds: float):
        """``caf_multi_surface_set_timeout``: every later :meth:`run` / :meth:`run_batch` fails with ``CafError`` (code
        ``CAF_ERR_TIMEOUT``) instead of waiting longer than ``seconds`` for a device; the object is unusable afterwards and
        :meth:`close` does not wait for the device that did not answer.  0 = no deadline (the default)."""
        check(self.lib.caf_multi_surface_set_timeout(self._h, float(seconds)), self.lib)

    def worker_info(self, worker: int):
        """-> (device, row_begin, row_end, row-kernel name)."""
        d, a, b = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_size_t()
        k = ctypes.c_char_p()
        check(self.lib.caf_multi_surface_worker_info(self._h, int(worker), ctypes.byref(d), ctypes.byref(a), ctypes.byref(b),
                                                     ctypes.byref(k)), self.lib)
        return d.value, a.value, b.value, (k.value or b"").decode()

    def slab_ptr(self, worker: int) -> int:
        """``caf_multi_surface_slab``: DEVICE address of the worker's rows (``surface_on_device=True``), else 0."""
        return int(self.lib.caf_multi_surface_slab(self._h, int(worker)) or 0)

    def host_empty(self, shape, dtype) -> np.ndarray:
        """``caf_multi_surface_host_alloc``: pinned memory EVERY worker writes in place.  The array keeps this object
        alive; the memory is released when the array (and every view of it) is gone, and :meth:`close` refuses before."""
        dt = np.dtype(dtype)
        nbytes = max(int(np.prod(shape)) * dt.itemsize, 1)
        p = ctypes.c_void_p()
        check(self.lib.caf_multi_surface_host_alloc(self._h, nbytes, ctypes.byref(p)), self.lib)
        buf = (ctypes.c_char * nbytes).from_address(p.value)
        buf._caf_owner = self  # the array keeps this object (and so the memory) alive
        self._host_bufs[p.value] = buf
        lib, h, addr = self.lib, self._h, p.value
        weakref.finalize(buf, lambda: lib.caf_multi_surface_host_free(h, ctypes.c_void_p(addr)))
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def run(self, needle, haystack, want_surface: Optional[bool] = None, out: Optional[np.ndarray] = None):
        """One surface -> (surface[F,2n] | None, row_idx[F], row_val[F], peak record).  ``want_surface`` defaults to True, and
        to False for an object created with ``surface_on_device=True`` (its rows stay in the workers' HBM: :meth:`slab_ptr`);
        asking such an object for a host surface raises ValueError."""
        if want_surface is None:
            want_surface = not self.surface_on_device
        if want_surface and self.surface_on_device:
            raise ValueError("this MultiSurface keeps the surface on the devices (surface_on_device=True): "
                             "call run(..., want_surface=False) and read slab_ptr(worker)")
        nd = np.ascontiguousarray(needle, dtype=self._cdt)
        hs = np.ascontiguousarray(haystack, dtype=self._cdt)
        if nd.shape != (self.n,) or hs.shape != (self.n,):
            raise AssertionError("assertion failed: a.len() == self.n")  # xcor_rustfft.rs:54-55
        F = self.rows
        surf = Engine._surface_out(out, F, self.n, self._rdt) if want_surface else None
        ridx = np.zeros(F, dtype=np.uint64)
        rval = np.zeros(F, dtype=self._rdt)
        peak = np.zeros(1, dtype=self.PEAK_DTYPE)
        check(self.lib.caf_multi_surface_run(self._h, ctypes.c_void_p(nd.ctypes.data), ctypes.c_void_p(hs.ctypes.data),
                                             ctypes.c_void_p(surf.ctypes.data) if want_surface else None, _uptr(ridx),
                                             ctypes.c_void_p(rval.ctypes.data), peak.ctypes.data_as(ctypes.POINTER(CafPeak))),
              self.lib)
        return surf, ridx, rval, peak[0]

    def run_batch(self, needles=None, haystacks=None, batch: Optional[int] = None, want_rows: bool = True):
        """``caf_multi_surface_run_batch``: B pairs in one call -> (row_idx[B,F] | None, row_val[B,F] | None, peaks[B]).
        needles / haystacks: [B][n] complex arrays; both None re-runs the ``batch`` pairs of the previous call, which are
        still in every worker's HBM.  Surfaces stay on the devices (``surface_on_device=True``: :meth:`batch_results`)."""
        if (needles is None) != (haystacks is None):
            raise ValueError("needles and haystacks must both be given or both be None")
        if needles is not None:
            nd = np.ascontiguousarray(needles, dtype=self._cdt)
            hs = np.ascontiguousarray(haystacks, dtype=self._cdt)
            if nd.ndim != 2 or nd.shape != hs.shape or nd.shape[1] != self.n:
                raise AssertionError("assertion failed: a.len() == self.n")  # xcor_rustfft.rs:54-55
            B = nd.shape[0]
            pn, ph = ctypes.c_void_p(nd.ctypes.data), ctypes.c_void_p(hs.ctypes.data)
        else:
            if batch is None:
                raise ValueError("batch= is needed to re-run the resident pairs")
            B, pn, ph = int(batch), None, None
        ridx = np.zeros((B, self.rows), dtype=np.uint64) if want_rows else None
        rval = np.zeros((B, self.rows), dtype=self._rdt) if want_rows else None
        peaks = np.zeros(B, dtype=self.PEAK_DTYPE)
        check(self.lib.caf_multi_surface_run_batch(self._h, pn, ph, B, _uptr(ridx) if want_rows else None,
                                                   ctypes.c_void_p(rval.ctypes.data) if want_rows else None,
                                                   peaks.ctypes.data_as(ctypes.POINTER(CafPeak))), self.lib)
        return ridx, rval, peaks

    def batch_results(self, worker: int):
        """``caf_multi_surface_batch_results`` -> dict(batch, slab, row_idx, row_val, peaks: DEVICE addresses on the worker's
        GPU (0 if absent); shard_peaks: the worker's find_peak records of the last batch as a structured array)."""
        b = ctypes.c_size_t()
        sl, ri, rv, pk, hp = (ctypes.c_void_p() for _ in range(5))
        check(self.lib.caf_multi_surface_batch_results(self._h, int(worker), ctypes.byref(b), ctypes.byref(sl), ctypes.byref(ri),
                                                       ctypes.byref(rv), ctypes.byref(pk), ctypes.byref(hp)), self.lib)
        shard = np.zeros(0, dtype=self.PEAK_DTYPE)
        if hp.value and b.value:
            shard = np.frombuffer((ctypes.c_char * (b.value * 32)).from_address(hp.value), dtype=self.PEAK_DTYPE).copy()
        return {"batch": b.value, "slab": sl.value or 0, "row_idx": ri.value or 0, "row_val": rv.value or 0,
                "peaks": pk.value or 0, "shard_peaks": shard}

    def run_stats(self):
        """Last run: ({'shards_s', 'reduce_s'}, shard peak records[ndev])."""
        a = (ctypes.c_double * 2)()
        sp = np.zeros(self.ndev, dtype=self.PEAK_DTYPE)
        check(self.lib.caf_multi_surface_run_stats(self._h, a, sp.ctypes.data_as(ctypes.POINTER(CafPeak))), self.lib)
        return {"shards_s": a[0], "reduce_s": a[1]}, sp

    def timing_begin(self):
        check(self.lib.caf_multi_surface_timing_begin(self._h), self.lib)

    def timing_end(self):
        """-> (kernel ms total per worker, launches per worker)."""
        ms = np.zeros(self.ndev, dtype=np.float64)
        nl = np.zeros(self.ndev, dtype=np.uint64)
        check(self.lib.caf_multi_surface_timing_end(self._h, _dptr(ms), _uptr(nl)), self.lib)
        return ms, nl

    def close(self):
        if getattr(self, "_h", None):
            if len(self._host_bufs):
                raise RuntimeError(f"MultiSurface.close(): {len(self._host_bufs)} host_empty() array(s) still alive; delete them "
                                   "(and every view of them) first: the object owns the pinned memory under them")
            self.lib.caf_multi_surface_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def debug_guard_bands(nbytes: int, lib=None) -> None:
    """``caf_debug_guard_bands``: red zones of ``nbytes`` (rounded up to 4 KiB pages) around every allocation the
    library makes from now on (process-wide; 0 = off)."""
    L = _lib.load(lib)
    check(L.caf_debug_guard_bands(int(nbytes)), L)


def debug_check_guards(lib=None) -> Tuple[int, int]:
    """``caf_debug_check_guards`` -> (allocations checked, violations); raises CafError naming the first damaged
    allocation if any red zone was written."""
    L = _lib.load(lib)
    a, b = ctypes.c_size_t(), ctypes.c_size_t()
    check(L.caf_debug_check_guards(ctypes.byref(a), ctypes.byref(b)), L)
    return a.value, b.value


_default: dict = {}


def default_engine(device: int = 0) -> Engine:
    e = _default.get(device)
    if e is None or e._h is None:
        e = _default[device] = Engine(device)
    return e


class CafHip:
    """``pub struct CafHip {}  impl CafSurface for CafHip`` -- associated functions
    without ``self`` like the reference's strategies (mod.rs:67-68,118-119)."""

    @staticmethod
    def caf_surface(needle, haystack, freqs_hz, fs: int) -> List[CafSurfaceRow]:
        return default_engine().caf_surface(needle, haystack, freqs_hz, fs)

    @staticmethod
    def find_peak(arr: Sequence[CafSurfaceRow]) -> Tuple[float, int]:
        return default_engine().find_peak(arr)

    @staticmethod
    def apply_freq_shift(samples, freq_shift: float, fs: int) -> np.ndarray:
        return default_engine().apply_freq_shift(samples, freq_shift, fs)


class Xcor:
    """xcor_rustfft::Xcor (xcor_rustfft.rs:14-93): ``new(n)``, ``run(a, b)``, ``clone()``.
    The FFT plan for n is cached in the engine's context, which is what ``new`` /
    ``clone`` buy in the reference."""

    def __init__(self, n: int, engine: Optional[Engine] = None):
        if n <= 0 or (n & (n - 1)):
            raise CafError(_lib.CAF_ERR_LENGTH, f"Xcor::new: n={n} is not a power of two")
        self.n = int(n)
        self.engine = engine or default_engine()

    def run(self, a, b) -> np.ndarray:
        if len(a) != self.n:
            raise AssertionError("assertion failed: a.len() == self.n")  # xcor_rustfft.rs:54
        if len(b) != self.n:
            raise AssertionError("assertion failed: b.len() == self.n")  # xcor_rustfft.rs:55
        return self.engine.xcor(a, b)

    def clone(self) -> "Xcor":
        return Xcor(self.n, self.engine)
