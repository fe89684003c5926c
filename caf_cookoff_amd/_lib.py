"""ctypes loader for libcaf_hip.so (the C ABI of include/caf_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C caf_cookoff_amd/csrc``.  There is no fallback: a missing library or a
missing GPU raises, it never silently computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libcaf_hip.so"
# -DCAF_MEASURE build: rejected kernel variants + ablation switches (tools/ and the variant tests only)
# (CAF_HIP_MEASURE_LIB: another build of it, e.g. the host-sanitizer one of tools/asan_host_run.sh)
MEASURE_LIB_PATH = Path(os.environ.get("CAF_HIP_MEASURE_LIB", _HERE / "libcaf_hip_measure.so"))

CAF_OK = 0
CAF_ERR_BAD_ARG = 1
CAF_ERR_LENGTH = 2
CAF_ERR_HIP = 3
CAF_ERR_NOMEM = 4
CAF_ERR_NO_DEVICE = 5
CAF_ERR_STATE = 6
CAF_ERR_RCCL = 7
CAF_ERR_TIMEOUT = 8

CAF_C128 = 0
CAF_C64 = 1
CAF_VIEW_GO = 1
CAF_VIEW_PYTHON = 2
CAF_STREAM_SPLIT = 1
CAF_STREAM_THREE_KERNELS = 2
CAF_STREAM_TWO_KERNELS = 4
CAF_STREAM_ONE_KERNEL = 8
CAF_STREAM_MEMCPY_NODES = 16
CAF_MULTI_REDUCE_RCCL = 1
CAF_MULTI_SURFACE_ON_DEVICE = 2


class CafPeak(ctypes.Structure):
    """``caf_peak`` of include/caf_hip.h (mod.rs:31-42 result + value/row)."""
    _fields_ = [("val", ctypes.c_double), ("freq", ctypes.c_double),
                ("idx", ctypes.c_uint64), ("row", ctypes.c_int64)]


class CafError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"caf_hip error {code}: {msg}")
        self.code = code


# every symbol include/caf_hip.h declares: (name, restype, argtypes)
_vp = ctypes.c_void_p
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_up = ctypes.POINTER(ctypes.c_uint64)
_pp = ctypes.POINTER(CafPeak)
_sz = ctypes.c_size_t
_u32 = ctypes.c_uint32
_int = ctypes.c_int
SYMBOLS = [
    ("caf_abi_version", _int, []),
    ("caf_last_error_string", ctypes.c_char_p, []),
    ("caf_device_count", _int, []),
    ("caf_ctx_create", _int, [_int, ctypes.POINTER(_vp)]),
    ("caf_ctx_destroy", _int, [_vp]),
    ("caf_ctx_set_stream", _int, [_vp, _vp]),
    ("caf_ctx_reset_stream", _int, [_vp]),
    ("caf_ctx_synchronize", _int, [_vp]),
    ("caf_ctx_device_info", _int, [_vp, ctypes.POINTER(_int), ctypes.c_char_p, _sz]),
    ("caf_apply_freq_shift_c128", _int, [_vp, _dp, _sz, ctypes.c_double, _u32, _dp]),
    ("caf_apply_freq_shift_c64", _int, [_vp, _fp, _sz, ctypes.c_double, _u32, _fp]),
    ("caf_xcor_c128", _int, [_vp, _dp, _dp, _sz, _dp]),
    ("caf_xcor_c64", _int, [_vp, _fp, _fp, _sz, _fp]),
    ("caf_surface_c128", _int, [_vp, _dp, _dp, _sz, _dp, _sz, _u32, _dp, _up, _dp, _pp]),
    ("caf_surface_c64", _int, [_vp, _fp, _fp, _sz, _dp, _sz, _u32, _fp, _up, _fp, _pp]),
    ("caf_host_alloc", _int, [_vp, _sz, ctypes.POINTER(_vp)]),
    ("caf_host_free", _int, [_vp, _vp]),
    ("caf_host_register", _int, [_vp, _vp, _sz]),
    ("caf_host_unregister", _int, [_vp, _vp]),
    ("caf_find_peak", _int, [_vp, _dp, _up, _dp, _sz, _pp]),
    ("caf_plan_create", _int, [_vp, _sz, _dp, _sz, _u32, _int, _sz, _sz, ctypes.POINTER(_vp)]),
    ("caf_plan_destroy", _int, [_vp]),
    ("caf_plan_path", ctypes.c_char_p, [_vp]),
    ("caf_plan_rows", _sz, [_vp]),
    ("caf_plan_kernel_name", ctypes.c_char_p, [_vp]),
    ("caf_surface_dev", _int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    ("caf_plan_timing_begin", _int, [_vp]),
    ("caf_plan_timing_end", _int, [_vp, _dp, _up]),
    ("caf_surface_view", _int, [_vp, _int, _vp, _sz, _sz, _int, _vp]),
    ("caf_stream_create", _int, [_vp, _sz, _int, _int, ctypes.POINTER(_vp)]),
    ("caf_stream_create_ex", _int, [_vp, _sz, _int, _int, ctypes.c_uint, ctypes.POINTER(_vp)]),
    ("caf_stream_destroy", _int, [_vp]),
    ("caf_stream_host_buffers", _int, [_vp, _int, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    ("caf_stream_run", _int, [_vp, _vp, _vp, _sz, _pp, _up, _vp]),
    ("caf_stream_run_stats", _int, [_vp, _dp]),
    ("caf_stream_submit", _int, [_vp, _int]),
    ("caf_stream_wait", _int, [_vp, _int, _pp, _up, _vp]),
    ("caf_stream_surface", _vp, [_vp, _int]),
    ("caf_multi_stream_share", _int, [_sz, _int, _int, ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    ("caf_multi_stream_create", _int, [ctypes.POINTER(_int), _int, _sz, _dp, _sz, _u32, _int, _int, _int, ctypes.POINTER(_vp)]),
    ("caf_multi_stream_devices", _int, [_vp]),
    ("caf_multi_stream_run", _int, [_vp, _vp, _vp, _sz, _pp, _up, _vp]),
    ("caf_multi_stream_set_timeout", _int, [_vp, ctypes.c_double]),
    ("caf_multi_stream_surface", _vp, [_vp, _int, _int]),
    ("caf_multi_stream_locate", _int, [_vp, _sz, _sz, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_sz),
                                       ctypes.POINTER(_int)]),
    ("caf_multi_stream_destroy", _int, [_vp]),
    ("caf_multi_surface_shard", _int, [_sz, _int, _int, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    ("caf_multi_surface_reduce", _int, [_pp, _int, _pp]),
    ("caf_rccl_library", _int, [ctypes.c_char_p]),
    ("caf_multi_surface_create", _int, [ctypes.POINTER(_int), _int, _sz, _dp, _sz, _u32, _int, ctypes.c_uint, ctypes.POINTER(_vp)]),
    ("caf_multi_surface_devices", _int, [_vp]),
    ("caf_multi_surface_slab", _vp, [_vp, _int]),
    ("caf_multi_surface_worker_info", _int, [_vp, _int, ctypes.POINTER(_int), ctypes.POINTER(_sz), ctypes.POINTER(_sz),
                                             ctypes.POINTER(ctypes.c_char_p)]),
    ("caf_multi_surface_run", _int, [_vp, _vp, _vp, _vp, _up, _vp, _pp]),
    ("caf_multi_surface_set_timeout", _int, [_vp, ctypes.c_double]),
    ("caf_multi_surface_run_batch", _int, [_vp, _vp, _vp, _sz, _up, _vp, _pp]),
    ("caf_multi_surface_batch_results", _int, [_vp, _int, ctypes.POINTER(_sz), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                               ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    ("caf_multi_surface_run_stats", _int, [_vp, _dp, _pp]),
    ("caf_multi_surface_timing_begin", _int, [_vp]),
    ("caf_multi_surface_timing_end", _int, [_vp, _dp, _up]),
    ("caf_multi_surface_host_alloc", _int, [_vp, _sz, ctypes.POINTER(_vp)]),
    ("caf_multi_surface_host_free", _int, [_vp, _vp]),
    ("caf_multi_surface_host_register", _int, [_vp, _vp, _sz]),
    ("caf_multi_surface_host_unregister", _int, [_vp, _vp]),
    ("caf_multi_surface_destroy", _int, [_vp]),
    ("caf_peak_exchange_stage", _int, [_vp, _int, _vp, _sz, _vp, _vp, _sz, _vp]),
    ("caf_debug_guard_bands", _int, [_sz]),
    ("caf_debug_check_guards", _int, [ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
]

_libs: dict = {}


_torch_rccl = None  # torch's own librccl.so, if its HIP runtime was bound first (see _prefer_torch_hip_runtime)


def _prefer_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1
    (same SONAMEs as /opt/rocm).  Two HSA runtimes in one process cannot both own the
    GPU, so when torch is installed bind to ITS runtime before libcaf_hip.so pulls in
    the system one; torch imported later then finds its own libraries already loaded.
    The library needs only hip_4.2/6.0-versioned symbols, present in both."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("CAF_HIP_SYSTEM_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    global _torch_rccl
    libdir = Path(spec.origin).parent / "lib"
    if (libdir / "librccl.so").exists():
        _torch_rccl = libdir / "librccl.so"  # built against the runtime bound below: the one RCCL to dlopen in this process
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = libdir / name
        if p.exists():
            try:
                ctypes.CDLL(str(p), mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load(path=None) -> ctypes.CDLL:
    """Load libcaf_hip.so (or the library at `path`, e.g. MEASURE_LIB_PATH) and bind every
    declared symbol; raise if absent."""
    path = Path(path if path is not None else os.environ.get("CAF_HIP_LIB", LIB_PATH)).resolve()
    if str(path) in _libs:
        return _libs[str(path)]
    if not path.exists():
        raise ImportError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C caf_cookoff_amd/csrc). "
            "caf_cookoff_amd has no CPU fallback.")
    _prefer_torch_hip_runtime()
    lib = ctypes.CDLL(str(path))
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if _torch_rccl is not None:  # CAF_MULTI_REDUCE_RCCL: load the RCCL that matches the HIP runtime of this process
        lib.caf_rccl_library(str(_torch_rccl).encode())
    _libs[str(path)] = lib
    return lib


def check(rc: int, lib=None) -> None:
    if rc != CAF_OK:
        msg = (lib or load()).caf_last_error_string()
        raise CafError(rc, msg.decode() if msg else "")
