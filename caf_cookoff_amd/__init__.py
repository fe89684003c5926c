"""MI355X-native Cross Ambiguity Function engine (drop-in for the caf_rust hot path).

Importing the package does not touch the GPU; the first Engine does.  The HIP
library is loaded lazily by :mod:`caf_cookoff_amd._lib` and its absence is an
error, never a fallback.
"""
from ._lib import CAF_C64, CAF_C128, CafError, CafPeak, LIB_PATH, MEASURE_LIB_PATH, load  # noqa: F401
from .caf import (CafHip, CafSurfaceRow, Engine, MultiStream, MultiSurface, Plan, Stream, Xcor, debug_check_guards,  # noqa: F401
                  debug_guard_bands, default_engine, multi_stream_share, multi_surface_reduce, multi_surface_shard)
from .io import load_files, read_file_c64, read_file_c64_f32, write_file_binary  # noqa: F401
from .shifts import bench_shifts, gen_float_shifts, shard_range  # noqa: F401

__all__ = ["CafHip", "CafSurfaceRow", "Engine", "Plan", "Stream", "MultiStream", "multi_stream_share", "MultiSurface", "multi_surface_shard",
           "multi_surface_reduce", "debug_guard_bands", "debug_check_guards", "Xcor", "CafError", "CafPeak", "default_engine",
           "read_file_c64", "read_file_c64_f32", "write_file_binary", "load_files", "gen_float_shifts",
           "bench_shifts", "shard_range", "load", "LIB_PATH", "MEASURE_LIB_PATH", "CAF_C128", "CAF_C64"]
