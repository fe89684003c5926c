"""Multi-GPU decomposition of the CAF path: contiguous Doppler-row shards + one tiny
reduction for the global peak.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  Rows are independent (mod.rs:135-162 carries no cross-row
state), so there is NO data-path collective: every rank holds the (replicated, 128 KiB)
inputs, computes rows [r*F/G, (r+1)*F/G) of every surface in the batch and keeps its
slab of the surface.  The only exchange is find_peak (mod.rs:31-42):

    every rank contributes (value, key) with key = (global_row << 32 | idx), INT64_MAX if
    it has no peak; ONE all_gather (16 B per surface per rank) makes all pairs visible and
    each rank reduces locally: gmax = max value, then the lowest key among the holders of
    gmax -> lowest global row among equal peaks wins, which is the reference's
    first-strictly-greater scan.

RCCL has no MAXLOC.  The collective is latency-bound (a few us over xGMI for 16 B x batch
per rank) and is issued once per batch, not per surface; `method="allreduce"` keeps the
two-step MAX / MIN-key form (two dependent collectives).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

NO_PEAK_KEY = torch.iinfo(torch.int64).max


def encode_key(row: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(global row, lag index) -> sortable int64; row < 2^31, idx < 2^32."""
    return (row.to(torch.int64) << 32) | idx.to(torch.int64)


def decode_key(key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (row, idx); row = -1 where no rank had a peak > 0 (mod.rs:32-35 initial max)."""
    none = key == NO_PEAK_KEY
    row = torch.where(none, torch.full_like(key, -1), key >> 32)
    idx = torch.where(none, torch.zeros_like(key), key & 0xFFFFFFFF)
    return row, idx


def reduce_global_peak(val: torch.Tensor, row: torch.Tensor, idx: torch.Tensor,
                       group: Optional[dist.ProcessGroup] = None, method: str = "allgather",
                       always_collective: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Combine per-rank shard peaks into the global find_peak result.

    val/row/idx: [batch] local peak value (float64), GLOBAL row position (-1 = no peak)
    and lag index of this rank's shard (the caf_peak records of caf_surface_dev).
    Returns (gmax, grow, gidx), identical on every rank.  Works on CPU (gloo) and GPU
    (nccl/RCCL) tensors; without an initialised process group it is the identity, and so it is in a
    group of one unless `always_collective` (tests: runs the RCCL calls on a single GPU)."""
    val = val.to(torch.float64)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always_collective):
        has = row >= 0
        return (torch.where(has, val, torch.zeros_like(val)), torch.where(has, row, torch.full_like(row, -1)),
                torch.where(has, idx, torch.zeros_like(idx)))
    if method == "allgather":
        has = row >= 0
        mine = torch.stack([torch.where(has, val, torch.zeros_like(val)).view(torch.int64),
                            torch.where(has, encode_key(row, idx), torch.full_like(row, NO_PEAK_KEY, dtype=torch.int64))],
                           dim=1).contiguous()                                   # [batch, 2] int64 bit patterns
        world = dist.get_world_size(group)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)  # (the list form also exists on gloo)
        allp = torch.stack(parts, dim=0)
        vals = allp[:, :, 0].view(torch.float64)                                  # [world, batch]
        keys = allp[:, :, 1]
        gmax = vals.max(dim=0).values
        keys = torch.where((vals == gmax) & (gmax > 0), keys, torch.full_like(keys, NO_PEAK_KEY))
        grow, gidx = decode_key(keys.min(dim=0).values)
        return gmax, grow, gidx
    gmax = torch.where(row >= 0, val, torch.zeros_like(val)).clone()
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    mine = (row >= 0) & (val == gmax) & (gmax > 0)
    key = torch.where(mine, encode_key(row, idx), torch.full_like(row, NO_PEAK_KEY, dtype=torch.int64))
    dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
    grow, gidx = decode_key(key)
    return gmax, grow, gidx


class PeakExchange:
    """The same exchange (``method="allreduce"``) with the library's three element-sized kernels either side of the two
    collectives instead of ~20 tensor operations: ``caf_peak_exchange_stage`` packs the shard records
    (the ``[count, 4]`` float64 ``caf_peak`` tensor :func:`caf_surface_dev` filled, global row positions) into values,
    all-reduce(MAX) in place, packs (row << 32 | idx) keys of the holders of the maximum, all-reduce(MIN) in place, and writes the
    global ``caf_peak`` records ``{val, freq, idx, row}`` (``{0, 0, 0, -1}`` where no shard had a peak) -- ~10 us of device time
    per 256-surface step instead of ~100.  GPU tensors and an engine whose stream is torch's current stream; without a process
    group (or in a group of one, unless ``always_collective``) the collectives are skipped and the result is the shard's own.
    ``peaks_of(out)`` -> (gmax, grow, gidx) in :func:`reduce_global_peak`'s form."""

    def __init__(self, eng, count: int, freqs_all, device):
        self.eng, self.count = eng, int(count)
        self.freqs = torch.as_tensor(freqs_all, dtype=torch.float64).to(device).contiguous()
        self.red = torch.empty(4 * self.count, dtype=torch.float64, device=device)
        self.red_i = self.red.view(torch.int64)
        self.out = torch.empty((self.count, 4), dtype=torch.float64, device=device)

    def __call__(self, peaks: torch.Tensor, group: Optional[dist.ProcessGroup] = None, always_collective: bool = False,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = self.count
        assert peaks.is_cuda and peaks.dtype == torch.float64 and peaks.shape == (n, 4) and peaks.is_contiguous()
        out = self.out if out is None else out
        coll = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or always_collective)
        self.eng.peak_exchange_stage(0, peaks.data_ptr(), n, self.red.data_ptr())
        if coll:
            dist.all_reduce(self.red[n:2 * n], op=dist.ReduceOp.MAX, group=group)
        self.eng.peak_exchange_stage(1, peaks.data_ptr(), n, self.red.data_ptr())
        if coll:
            dist.all_reduce(self.red_i[3 * n:4 * n], op=dist.ReduceOp.MIN, group=group)
        self.eng.peak_exchange_stage(2, 0, n, self.red.data_ptr(), self.freqs.data_ptr(), self.freqs.numel(), out.data_ptr())
        return out

    @staticmethod
    def peaks_of(out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        oi = out.view(torch.int64)
        return out[:, 0], oi[:, 3], oi[:, 2]
