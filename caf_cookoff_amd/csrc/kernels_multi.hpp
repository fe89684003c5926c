// kernels_multi.hpp -- the two element-sized kernels either side of the RCCL peak reduction of a row-sharded
// surface (caf_multi_surface_*, CAF_MULTI_REDUCE_RCCL).  find_peak (caf_rust/src/caf/mod.rs:31-42) keeps the FIRST row
// whose peak is strictly greater than the best so far; over contiguous row shards that is "largest value, then lowest
// global row".  RCCL has no MAXLOC, so the reduction is two dependent all-reduces:
//   1. all-reduce(max) over each shard's peak value           -> gmax on every device
//   2. all-reduce(min) over key = (global_row << 32 | idx) of the shards that hold gmax, UINT64_MAX elsewhere
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/caf_hip.h"

namespace caf {

// One reduction covers `count` surfaces (1 for caf_multi_surface_run, B for caf_multi_surface_run_batch).  The device buffer
// `red` is four arrays of `count` 8-byte words: [val | gmax | key | gkey].
// val[b] = this shard's contribution to the max of surface b (0.0 if the shard has no peak: the reference's initial maximum)
__global__ void k_shard_peak_val(const caf_peak *__restrict__ peak, double *__restrict__ red, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) red[b] = peak[b].row >= 0 ? peak[b].val : 0.0;
}

// gmax[b] = red[count + b] (all-reduced); key[b] = this shard's candidate for the min-key reduction of surface b
__global__ void k_shard_peak_key(const caf_peak *__restrict__ peak, double *__restrict__ red, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) {
        const double gmax = red[count + b];
        const bool mine = peak[b].row >= 0 && gmax > 0.0 && peak[b].val == gmax;
        ((uint64_t *)red)[2 * (size_t)count + b] = mine ? (((uint64_t)peak[b].row << 32) | (peak[b].idx & 0xffffffffull)) : ~0ull;
    }
}

// ---- the same exchange for hosts that run the collectives THEMSELVES (one process per GPU: torch.distributed over RCCL, MPI ...):
// caf_peak_exchange_stage (api/ops.inc).  In-place collectives and SIGNED 64-bit keys (torch and MPI reduce int64, not uint64):
// a row position is < 2^31, so a key is a positive int64 and INT64_MAX means "this shard does not hold the maximum".
//   stage 0   val[b] = gmax[b] = this shard's peak value (0.0 without a peak)      -> the host all-reduces gmax with MAX, in place
//   stage 1   key[b] = gkey[b] = (row << 32 | idx) if the shard holds gmax[b] > 0  -> the host all-reduces gkey with MIN, in place
//   stage 2   out[b] = {gmax, freqs_all[row], idx, row}, or the reference's initial maximum {0, 0, 0, -1} (mod.rs:32-35)
__global__ void k_px_val(const caf_peak *__restrict__ peak, double *__restrict__ red, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) red[b] = red[count + b] = peak[b].row >= 0 ? peak[b].val : 0.0;
}
__global__ void k_px_key(const caf_peak *__restrict__ peak, double *__restrict__ red, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) {
        const double gmax = red[count + b];
        const bool mine = peak[b].row >= 0 && gmax > 0.0 && peak[b].val == gmax;
        const int64_t key = mine ? (int64_t)(((uint64_t)peak[b].row << 32) | (peak[b].idx & 0xffffffffull)) : INT64_MAX;
        ((int64_t *)red)[2 * (size_t)count + b] = ((int64_t *)red)[3 * (size_t)count + b] = key;
    }
}
__global__ void k_px_out(const double *__restrict__ red, const double *__restrict__ freqs_all, long nfreq_all,
                         caf_peak *__restrict__ out, int count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < count) {
        const int64_t key = ((const int64_t *)red)[3 * (size_t)count + b];
        caf_peak pk;
        if (key == INT64_MAX) {
            pk.val = 0.0; pk.freq = 0.0; pk.idx = 0; pk.row = -1;
        } else {
            pk.val = red[count + b];
            pk.row = key >> 32;
            pk.idx = (uint64_t)key & 0xffffffffull;
            pk.freq = pk.row < nfreq_all ? freqs_all[pk.row] : 0.0;
        }
        out[b] = pk;
    }
}

}  // namespace caf
