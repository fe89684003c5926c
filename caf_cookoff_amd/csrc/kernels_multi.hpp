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

}  // namespace caf
