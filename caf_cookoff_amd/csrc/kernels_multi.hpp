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

// red[0] = this shard's contribution to the max (0.0 if the shard has no peak: the reference's initial maximum)
__global__ void k_shard_peak_val(const caf_peak *__restrict__ peak, double *__restrict__ red)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) red[0] = peak->row >= 0 ? peak->val : 0.0;
}

// red[1] = gmax (all-reduced); key[0] = this shard's candidate for the min-key reduction
__global__ void k_shard_peak_key(const caf_peak *__restrict__ peak, const double *__restrict__ red, uint64_t *__restrict__ key)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double gmax = red[1];
        const bool mine = peak->row >= 0 && gmax > 0.0 && peak->val == gmax;
        key[0] = mine ? (((uint64_t)peak->row << 32) | (peak->idx & 0xffffffffull)) : ~0ull;
    }
}

}  // namespace caf
