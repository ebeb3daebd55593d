// bf_engine.hpp -- launch interface of the exhaustive k-NN scan (brute_force.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hs {

// workspace size for the per-(chunk, wave) sorted runs; also fixes the launch geometry
size_t bf_partial_bytes(uint32_t n, uint32_t nq, uint32_t k, uint32_t *grid_x, uint32_t *rows_per_block);
// base n x dim (every dim <= 4096; dim % 16 == 0 is the tuned kernel), labels nullable (row index), queries nq x dim, all device pointers; out_* nq x k sorted by
// (dist, label) ascending, ~0 / +inf beyond out_counts[q] (= min(k, n)).  k <= 64.
hipError_t launch_brute_force(const float *base, const uint64_t *labels, uint32_t n, uint32_t dim, int metric, const float *queries,
                              uint32_t nq, uint32_t k, void *partial, uint32_t grid_x, uint32_t rows_per_block, uint64_t *out_labels,
                              float *out_dists, uint32_t *out_counts, hipStream_t stream);

}  // namespace hs
