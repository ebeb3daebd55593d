// convert_engine.hpp -- interface between host_graph.hpp (SlimGraph::convert_gpu) and convert_gpu.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace hs {

// One task per (node, level) list of the vanilla graph: level-0 lists first (task v = node v), then the upper-level lists in
// node order (task n + upb[v] + l - 1 = level l of node v).
struct ConvertInput {
  const float *vec = nullptr;   // n x dim, row-major (host)
  uint32_t n = 0, dim = 0;
  int metric = 0;
  std::vector<uint32_t> t_node, t_level, t_off, t_size;   // per task: node, level, slice of `lists`
  std::vector<uint32_t> t_mlim;                           // degree budget of the first pruning (hnswalg_slim.h:957-961, 969-973)
  std::vector<uint32_t> t_limit;                          // capacity of the level (maxM0 / maxM, :1037)
  std::vector<uint32_t> lists;                            // concatenated source lists
  std::vector<uint32_t> upb;                              // n: index of the node's first upper-level task
};

// Phases 1-3 of convertFromHNSW on device `device`: fin[t * 32 .. + fin_cnt[t]) = the final list of task t (after the reverse-edge
// union and the re-prune, before the hierarchical filter).  needs_host: some list exceeded the on-chip buffers -- the caller
// falls back to the CPU conversion.  kernel_ms (nullable): device time of the kernels.
hipError_t gpu_convert_lists(const ConvertInput &in, int device, std::vector<uint32_t> &fin, std::vector<uint32_t> &fin_cnt, bool &needs_host,
                             double *kernel_ms);

}  // namespace hs
