// engine.hpp -- device-side index view + launch interface shared by beam_search.hip and capi.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "heap_emul.hpp"

namespace hs {

// Read-only index resident in HBM (one copy per GPU).
struct DevIndex {
  const float *vec;         // n x dim fp32, row-major, rows 64-byte aligned when dim % 16 == 0
  const uint32_t *row_ptr0; // n+1     : CSR row pointers of the level-0 adjacency
  const uint32_t *cols;     // n_edges : neighbour ids (level 0 first, then upper-level slices)
  const uint32_t *up_base;  // n       : first up_ptr entry of node i, 0xFFFFFFFF when level(i)==0
  const uint32_t *up_ptr;   //         : level-l slice of i = cols[up_ptr[b+l-1] .. up_ptr[b+l])
  const uint32_t *tile0;    // n x tile_stride: level-0 ids of node i padded with 0xFFFFFFFF to one
                            //         aligned tile (stride 16/32/48/64 ids), or null when max degree > 64
  const uint2 *uptile;      // (#upper slots) x up_stride {neighbour id, up_base[neighbour]}: level l of node i sits at slot
                            //         up_base[i] + l - 1, padded with 0xFFFFFFFF; null when an upper list exceeds 64 ids
  const uint64_t *labels;   // n
  const uint8_t *deleted;   // n       : delete mark as the reference reads it
  uint32_t n, dim, tile_stride, up_stride, ep_base;   // ep_base = up_base[enterpoint]
  int32_t maxlevel, threshold_level;
  uint32_t enterpoint;
  int32_t has_deleted, kind, metric;
};

enum : uint32_t { ST_TODO = 0, ST_DONE = 1, ST_OVERFLOW = 2, ST_HAZARD = 3 };

struct SearchArgs {
  const float *queries;  // nq x dim (device)
  uint32_t nq, k, ef;    // ef = max(ef_, k)   (hnswalg_slim.h:2080)
  uint32_t cand_cap;     // candidate-heap capacity (entries)
  uint32_t hash_slots;   // visited-set tier-1 (LDS) slots
  uint32_t *spill;       // per-query tier-2 region in global memory (nullable): [spill_slots u32 visited-set
                         // slots][cand2_cap x 8 B candidate-heap slots], spill_stride words apart
  uint32_t spill_slots, spill_stride, cand2_cap, log_cap;  // then log_cap x 8 B: result-set insertion log, then hop_cap bytes
  int32_t mode;          // hs_mode
  int32_t mark_ep;       // tag the enter point visited before the descent (slim (q,k) overloads)
  uint32_t select_mask;  // process query qi iff (1 << status[qi]) & select_mask
  uint32_t grid;         // workgroups to launch (grid-stride over queries)
  uint32_t *out_labels32;
  uint64_t *out_labels64;
  float *out_dists;
  uint32_t *out_counts;
  uint32_t *stats;       // nq x 4 {n_dist, n_hops, n_nbr, pass that answered}   (nullable)
  Pair *raw_top;         // nq x raw_stride (nullable; strict kernel only)
  uint32_t *raw_size;    // nq
  uint32_t raw_stride;
  uint32_t *status;      // nq
  uint32_t *counters;    // [0] visited-set overflows, [1] candidate-heap overflows, [2] tie hazards, [3] tier-2 spills (this pass)
  uint32_t pass_id;
  uint32_t *queue;       // device-wide work counter (unused by the shipped kernels; zeroed per launch group)
  uint32_t hash_fill_shift;   // visited-set tier 1 is frozen at 1 - 2^-shift of its slots (0 = the default 2: 75 %)
  uint32_t flat;         // fast kernel: start the level-0 search without the candidate heap (beam_search.hip, FLAT)
  uint32_t hop_cap;      // bytes of the per-expansion accept counts behind the insertion log (one per expansion of the flat start)
  uint32_t vis_bits;     // fast kernel: 0 = visited-set tier 1 in 32-bit slots; else the width of the id space for the
                         // 16-bit form (search_common.hpp; hash_slots / 4 buckets, a power of two, vis_bits - log2(buckets) <= 16)
  // Fast kernel in two launches (see launch_order): phase 1 stops after the upper-level descent and leaves
  // {level-0 entry node, its distance, n_dist, n_hops} in entry[qi]; phase 2 takes its queries in the order order[] gives
  // and starts each from its entry.  phase 0: one launch does both, in index order.
  uint32_t phase;
  uint4 *entry;
  const uint32_t *order;
  // Last-resort pass (strict kernel, pass_id 2): candidate heap and tier-2 visited set in per-WORKGROUP regions of global memory
  // (cand_cap entries / spill_slots words each, indexed by blockIdx.x), so that the launch -- which goes out with every batch and is
  // normally empty -- asks for a dozen KiB of LDS instead of a drained CU.  Null in every other pass.
  void *fb_cand;
  uint32_t *fb_spill;
  // Flat kernel (flat_search.hip): visited set of fl_nb 16-byte buckets; bucket = h mod fl_nb, remainder = h div fl_nb =
  // umulhi(h, fl_mul) >> fl_sh (exact for h < 2^vis_bits; remainders fit 15 bits)
  uint32_t fl_nb, fl_mul, fl_sh;
};

// Bytes of dynamic LDS one query (one wavefront) needs.
size_t strict_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots);
size_t fast_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots);
// Fast path availability for this shape (level-0 tile present, threshold_level == 0, k < ef <= 512).
bool fast_supported(const DevIndex &ix, uint32_t ef, uint32_t k);

// Strict kernel: the reference's result/candidate arrays with libstdc++ heap mechanics, reference
// output order.  Fast kernel: same traversal and candidate mechanics, result set kept as a sorted
// register array; queries whose answer could depend on the result heap's layout are flagged ST_HAZARD.
hipError_t launch_strict(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
// order[] = the queries sorted by decreasing entry[].y (distance of the level-0 entry), bucket-exact: the queries likely
// to take the most expansions start first, so that a launch does not end on a few late-started long ones.
hipError_t launch_order(const uint4 *entry, uint32_t *order, uint32_t nq, hipStream_t stream);
// Lean kernel (lean_search.hip): the fast kernel's algorithm with a keys-only result set (no ids, no LDS staging of the merge).
bool lean_supported(const DevIndex &ix, uint32_t ef, uint32_t k);
size_t lean_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots);
hipError_t launch_lean(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
// Flat kernel (flat_search.hip): lazy candidate heap -- replayed from the insertion log only when its layout decides a pop.
bool flatk_supported(const DevIndex &ix, uint32_t ef, uint32_t k);
size_t flatk_lds_bytes(uint32_t dim, uint32_t ef, uint32_t nb);
uint32_t flatk_waves_per_cu(uint32_t dim, uint32_t ef);
hipError_t launch_flatk(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
hipError_t flat_heap_ops(const uint32_t *d_ops, uint32_t n_ops, uint2 *d_spill, uint2 *d_heap, uint2 *d_pops, uint32_t *d_n, int wave_pop,
                         uint32_t lds_slots, hipStream_t stream);
hipError_t launch_fast(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);

}  // namespace hs
