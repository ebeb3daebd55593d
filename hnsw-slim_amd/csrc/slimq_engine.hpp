// slimq_engine.hpp -- device view + launch interface of the HNSW-SlimQ (RaBitQ) search path.
#pragma once
#include "engine.hpp"

namespace hs {

// RaBitQ side of a SlimQ index resident in HBM; the graph itself (tile0 / CSR / labels) lives in DevIndex.
struct DevSlimQ {
  const uint32_t *rec;    // n x rec_words: {f_add, f_rescale, cluster id, f_error}, then padded/64 u64 sign-code words
  const uint32_t *ftile;  // n x tile_stride x rec_words: level-0 adjacency with the neighbours' records inline (4th header
                          // word = neighbour id, 0xFFFFFFFF = empty slot); null when the max degree exceeds 64
  const uint32_t *uptile; // (#upper slots) x up_stride x (rec_words + 4): tile of (node, level >= 1) at slot
                          // up_base[node] + level - 1, records as in ftile followed by {up_base[neighbour], 0, 0, 0}; nullable
  uint32_t up_stride, ep_base;   // ep_base = up_base[enterpoint]
  const float *raw;       // n x dim: the dataset rows of setDataset(), by INTERNAL id (hnswalg_slimq.h:748)
  const float *cent;      // ncl x padded: rotated centroids
  const uint8_t *flips;   // 4 x padded/8: FhtKacRotator sign flips
  uint32_t rec_words, padded, trunc, ncl;
  float fht_scale;        // 1 / sqrt(trunc)
  double t_const;         // quant::faster_config(padded, 4).t_const
};

struct SlimQArgs {
  const float *queries;   // nq x dim (device)
  uint32_t nq, k;
  uint32_t pool_cap;      // SearchBuffer capacity = ef_ (setEf, hnswalg_slimq.h:346-349)
  uint32_t hash_slots;    // expanded-node set, power of two: in LDS, or -- second pass -- in global memory:
  uint32_t *fb_tab;       // nullable; hash_slots words per WORKGROUP (indexed by blockIdx.x): the always-issued, normally empty
                          // second pass then asks for no more LDS than the first one (it waited for 64 KiB to drain)
  uint32_t select_mask, grid;
  uint64_t *out_labels;   // nq x k, the reference's heap-array order (hnswalg_slimq.h:1921-1923); ~0 beyond count
  float *out_dists;       // nq x k exact distances of those entries
  uint32_t *out_counts;   // nq: entries found (<= k)
  uint32_t *stats;        // nq x 4 {expansions, estimates, pool inserts, revisits}   (nullable)
  uint32_t *status;       // nq
  uint32_t *counters;     // [0] expanded-set overflows
  uint32_t *trace;        // debug (nullable): nq x trace_cap, the pops in order: node id, bit 31 set = revisit
  uint32_t trace_cap;
  const uint32_t *prep;   // nq x slimq_prep_words(): per-query records written by launch_slimq_prep
  // first pass in two launches (as SearchArgs::phase, engine.hpp): 1 = descent only -> entry[qi] = {node, estimate,
  // estimates so far, -}; 2 = level-0 search from entry[], queries taken in the order order[] gives; 0 = one launch
  uint32_t phase;
  uint4 *entry;
  const uint32_t *order;
};

// Query preparation (rotation, split query, centroid table) -> prep[nq x slimq_prep_words(ncl, padded)]:
//   [0] delta [1] vl [2] k1xsumq [3] -  [4 .. 4+ncl) g_add per cluster  [8-byte aligned] 4 bit planes (u64) per 64-dim block.
// dbg_y (nullable): nq x padded, the rotated queries.
uint32_t slimq_prep_words(uint32_t ncl, uint32_t padded);
hipError_t launch_slimq_prep(const DevSlimQ &sq, uint32_t dim, int metric, const float *queries, uint32_t nq, uint32_t *prep,
                             float *dbg_y, hipStream_t stream);
size_t slimq_lds_bytes(uint32_t dim, uint32_t padded, uint32_t ncl, uint32_t k, uint32_t hash_slots);
bool slimq_supported(uint32_t pool_cap);
hipError_t launch_slimq(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, hipStream_t stream);

}  // namespace hs
