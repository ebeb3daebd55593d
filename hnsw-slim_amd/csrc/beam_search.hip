// beam_search.hip -- the hot path on gfx950: one wavefront (64 lanes) per query runs the reference's
// whole searchKnn (upper-layer greedy descent -> level-0 best-first beam -> k-selection) against an
// index resident in HBM.
//
// Replaces (paths relative to /root/reference/third_party/hnswlib/):
//   HierarchicalNSWSlim::searchKnn            hnswalg_slim.h:1907-2028, 2030-2131
//   HierarchicalNSWSlim::searchBaseLayerST    hnswalg_slim.h:321-457   (bare_bone and !bare_bone)
//   HierarchicalNSWSlim::searchBaseLayer      hnswalg_slim.h:222-316   (0 < layer <= threshold_level)
//   HierarchicalNSW::searchKnn / searchBaseLayerST  hnswalg.h:1378-1440 / 326-479
//   L2SqrSIMD16ExtAVX512 / InnerProductSIMD16ExtAVX512   space_l2.h:25-54 / space_ip.h:146-199
//   VisitedList                                visited_list_pool.h:10-31 (exact, as an LDS hash set)
//
// Design (MI355X-first, gather-bound -> no MFMA):
//   * adjacency: CSR in HBM, one coalesced read of the popped node's id slice (<=64 ids per pass);
//   * visited set: open-addressing hash of 32-bit ids in LDS (exact: a false positive would change
//     results); lanes insert their neighbour id with ds_cmpst, the ballot of "newly inserted" gives the
//     unvisited list in adjacency order;
//   * distances: 4 lanes per neighbour row, 16 rows per pass; lane `sub` loads the 16-byte chunk
//     16*s+4*sub of every 64-byte step and owns AVX-512 lane accumulators 4*sub..4*sub+3, so the fp32
//     sum is formed in exactly the reference's order (dist_recipe.hpp) -> bit-identical distances;
//   * result / candidate heaps: the reference's raw arrays live in LDS and lane 0 applies the very
//     push_heap/pop_heap sift sequence of libstdc++ (heap_emul.hpp), processing the pass's distances
//     in adjacency order with the evolving lowerBound -> identical tie behaviour, identical ids;
//   * a query that outgrows its LDS scratch (candidate heap or hash set) is flagged and re-run by the
//     same kernel with a whole CU's LDS (launch_beam_search, only_overflow pass).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "dist_recipe.hpp"
#include "engine.hpp"
#include "heap_emul.hpp"

namespace hs {

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

struct LdsLayout {
  uint32_t off_q, off_top, off_cand, off_hash, off_nid, off_nd, total;
};
__host__ __device__ inline LdsLayout lds_layout(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  LdsLayout l;
  l.off_q = 0;
  l.off_top = align_up(dim * 4, 16);
  l.off_cand = l.off_top + align_up((ef + 1) * 8, 16);
  l.off_hash = l.off_cand + align_up(cand_cap * 8, 16);
  l.off_nid = l.off_hash + hash_slots * 4;
  l.off_nd = l.off_nid + 64 * 4;
  l.total = l.off_nd + 64 * 4;
  return l;
}
size_t beam_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  return lds_layout(dim, ef, cand_cap, hash_slots).total;
}

__device__ __forceinline__ void wave_sync() { __syncthreads(); }  // one wavefront per workgroup
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// Visited-set insert: true when `id` was not present (visited_list_pool.h semantics: test-and-mark).
__device__ __forceinline__ bool hash_insert(uint32_t *tab, uint32_t mask, uint32_t id) {
  uint32_t h = (id * 2654435761u) >> 7;
  while (true) {
    h &= mask;
    uint32_t old = atomicCAS(&tab[h], kEmpty, id);
    if (old == kEmpty) return true;
    if (old == id) return false;
    h++;
  }
}

// Distances query -> rows nid[0..cnt), 16 rows per pass, 4 lanes per row; nd[j] receives the value.
template <int METRIC>
__device__ __forceinline__ void wave_dists(const DevIndex &ix, const float *qv, const uint32_t *nid, float *nd,
                                           uint32_t cnt, int lane) {
  const int sub = lane & 3, grp = lane >> 2;
  const uint32_t steps = ix.dim >> 4;
  for (uint32_t base = 0; base < cnt; base += 16) {
    const uint32_t j = base + grp;
    if (j < cnt) {
      const uint32_t id = nid[j];
      const float4 *row = reinterpret_cast<const float4 *>(ix.vec + (size_t)id * ix.dim) + sub;
      const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (uint32_t s = 0; s < steps; s++) {
        const float4 x4 = row[s * 4];
        const float4 q4 = qq[s * 4];
        const float x[4] = {x4.x, x4.y, x4.z, x4.w};
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        if (METRIC == METRIC_L2) l2_step4(acc, q, x);
        else ip_step4(acc, q, x);
      }
      if (METRIC == METRIC_L2) {
        // TmpRes[0] + TmpRes[1] + ... + TmpRes[15], left to right (space_l2.h:49-51)
        float r = ((acc[0] + acc[1]) + acc[2]) + acc[3];
#pragma unroll
        for (int k = 1; k < 4; k++) {
          const float p = __shfl_up(r, 1, 64);
          if (sub == k) r = (((p + acc[0]) + acc[1]) + acc[2]) + acc[3];
        }
        if (sub == 3) nd[j] = r;
      } else {
        // _mm512_reduce_add_ps: halves 16 -> 8 -> 4 -> 2 -> 1 (space_ip.h:197), then 1 - ip (:201-204)
        float h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) h[i] = acc[i] + __shfl_down(acc[i], 2, 64);
#pragma unroll
        for (int i = 0; i < 4; i++) h[i] = h[i] + __shfl_down(h[i], 1, 64);
        const float a0 = h[0] + h[2], a1 = h[1] + h[3];
        const float ip = a0 + a1;
        if (sub == 0) nd[j] = 1.0f - ip;
      }
    }
  }
}

struct QState {
  uint32_t top_size, cand_size, n_ins;
  float lb;
  uint32_t n_dist, n_hops, n_nbr;
#ifdef HS_PROFILE
  unsigned long long t[8];  // cycles: 0 pop, 1 adjacency, 2 visited, 3 distances, 4 accept, 5 upper, 6 init, 7 final
#endif
};
// Diagnostic build only (make prof): per-phase shader-clock stamps, written to the stats buffer's
// tail; the product build compiles these to nothing.
#ifdef HS_PROFILE
#define HS_T0() unsigned long long _t0 = clock64()
#define HS_LAP(st, i) do { unsigned long long _t1 = clock64(); (st).t[i] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define HS_T0() do {} while (0)
#define HS_LAP(st, i) do {} while (0)
#endif

// One best-first beam over the `level` slices (level 0: searchBaseLayerST; >0: searchBaseLayer).
// Returns false when the on-chip scratch overflowed.
template <int METRIC>
__device__ bool beam(const DevIndex &ix, const SearchArgs &a, int level, bool bare, const float *qv, Pair *top,
                     Pair *cand, uint32_t *hash, uint32_t *nid, float *nd, QState &st, int lane) {
  const uint32_t ef = a.ef;
  const uint32_t hmask = a.hash_slots - 1;
  const uint32_t hash_limit = a.hash_slots - (a.hash_slots >> 2);
  HS_T0();
  while (true) {
    wave_sync();
    if (st.cand_size == 0) break;
    const Pair c = cand[0];
    const float cd = unif(c.d);
    const uint32_t cid = uni(c.id);
    const bool stop = bare ? (cd > st.lb) : (cd > st.lb && st.top_size == ef);  // hnswalg_slim.h:340 / :346-347, :237
    if (stop) break;
    wave_sync();
    if (lane == 0) pop_heap(cand, (long)st.cand_size, GreaterD());  // :353-354
    st.cand_size--;
    st.n_hops++;
    wave_sync();
    HS_LAP(st, 0);
    uint32_t s, e;
    if (level == 0) {
      s = ix.row_ptr0[cid];
      e = ix.row_ptr0[cid + 1];
    } else {
      const uint32_t b = ix.up_base[cid];
      if (b == kNone) continue;  // neighbors == nullptr (:247-249)
      s = ix.up_ptr[b + level - 1];
      e = ix.up_ptr[b + level];
    }
    s = uni(s);
    e = uni(e);
    for (uint32_t base = s; base < e; base += 64) {
      const uint32_t m = min(64u, e - base);
      if (st.n_ins + m > hash_limit || st.cand_size + m > a.cand_cap) return false;
      uint32_t id = 0;
      bool isnew = false;
      if ((uint32_t)lane < m) id = ix.cols[base + lane];
#ifdef HS_PROFILE
      id = __shfl(id, lane, 64);  // force the load to land before the stamp
#endif
      HS_LAP(st, 1);
      if ((uint32_t)lane < m) isnew = hash_insert(hash, hmask, id);  // :392-393
      const unsigned long long nm = __ballot(isnew);
      const uint32_t cnt = __popcll(nm);
      st.n_nbr += m;
      if (cnt == 0) continue;
      wave_sync();
      if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
      wave_sync();
      st.n_ins += cnt;
      st.n_dist += cnt;
      HS_LAP(st, 2);
      wave_dists<METRIC>(ix, qv, nid, nd, cnt, lane);  // :395-396
      wave_sync();
      HS_LAP(st, 3);
      uint32_t ts = st.top_size, cs = st.cand_size;
      float lb = st.lb;
      if (lane == 0) {
        for (uint32_t j = 0; j < cnt; j++) {
          const float d = nd[j];
          if (ts < ef || lb > d) {  // :403-404
            const uint32_t nb = nid[j];
            cand[cs].d = d;  // :408-411
            cand[cs].id = nb;
            cs++;
            push_heap(cand, (long)cs, GreaterD());
            if (bare || !ix.deleted[nb]) {  // :418
              top[ts].d = d;
              top[ts].id = nb;
              ts++;
              push_heap(top, (long)ts, LessD());
            }
            while (ts > ef) {  // :434-448
              pop_heap(top, (long)ts, LessD());
              ts--;
            }
            if (ts > 0) lb = top[0].d;  // :450-452
          }
        }
      }
      st.top_size = uni(ts);
      st.cand_size = uni(cs);
      st.lb = unif(lb);
      HS_LAP(st, 4);
    }
  }
  return true;
}

// The whole searchKnn of one query, executed by one wavefront.
template <int METRIC>
__device__ void search_one(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;

  const LdsLayout L = lds_layout(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  Pair *top = reinterpret_cast<Pair *>(smem + L.off_top);
  Pair *cand = reinterpret_cast<Pair *>(smem + L.off_cand);
  uint32_t *hash = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  uint32_t *nid = reinterpret_cast<uint32_t *>(smem + L.off_nid);
  float *nd = reinterpret_cast<float *>(smem + L.off_nd);

  const uint32_t k = a.k;
  if (ix.n == 0) {  // cur_element_count == 0 (hnswalg_slim.h:2031-2032)
    if (lane == 0) {
      if (a.out_counts) a.out_counts[qi] = 0;
      a.status[qi] = ST_DONE;
    }
    return;
  }

  for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
  for (uint32_t i = lane; i < a.hash_slots; i += 64) hash[i] = kEmpty;
  QState st;
#ifdef HS_PROFILE
  for (int i = 0; i < 8; i++) st.t[i] = 0;
#endif
  HS_T0();
  st.n_dist = st.n_hops = st.n_nbr = 0;
  st.n_ins = 0;
  st.top_size = st.cand_size = 0;

  // ---- enter point (hnswalg_slim.h:2033-2035) -------------------------------------------------
  uint32_t cur = ix.enterpoint;
  if (lane == 0) nid[0] = cur;
  wave_sync();
  wave_dists<METRIC>(ix, qv, nid, nd, 1, lane);
  wave_sync();
  float curdist = unif(nd[0]);
  st.n_dist = 1;
  if (a.mark_ep) {  // visited_array[enterpoint] = tag (hnswalg_slim.h:1919)
    if (lane == 0) hash_insert(hash, a.hash_slots - 1, cur);
    st.n_ins++;
  }

  HS_LAP(st, 6);
  // ---- upper layers: greedy descent (hnswalg_slim.h:2040-2078, hnswalg.h:1389-1415) ------------
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      st.n_hops++;
      const uint32_t b = ix.up_base[cur];
      if (b == kNone) continue;
      const uint32_t s = uni(ix.up_ptr[b + lvl - 1]), e = uni(ix.up_ptr[b + lvl]);
      for (uint32_t base = s; base < e; base += 64) {
        const uint32_t m = min(64u, e - base);
        wave_sync();
        if ((uint32_t)lane < m) nid[lane] = ix.cols[base + lane];
        wave_sync();
        wave_dists<METRIC>(ix, qv, nid, nd, m, lane);
        wave_sync();
        st.n_nbr += m;
        st.n_dist += m;
        // first index attaining the minimum == what the sequential `if (d < curdist)` scan ends on
        float d = (uint32_t)lane < m ? nd[lane] : FLT_MAX;
        uint32_t l = lane;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const float od = __shfl_xor(d, off, 64);
          const uint32_t ol = __shfl_xor(l, off, 64);
          if (od < d || (od == d && ol < l)) { d = od; l = ol; }
        }
        if (l < m && d < curdist) {  // hnswalg_slim.h:2071-2075
          curdist = d;
          cur = uni(nid[l]);
          changed = true;
        }
      }
    }
  }

  HS_LAP(st, 5);
  // ---- level-0 (and threshold-level) beams ----------------------------------------------------
  bool bare = !ix.has_deleted;  // hnswalg_slim.h:2114, hnswalg.h:1421 (no filter on this path)
  const bool ep_deleted = ix.deleted[cur] != 0;
  if (ix.kind == 0) st.n_dist++;  // searchBaseLayerST recomputes the entry distance (hnswalg.h:351)
  wave_sync();
  if (lane == 0) {
    hash_insert(hash, a.hash_slots - 1, cur);  // visited_array[currObj] = tag (hnswalg_slim.h:2102)
    if (ix.kind == 0 && !bare && ep_deleted) {  // hnswalg.h:359-362
      cand[0].d = FLT_MAX;
      cand[0].id = cur;
    } else {
      top[0].d = curdist;  // hnswalg_slim.h:2100-2101
      top[0].id = cur;
      cand[0] = top[0];
    }
  }
  st.n_ins++;
  st.cand_size = 1;
  if (ix.kind == 0 && !bare && ep_deleted) {
    st.top_size = 0;
    st.lb = FLT_MAX;
  } else {
    st.top_size = 1;
    st.lb = ep_deleted ? FLT_MAX : curdist;  // hnswalg_slim.h:2104-2106
  }
  bool ok = true;
  for (int lvl = min(ix.threshold_level, ix.maxlevel); lvl > 0 && ok; lvl--) {  // hnswalg_slim.h:2108-2113
    ok = beam<METRIC>(ix, a, lvl, /*bare=*/false, qv, top, cand, hash, nid, nd, st, lane);
    // next beam starts from candidate_set <- copy of top_candidates (+ make_heap) (:228-233, :327-332)
    wave_sync();
    if (lane == 0) {
      for (uint32_t i = 0; i < st.top_size; i++) cand[i] = top[i];
      make_heap(cand, (long)st.top_size, GreaterD());
    }
    st.cand_size = st.top_size;
  }
  if (ok) ok = beam<METRIC>(ix, a, 0, bare, qv, top, cand, hash, nid, nd, st, lane);
  wave_sync();
  if (!ok) {
    if (lane == 0) {
      a.status[qi] = ST_OVERFLOW;
      atomicAdd(a.overflow_count, 1u);
    }
    return;
  }

  // ---- raw result heap (parity/debug) ---------------------------------------------------------
#ifdef HS_PROFILE
  _t0 = clock64();
#endif
  if (a.raw_top) {
    for (uint32_t i = lane; i < st.top_size; i += 64) a.raw_top[(size_t)qi * a.raw_stride + i] = top[i];
    if (lane == 0) a.raw_size[qi] = st.top_size;
  }
  // ---- k-selection ----------------------------------------------------------------------------
  uint32_t ts = st.top_size;
  if (a.mode == 0) {
    // std::nth_element(top, top+k, top+size) then result[i] = label(top[i]) (hnswalg_slim.h:2126-2130)
    if (lane == 0 && ts >= k) nth_element(top, (long)k, (long)ts, LessD());
  } else {
    // while (size > k) pop_heap (hnswalg_slim.h:2019-2022, hnswalg.h:1430-1432)
    if (lane == 0)
      while (ts > k) {
        pop_heap(top, (long)ts, LessD());
        ts--;
      }
    ts = min(ts, k);
  }
  wave_sync();
  const uint32_t valid = min(ts, k);
  for (uint32_t i = lane; i < k; i += 64) {
    const bool v = i < valid;
    const Pair p = v ? top[i] : Pair{__builtin_inff(), 0};
    const uint64_t label = v ? ix.labels[p.id] : ~0ull;
    if (a.out_labels32) a.out_labels32[(size_t)qi * k + i] = v ? (uint32_t)label : 0xFFFFFFFFu;  // size_t -> tableint truncation (:2129)
    if (a.out_labels64) a.out_labels64[(size_t)qi * k + i] = label;
    if (a.out_dists) a.out_dists[(size_t)qi * k + i] = p.d;
  }
  if (lane == 0) {
    if (a.out_counts) a.out_counts[qi] = valid;
    if (a.stats) {
      a.stats[qi * 4 + 0] = st.n_dist;
      a.stats[qi * 4 + 1] = st.n_hops;
      a.stats[qi * 4 + 2] = st.n_nbr;
      a.stats[qi * 4 + 3] = a.only_overflow ? 1u : 0u;
#ifdef HS_PROFILE
      HS_LAP(st, 7);
      // diagnostic build: phase cycles go to a buffer of their own behind the nq x 4 stats block
      unsigned long long *pt = reinterpret_cast<unsigned long long *>(a.stats + (size_t)a.nq * 4) + (size_t)qi * 8;
      for (int i = 0; i < 8; i++) pt[i] = st.t[i];
#endif
    }
    a.status[qi] = ST_DONE;
  }
}

// grid-stride over queries: the first pass launches one workgroup (= one wavefront) per query, the
// fallback pass a CU-count sized grid that only picks up queries flagged ST_OVERFLOW.
template <int METRIC>
__global__ void __launch_bounds__(64) beam_search_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    if (a.only_overflow && a.status[qi] != ST_OVERFLOW) continue;
    search_one<METRIC>(ix, a, qi, smem);
    wave_sync();
  }
}

hipError_t launch_beam_search(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;
  const size_t lds = beam_lds_bytes(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  auto kern = ix.metric == METRIC_L2 ? beam_search_kernel<METRIC_L2> : beam_search_kernel<METRIC_IP>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const uint32_t grid = a.only_overflow ? min(a.nq, 256u) : a.nq;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, ix, a);
  return hipGetLastError();
}

}  // namespace hs
