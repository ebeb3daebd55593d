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
//   * adjacency: CSR in HBM (all levels) plus, for level 0, one aligned tile of ids per node so that an
//     expansion is ONE coalesced 64..256-byte read issued straight from the popped id;
//   * visited set: open-addressing hash of 32-bit ids in LDS (exact: a false positive would change
//     results); lanes insert their neighbour id with ds_cmpst, the ballot of "newly inserted" gives the
//     unvisited list in adjacency order;
//   * distances: 4 lanes per neighbour row, 16 rows per pass; lane `sub` loads the 16-byte chunk
//     16*s+4*sub of every 64-byte step and owns AVX-512 lane accumulators 4*sub..4*sub+3, so the fp32
//     sum is formed in exactly the reference's order (dist_recipe.hpp) -> bit-identical distances;
//   * candidate heap: the reference's raw array in LDS with libstdc++'s push_heap/pop_heap sift sequence
//     applied by lane 0 -> the expansion order among equal-distance candidates is the reference's;
//   * result set, two kernels:
//       strict: the reference's raw top_candidates array + libstdc++ heap/nth_element mechanics
//               (heap_emul.hpp) -> identical array, identical output ORDER, any tie pattern;
//       fast  : a sorted array held in registers (rank r in lane r%64), insertion = ballot + one DPP
//               wave shift; which of several equal-distance entries survives/gets selected is then not
//               defined, so a query whose k-subset could depend on it (k-th and (k+1)-th distance equal)
//               is flagged ST_HAZARD and answered by the strict kernel.  Output sorted by distance.
//   * a query that outgrows its LDS scratch is flagged ST_OVERFLOW and re-run by the strict kernel with
//     a whole CU's LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>

#include "dist_recipe.hpp"
#include "engine.hpp"
#include "heap_emul.hpp"

namespace hs {

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// ---- LDS layouts --------------------------------------------------------------------------------
struct StrictLds { uint32_t off_q, off_top, off_cand, off_hash, off_nid, off_nd, total; };
__host__ __device__ inline StrictLds strict_layout(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  StrictLds l;
  l.off_q = 0;
  l.off_top = align_up(dim * 4, 16);
  l.off_cand = l.off_top + align_up((ef + 1) * 8, 16);
  l.off_hash = l.off_cand + align_up(cand_cap * 8, 16);
  l.off_nid = l.off_hash + hash_slots * 4;
  l.off_nd = l.off_nid + 64 * 4;
  l.total = l.off_nd + 64 * 4;
  return l;
}
struct FastLds { uint32_t off_q, off_cand, off_hash, off_nid, off_nd, total; };
__host__ __device__ inline FastLds fast_layout(uint32_t dim, uint32_t cand_cap, uint32_t hash_slots) {
  FastLds l;
  l.off_q = 0;
  l.off_cand = align_up(dim * 4, 16);
  l.off_hash = l.off_cand + align_up((cand_cap + 2) * 8, 16);  // heap element i lives at slot i+1
  l.off_nid = l.off_hash + hash_slots * 4;
  l.off_nd = l.off_nid + 64 * 4;
  l.total = l.off_nd + 64 * 4;
  return l;
}
size_t strict_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  return strict_layout(dim, ef, cand_cap, hash_slots).total;
}
size_t fast_lds_bytes(uint32_t dim, uint32_t cand_cap, uint32_t hash_slots) {
  return fast_layout(dim, cand_cap, hash_slots).total;
}
bool fast_supported(const DevIndex &ix, uint32_t ef, uint32_t k) {
  return ix.tile0 != nullptr && ix.threshold_level == 0 && ef > k && ef <= 512;
}

// ---- wave helpers -------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() { __syncthreads(); }  // one wavefront per workgroup
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
// whole-wave shift right by one lane (DPP wave_shr:1, a single VALU op on GFX9); lane 0 receives `carry`
__device__ __forceinline__ uint32_t wave_shr1(uint32_t carry, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}

struct Counters {
  uint32_t n_dist, n_hops, n_nbr, n_ins;
#ifdef HS_PROFILE
  unsigned long long t[8];  // cycles: 0 pop, 1 adjacency, 2 visited, 3 distances, 4 accept, 5 upper, 6 init, 7 final
#endif
};
// Diagnostic build only (make prof): per-phase shader-clock stamps, written behind the stats block; the
// product build compiles these to nothing.
#ifdef HS_PROFILE
#define HS_T0() unsigned long long _t0 = clock64()
#define HS_T0_RESET() _t0 = clock64()
#define HS_LAP(st, i) do { unsigned long long _t1 = clock64(); (st).t[i] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define HS_T0() do {} while (0)
#define HS_T0_RESET() do {} while (0)
#define HS_LAP(st, i) do {} while (0)
#endif

// Visited-set insert: true when `id` was not present (visited_list_pool.h semantics: test-and-mark).
__device__ __forceinline__ bool hash_insert(uint32_t *tab, uint32_t mask, uint32_t id) {
  uint32_t h = (id * 2654435761u) >> 7;
  while (true) {
    h &= mask;
    const uint32_t old = atomicCAS(&tab[h], kEmpty, id);
    if (old == kEmpty) return true;
    if (old == id) return false;
    h++;
  }
}

// Distances query -> rows nid[0..cnt), 16 rows per pass, 4 lanes per row; nd[j] receives the value.
// `between()` runs after the first pass's first eight 16-byte loads per lane have been ISSUED and before
// they are consumed: LDS-only work placed there (the candidate heap's pop) hides under the HBM latency.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int METRIC, class Hook = NoHook>
__device__ __forceinline__ void wave_dists(const DevIndex &ix, const float *qv, const uint32_t *nid, float *nd,
                                           uint32_t cnt, int lane, Hook between = Hook()) {
  const int sub = lane & 3, grp = lane >> 2;
  const uint32_t steps = ix.dim >> 4;
  for (uint32_t base = 0; base < cnt; base += 16) {
    const uint32_t j = base + grp;
    const bool act = j < cnt;
    const uint32_t id = act ? nid[j] : 0;
    const float4 *row = reinterpret_cast<const float4 *>(ix.vec + (size_t)id * ix.dim) + sub;
    const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 buf[8];
    if (act) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        if ((uint32_t)i < steps) buf[i] = row[i * 4];
    }
    if (base == 0) between();
    if (act) {
      for (uint32_t s0 = 0; s0 < steps; s0 += 8) {
        float4 nxt[8];
        if (s0 + 8 < steps) {
#pragma unroll
          for (int i = 0; i < 8; i++)
            if (s0 + 8 + i < steps) nxt[i] = row[(s0 + 8 + i) * 4];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (s0 + i < steps) {
            const float4 q4 = qq[(s0 + i) * 4];
            const float x[4] = {buf[i].x, buf[i].y, buf[i].z, buf[i].w};
            const float q[4] = {q4.x, q4.y, q4.z, q4.w};
            if (METRIC == METRIC_L2) l2_step4(acc, q, x);
            else ip_step4(acc, q, x);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) buf[i] = nxt[i];
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

// Shared prologue: stage the query, clear the visited set, entry distance, upper-layer greedy descent
// (hnswalg_slim.h:2033-2078, hnswalg.h:1385-1415).  Leaves (cur, curdist) = level-0 entry.
template <int METRIC>
__device__ __forceinline__ void descend(const DevIndex &ix, const SearchArgs &a, uint32_t qi, float *qv, uint32_t *hash,
                                        uint32_t *nid, float *nd, Counters &c, uint32_t &cur, float &curdist, int lane) {
  for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
  for (uint32_t i = lane; i < a.hash_slots; i += 64) hash[i] = kEmpty;
  cur = ix.enterpoint;
  if (lane == 0) nid[0] = cur;
  wave_sync();
  wave_dists<METRIC>(ix, qv, nid, nd, 1, lane);
  wave_sync();
  curdist = unif(nd[0]);
  c.n_dist = 1;
  c.n_hops = c.n_nbr = c.n_ins = 0;
  if (a.mark_ep) {  // visited_array[enterpoint] = tag (hnswalg_slim.h:1919)
    if (lane == 0) hash_insert(hash, a.hash_slots - 1, cur);
    c.n_ins++;
  }
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      c.n_hops++;
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
        c.n_nbr += m;
        c.n_dist += m;
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
}

__device__ __forceinline__ void flag_query(const SearchArgs &a, uint32_t qi, uint32_t status, uint32_t counter, int lane) {
  if (lane == 0) {
    a.status[qi] = status;
    atomicAdd(a.counters + counter, 1u);
  }
}

__device__ __forceinline__ void write_stats(const SearchArgs &a, uint32_t qi, const Counters &c) {
  if (a.stats) {
    a.stats[qi * 4 + 0] = c.n_dist;
    a.stats[qi * 4 + 1] = c.n_hops;
    a.stats[qi * 4 + 2] = c.n_nbr;
    a.stats[qi * 4 + 3] = a.pass_id;
#ifdef HS_PROFILE
    // diagnostic build: phase cycles go to a buffer of their own behind the nq x 4 stats block
    unsigned long long *pt = reinterpret_cast<unsigned long long *>(a.stats + (size_t)a.nq * 4) + (size_t)qi * 8;
    for (int i = 0; i < 8; i++) pt[i] = c.t[i];
#endif
  }
}

// =================================================================================================
// strict kernel
// =================================================================================================
struct SState { uint32_t top_size, cand_size; float lb; };

// One best-first beam over the `level` slices (level 0: searchBaseLayerST; >0: searchBaseLayer).
// Returns 0 ok, 1 visited-set overflow, 2 candidate-heap overflow.
template <int METRIC>
__device__ int strict_beam(const DevIndex &ix, const SearchArgs &a, int level, bool bare, const float *qv, Pair *top,
                           Pair *cand, uint32_t *hash, uint32_t *nid, float *nd, SState &st, Counters &c, int lane) {
  const uint32_t ef = a.ef;
  const uint32_t hmask = a.hash_slots - 1;
  const uint32_t hash_limit = a.hash_slots - (a.hash_slots >> 2);
  HS_T0();
  while (true) {
    wave_sync();
    if (st.cand_size == 0) break;
    const Pair cp = cand[0];
    const float cd = unif(cp.d);
    const uint32_t cid = uni(cp.id);
    const bool stop = bare ? (cd > st.lb) : (cd > st.lb && st.top_size == ef);  // hnswalg_slim.h:340 / :346-347, :237
    if (stop) break;
    wave_sync();
    if (lane == 0) pop_heap(cand, (long)st.cand_size, GreaterD());  // :353-354
    st.cand_size--;
    c.n_hops++;
    HS_LAP(c, 0);
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
      if (c.n_ins + m > hash_limit) return 1;
      if (st.cand_size + m > a.cand_cap) return 2;
      uint32_t id = 0;
      bool isnew = false;
      if ((uint32_t)lane < m) id = ix.cols[base + lane];
#ifdef HS_PROFILE
      id = __shfl(id, lane, 64);  // force the load to land before the stamp
#endif
      HS_LAP(c, 1);
      if ((uint32_t)lane < m) isnew = hash_insert(hash, hmask, id);  // :392-393
      const unsigned long long nm = __ballot(isnew);
      const uint32_t cnt = __popcll(nm);
      c.n_nbr += m;
      if (cnt == 0) continue;
      wave_sync();
      if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
      wave_sync();
      c.n_ins += cnt;
      c.n_dist += cnt;
      HS_LAP(c, 2);
      wave_dists<METRIC>(ix, qv, nid, nd, cnt, lane);  // :395-396
      wave_sync();
      HS_LAP(c, 3);
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
      HS_LAP(c, 4);
    }
  }
  return 0;
}

template <int METRIC>
__device__ void search_one_strict(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;
  const StrictLds L = strict_layout(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  Pair *top = reinterpret_cast<Pair *>(smem + L.off_top);
  Pair *cand = reinterpret_cast<Pair *>(smem + L.off_cand);
  uint32_t *hash = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  uint32_t *nid = reinterpret_cast<uint32_t *>(smem + L.off_nid);
  float *nd = reinterpret_cast<float *>(smem + L.off_nd);
  const uint32_t k = a.k;
  Counters c;
#ifdef HS_PROFILE
  for (int i = 0; i < 8; i++) c.t[i] = 0;
#endif
  HS_T0();
  uint32_t cur;
  float curdist;
  descend<METRIC>(ix, a, qi, qv, hash, nid, nd, c, cur, curdist, lane);
  HS_LAP(c, 5);

  // ---- level-0 (and threshold-level) beams ----------------------------------------------------
  const bool bare = !ix.has_deleted;  // hnswalg_slim.h:2114, hnswalg.h:1421 (no filter on this path)
  const bool ep_deleted = ix.deleted[cur] != 0;
  if (ix.kind == 0) c.n_dist++;  // searchBaseLayerST recomputes the entry distance (hnswalg.h:351)
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
  c.n_ins++;
  SState st;
  st.cand_size = 1;
  if (ix.kind == 0 && !bare && ep_deleted) {
    st.top_size = 0;
    st.lb = FLT_MAX;
  } else {
    st.top_size = 1;
    st.lb = ep_deleted ? FLT_MAX : curdist;  // hnswalg_slim.h:2104-2106
  }
  int rc = 0;
  for (int lvl = min(ix.threshold_level, ix.maxlevel); lvl > 0 && rc == 0; lvl--) {  // hnswalg_slim.h:2108-2113
    rc = strict_beam<METRIC>(ix, a, lvl, /*bare=*/false, qv, top, cand, hash, nid, nd, st, c, lane);
    // next beam starts from candidate_set <- copy of top_candidates (+ make_heap) (:228-233, :327-332)
    wave_sync();
    if (lane == 0) {
      for (uint32_t i = 0; i < st.top_size; i++) cand[i] = top[i];
      make_heap(cand, (long)st.top_size, GreaterD());
    }
    st.cand_size = st.top_size;
  }
  if (rc == 0) rc = strict_beam<METRIC>(ix, a, 0, bare, qv, top, cand, hash, nid, nd, st, c, lane);
  wave_sync();
  if (rc != 0) {
    flag_query(a, qi, ST_OVERFLOW, rc - 1, lane);
    return;
  }
  HS_T0_RESET();
  // ---- raw result heap (parity/debug) ---------------------------------------------------------
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
    HS_LAP(c, 7);
    write_stats(a, qi, c);
    a.status[qi] = ST_DONE;
  }
}

// =================================================================================================
// fast kernel
// =================================================================================================
// Candidate min-heap in LDS, element i stored at slot i+1 so that the two children of any node share one
// 16-byte-aligned ds_read_b128.  Same sift decisions as std::push_heap/pop_heap with
// compare_by_first_rev (hnswalg_slim.h:177-183); executed by one lane.
__device__ __forceinline__ void cand_push(uint2 *h, uint32_t n /*size incl. new*/, float d, uint32_t id) {
  uint32_t hole = n - 1;
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const uint2 p = h[parent + 1];
    if (!(__uint_as_float(p.x) > d)) break;
    h[hole + 1] = p;
    hole = parent;
  }
  h[hole + 1] = make_uint2(__float_as_uint(d), id);
}
__device__ __forceinline__ void cand_pop(uint2 *h, uint32_t n /*size before pop*/) {
  if (n <= 1) return;
  const uint2 v = h[n];  // a[n-1]
  const uint32_t len = n - 1;
  uint32_t hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    const uint4 two = *reinterpret_cast<const uint4 *>(&h[child]);  // a[child-1], a[child]
    const bool left = __uint_as_float(two.z) > __uint_as_float(two.x);  // comp(a[child], a[child-1])
    h[hole + 1] = left ? make_uint2(two.x, two.y) : make_uint2(two.z, two.w);
    child = left ? child - 1 : child;
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h[hole + 1] = h[child];  // a[child-1]
    hole = child - 1;
  }
  const float vd = __uint_as_float(v.x);
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const uint2 p = h[parent + 1];
    if (!(__uint_as_float(p.x) > vd)) break;
    h[hole + 1] = p;
    hole = parent;
  }
  h[hole + 1] = v;
}

// Result set as a sorted (ascending distance) register array: rank r lives in lane r % 64, slot r / 64.
template <int S>
__device__ __forceinline__ float top_key_at(const float (&tk)[S], uint32_t r) {
  float v = 0.f;
#pragma unroll
  for (int s = 0; s < S; s++)
    if ((r >> 6) == (uint32_t)s) v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tk[s]), r & 63));
  return v;
}
template <int S>
__device__ __forceinline__ void top_insert(float (&tk)[S], uint32_t (&ti)[S], uint32_t &size, uint32_t ef, float d,
                                           uint32_t id, int lane) {
  uint32_t pos = 0;  // entries with key <= d stay in front (which equal-key entry is dropped is not defined here)
#pragma unroll
  for (int s = 0; s < S; s++) pos += __popcll(__ballot((uint32_t)(lane + 64 * s) < size && tk[s] <= d));
  uint32_t carry_k = 0, carry_i = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const uint32_t kb = __float_as_uint(tk[s]);
    const uint32_t last_k = __builtin_amdgcn_readlane(kb, 63), last_i = __builtin_amdgcn_readlane(ti[s], 63);
    const uint32_t up_k = wave_shr1(carry_k, kb), up_i = wave_shr1(carry_i, ti[s]);
    const uint32_t r = lane + 64 * s;
    tk[s] = r > pos ? __uint_as_float(up_k) : (r == pos ? d : tk[s]);
    ti[s] = r > pos ? up_i : (r == pos ? id : ti[s]);
    carry_k = last_k;
    carry_i = last_i;
  }
  size = min(size + 1, ef);
}

template <int METRIC, int S>
__device__ void search_one_fast(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;
  const FastLds L = fast_layout(ix.dim, a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  uint2 *cand = reinterpret_cast<uint2 *>(smem + L.off_cand);
  uint32_t *hash = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  uint32_t *nid = reinterpret_cast<uint32_t *>(smem + L.off_nid);
  float *nd = reinterpret_cast<float *>(smem + L.off_nd);
  const uint32_t k = a.k, ef = a.ef;
  const uint32_t hmask = a.hash_slots - 1;
  const uint32_t hash_limit = a.hash_slots - (a.hash_slots >> 2);
  Counters c;
#ifdef HS_PROFILE
  for (int i = 0; i < 8; i++) c.t[i] = 0;
#endif
  HS_T0();
  uint32_t cur;
  float curdist;
  descend<METRIC>(ix, a, qi, qv, hash, nid, nd, c, cur, curdist, lane);
  HS_LAP(c, 5);

  const bool bare = !ix.has_deleted;
  const bool ep_deleted = ix.deleted[cur] != 0;
  if (ix.kind == 0) c.n_dist++;  // hnswalg.h:351
  float tk[S];
  uint32_t ti[S];
#pragma unroll
  for (int s = 0; s < S; s++) { tk[s] = FLT_MAX; ti[s] = 0; }
  uint32_t top_size = 0, cand_size = 1;
  float lb;
  wave_sync();
  if (lane == 0) {
    hash_insert(hash, hmask, cur);  // hnswalg_slim.h:2102
    cand[1] = make_uint2(__float_as_uint((ix.kind == 0 && !bare && ep_deleted) ? FLT_MAX : curdist), cur);
  }
  c.n_ins++;
  if (ix.kind == 0 && !bare && ep_deleted) {  // hnswalg.h:359-362
    lb = FLT_MAX;
  } else {
    top_insert<S>(tk, ti, top_size, ef, curdist, cur, lane);  // hnswalg_slim.h:2100-2101
    lb = ep_deleted ? FLT_MAX : curdist;                       // :2104-2106
  }
  const uint32_t stride = ix.tile_stride;

  // ---- level-0 beam (hnswalg_slim.h:321-457) -----------------------------------------------------
  while (true) {
    wave_sync();
    if (cand_size == 0) break;
    const uint2 cp = cand[1];
    const float cd = unif(__uint_as_float(cp.x));
    const uint32_t cid = uni(cp.y);
    if (bare ? (cd > lb) : (cd > lb && top_size == ef)) break;  // :340 / :346-347
    // the popped node's whole level-0 list is one aligned tile: one coalesced read from the popped id
    uint32_t id = kNone;
    if ((uint32_t)lane < stride) id = ix.tile0[(size_t)cid * stride + lane];
    c.n_hops++;
    const bool valid = id != kNone;
    const uint32_t m = __popcll(__ballot(valid));
    HS_LAP(c, 1);
    if (c.n_ins + m > hash_limit) { flag_query(a, qi, ST_OVERFLOW, 0, lane); return; }
    if (cand_size + m > a.cand_cap) { flag_query(a, qi, ST_OVERFLOW, 1, lane); return; }
    bool isnew = false;
    if (valid) isnew = hash_insert(hash, hmask, id);  // :392-393
    const unsigned long long nm = __ballot(isnew);
    const uint32_t cnt = __popcll(nm);
    c.n_nbr += m;
    if (cnt == 0) {
      if (lane == 0) cand_pop(cand, cand_size);  // :353-354
      cand_size--;
      HS_LAP(c, 0);
      continue;
    }
    wave_sync();
    if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
    wave_sync();
    c.n_ins += cnt;
    c.n_dist += cnt;
    HS_LAP(c, 2);
    // row loads go out first; the candidate heap is re-heapified (:353-354) while they are in flight
    wave_dists<METRIC>(ix, qv, nid, nd, cnt, lane, [&]() {
      if (lane == 0) cand_pop(cand, cand_size);
    });  // :395-396
    cand_size--;
    wave_sync();
    HS_LAP(c, 3);
    // accept loop (:403-452) in adjacency order.  Once the result set is full lowerBound only decreases,
    // so entries that fail against the current bound can never pass later: skip them wholesale.
    const float my_d = (uint32_t)lane < cnt ? nd[lane] : FLT_MAX;
    const uint32_t my_id = (uint32_t)lane < cnt ? nid[lane] : 0;
    unsigned long long todo = __ballot((uint32_t)lane < cnt && (top_size < ef || lb > my_d));
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j));
      if (top_size < ef || lb > d) {  // :403-404
        const uint32_t nb = __builtin_amdgcn_readlane(my_id, j);
        cand_size++;
        if (lane == 0) cand_push(cand, cand_size, d, nb);  // :408-411
        if (bare || !ix.deleted[nb]) top_insert<S>(tk, ti, top_size, ef, d, nb, lane);  // :418-448
        if (top_size > 0) lb = top_key_at<S>(tk, top_size - 1);  // :450-452
      }
    }
    HS_LAP(c, 4);
  }
  HS_T0_RESET();
  // ---- k-selection: the k smallest distances; a tie across the k-th boundary makes the reference's
  //      choice depend on its heap layout (nth_element / pop_heap) -> strict kernel decides ----------
  if (top_size > k && top_key_at<S>(tk, k - 1) == top_key_at<S>(tk, k)) {
    flag_query(a, qi, ST_HAZARD, 2, lane);
    return;
  }
  const uint32_t valid_n = min(top_size, k);
#pragma unroll
  for (int s = 0; s < S; s++) {
    const uint32_t r = lane + 64 * s;
    if (r < k) {
      const bool v = r < valid_n;
      const uint64_t label = v ? ix.labels[ti[s]] : ~0ull;
      if (a.out_labels32) a.out_labels32[(size_t)qi * k + r] = v ? (uint32_t)label : 0xFFFFFFFFu;
      if (a.out_labels64) a.out_labels64[(size_t)qi * k + r] = label;
      if (a.out_dists) a.out_dists[(size_t)qi * k + r] = v ? tk[s] : __builtin_inff();
    }
  }
  if (lane == 0) {
    if (a.out_counts) a.out_counts[qi] = valid_n;
    HS_LAP(c, 7);
    write_stats(a, qi, c);
    a.status[qi] = ST_DONE;
  }
}

// ---- kernels: grid-stride over the queries selected by status ----------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(64) strict_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    if (!((1u << a.status[qi]) & a.select_mask)) continue;
    if (ix.n == 0) {  // cur_element_count == 0 (hnswalg_slim.h:2031-2032)
      if (threadIdx.x == 0) { if (a.out_counts) a.out_counts[qi] = 0; a.status[qi] = ST_DONE; }
      continue;
    }
    search_one_strict<METRIC>(ix, a, qi, smem);
    wave_sync();
  }
}
template <int METRIC, int S>
__global__ void __launch_bounds__(64) fast_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    if (!((1u << a.status[qi]) & a.select_mask)) continue;
    if (ix.n == 0) {
      if (threadIdx.x == 0) { if (a.out_counts) a.out_counts[qi] = 0; a.status[qi] = ST_DONE; }
      continue;
    }
    search_one_fast<METRIC, S>(ix, a, qi, smem);
    wave_sync();
  }
}

template <typename K>
static hipError_t launch(K kern, const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(std::max(1u, std::min(a.grid, a.nq))), dim3(64), lds, stream, ix, a);
  return hipGetLastError();
}

hipError_t launch_strict(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = strict_lds_bytes(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  return ix.metric == METRIC_L2 ? launch(strict_kernel<METRIC_L2>, ix, a, lds, stream)
                                : launch(strict_kernel<METRIC_IP>, ix, a, lds, stream);
}

template <int METRIC>
static hipError_t launch_fast_m(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.ef <= 64) return launch(fast_kernel<METRIC, 1>, ix, a, lds, stream);
  if (a.ef <= 128) return launch(fast_kernel<METRIC, 2>, ix, a, lds, stream);
  if (a.ef <= 256) return launch(fast_kernel<METRIC, 4>, ix, a, lds, stream);
  return launch(fast_kernel<METRIC, 8>, ix, a, lds, stream);
}
hipError_t launch_fast(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = fast_lds_bytes(ix.dim, a.cand_cap, a.hash_slots);
  return ix.metric == METRIC_L2 ? launch_fast_m<METRIC_L2>(ix, a, lds, stream) : launch_fast_m<METRIC_IP>(ix, a, lds, stream);
}

}  // namespace hs
