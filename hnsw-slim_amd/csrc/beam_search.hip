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
//   * adjacency: CSR in HBM (all levels) plus one aligned tile of ids per node for level 0 (an expansion is ONE
//     coalesced 64..256-byte read issued straight from the popped id) and {id, up_base} tiles for the upper levels
//     (a descent step is tile -> rows, two dependent accesses);
//   * visited set: bucketed hash of 32-bit ids in LDS (exact: a false positive would change results) with a second
//     tier in global memory; lanes insert their neighbour id with ds_cmpst, the ballot of "newly inserted" gives the
//     unvisited list in adjacency order;
//   * distances: 4 lanes per neighbour row, 16 rows per pass; lane `sub` loads the 16-byte chunk
//     16*s+4*sub of every 64-byte step and owns AVX-512 lane accumulators 4*sub..4*sub+3, so the fp32
//     sum is formed in exactly the reference's order (dist_recipe.hpp) -> bit-identical distances; rounds of eight
//     loads in flight per lane, compile-time dims for the common shapes; dim % 16 != 0 runs the reference's
//     SIMD4 / residual recipes one lane per row;
//   * candidate heap: the reference's raw array in LDS with libstdc++'s push_heap/pop_heap sift sequence
//     (push: the whole wave in one read/write round; pop: lane 0) -> the expansion order among equal-distance
//     candidates is the reference's;
//   * result set, two kernels:
//       strict: the reference's raw top_candidates array + libstdc++ heap/nth_element mechanics
//               (heap_emul.hpp) -> identical array, identical output ORDER, any tie pattern;
//       fast  : a sorted array held in registers (rank r in lane r%64); a tile's accept decisions are taken at once
//               from a closed form of the reference's sequential scan and the accepted entries merged in one pass;
//               which of several equal-distance entries survives/gets selected is then not defined, so for a query
//               whose k-subset could depend on it (k-th and (k+1)-th distance equal) the logged insertions are
//               replayed through the libstdc++ mechanics (or, if the log overflowed, the query is flagged ST_HAZARD
//               and answered by the strict kernel).  Output sorted by distance.
//   * a query that outgrows even the global-memory tiers is flagged ST_OVERFLOW and re-run by the strict kernel with
//     a whole CU's LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>

#include "dist_recipe.hpp"
#include "engine.hpp"
#include "heap_emul.hpp"
#include "search_common.hpp"
#include "wave_util.hpp"

namespace hs {


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
// Fast path: no result array in LDS (it lives in registers); heap element i sits at cand slot i+1.  When a
// tie has to be resolved the reference's result heap is rebuilt in the visited-set area, which is dead by then.
struct FastLds { uint32_t off_q, off_cand, off_hash, off_nid, off_nd, off_stage, total; };
__host__ __device__ inline FastLds fast_layout(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  FastLds l;
  l.off_q = 0;
  l.off_cand = align_up(dim * 4, 16);
  l.off_hash = l.off_cand + align_up((cand_cap + 2) * 8, 16);
  const uint32_t hash_bytes = hash_slots * 4 > (ef + 1) * 8 ? hash_slots * 4 : (ef + 1) * 8;
  l.off_nid = l.off_hash + align_up(hash_bytes, 16);
  l.off_nd = l.off_nid + 64 * 4;
  l.off_stage = l.off_nd + 64 * 4;              // ef x 8 B: result-set merge staging (batched accept)
  l.total = l.off_stage + align_up(ef * 8, 16);
  return l;
}
// The build compiles this file twice, side by side: -DHS_TU_METRIC=0 holds the L2 kernels and everything that is not
// per metric, -DHS_TU_METRIC=1 the inner-product kernels (the kernel instantiations are the build's critical path).
// Without the macro one translation unit holds both (make asm / prof / asan).
#if !defined(HS_TU_METRIC)
#define HS_TU_HAS_L2 1
#define HS_TU_HAS_IP 1
#elif HS_TU_METRIC == 0
#define HS_TU_HAS_L2 1
#define HS_TU_HAS_IP 0
#else
#define HS_TU_HAS_L2 0
#define HS_TU_HAS_IP 1
#endif
#if HS_TU_HAS_L2
size_t strict_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  return strict_layout(dim, ef, cand_cap, hash_slots).total;
}
size_t fast_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  return fast_layout(dim, ef, cand_cap, hash_slots).total;
}
bool fast_supported(const DevIndex &ix, uint32_t ef, uint32_t k) {
  // ef == k runs the boundary-watching variant, compiled for ef <= 128 only
  return ix.tile0 != nullptr && ix.threshold_level == 0 && (ef > k || (ef == k && ef <= 128 && !ix.has_deleted)) && ef <= 512;
}
#endif

struct Counters {
  uint32_t n_dist, n_hops, n_nbr;
#ifdef HS_PROFILE
  unsigned long long t[8];  // cycles: 0 pop, 1 adjacency, 2 visited, 3 distances, 4 accept, 5 upper, 6 init, 7 final
#endif
};
// Diagnostic build only (make prof): per-phase shader-clock stamps, written behind the stats block; the
// product build compiles these to nothing.
#ifdef HS_PROFILE
#define HS_T0() unsigned long long _t0 = clock64(); const unsigned long long _w0 = wall_clock64(); (void)_w0
#define HS_WALL(st) do { (st).t[6] = wall_clock64() - _w0; (st).t[7] = _w0; } while (0)
#define HS_T0_RESET() _t0 = clock64()
#define HS_LAP(st, i) do { unsigned long long _t1 = clock64(); (st).t[i] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define HS_T0() do {} while (0)
#define HS_WALL(st) do {} while (0)
#define HS_T0_RESET() do {} while (0)
#define HS_LAP(st, i) do {} while (0)
#endif

// Distances query -> rows nid[0..cnt), 16 rows per pass, 4 lanes per row; nd[j] receives the value.
// D16 = dim/16 when known at compile time (d=128 -> 8: eight 16-byte loads per lane, fully unrolled), 0 = runtime dim
// with dim % 16 == 0, -1 = any runtime dim (adds the recipes for dim % 16 != 0; those instantiations carry their cost),
// -dim/4 (< -1) = a compiled-in dim % 4 == 0 shape off the SIMD16 path (-25: d = 100).
// `between()` runs after the first pass's row loads have been ISSUED and before they are
// consumed: LDS-only work placed there (the candidate heap's pop) hides under the HBM latency.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// Loads in flight per lane and round for a compile-time dim.  Up to d = 128 the whole row; long rows (d >= 256: 1 KB and more per
// row) take 16 loads per round instead of 8 -- a hop over 3.8 KB rows (d = 960) is then 4 dependent load rounds instead of 8;
// those instantiations run at 2 wavefronts per SIMD (256 VGPRs), which is no loss where the bytes per hop, not the number of
// resident queries, set the pace.
__host__ __device__ constexpr int deep_buffer(int d16) {
  return d16 <= 0 ? 1 : d16 <= 8 ? d16 : d16 == 32 ? 32 : 16;   // (the query chunks of a round are read from LDS into registers too: 8 VGPRs per load)
}

// TO_REG: instead of writing nd[j], hand the value of row j to lane j in a register (one cross-lane move,
// no LDS write/read round trip); returned value is meaningful in lanes < cnt.
template <int METRIC, int D16 = 0, class Hook = NoHook, bool TO_REG = false>
__device__ __forceinline__ float wave_dists(const DevIndex &ix, const float *qv, const uint32_t *nid, float *nd,
                                            uint32_t cnt, int lane, Hook between = Hook()) {
  const int sub = lane & 3, grp = lane >> 2;
  float out = FLT_MAX;
  constexpr uint32_t kQuadDim = D16 < -1 ? (uint32_t)(-D16) * 4u : 0u;   // D16 = -dim/4: a compiled-in dim % 4 == 0 shape
  const uint32_t qdim = kQuadDim ? kQuadDim : ix.dim;
  if (D16 < 0 && (qdim & 15u) && !(qdim & 3u)) {
    // dim % 4 == 0: the reference's 4-lane recipes, 4 lanes per row and 16 rows per pass (wave_util.hpp quad_dist4)
    for (uint32_t base = 0; base < cnt; base += 16) {
      const uint32_t j = base + grp;
      const bool act = j < cnt;
      const uint32_t id = nid[act ? j : 0];
      const float r = quad_dist4<METRIC>(qv + sub, ix.vec + (size_t)id * qdim + sub, qdim, [&]() { if (base == 0) between(); });
      if (act && sub == 0) nd[j] = r;
    }
    return out;
  }
  if (D16 == -1 && (ix.dim & 15u)) {
    // other dims: the reference's residual / scalar recipes (dist_recipe.hpp l2_general / ip_general), one lane per row
    for (uint32_t base = 0; base < cnt; base += 64) {
      const uint32_t j = base + lane;
      if (j < cnt) {
        const float *row = ix.vec + (size_t)nid[j] * ix.dim;
        nd[j] = METRIC == METRIC_L2 ? l2_general(qv, row, ix.dim) : ip_general(qv, row, ix.dim);
      }
      if (base == 0) between();
    }
    return out;
  }
  if (D16 > 8) {
    // Long rows (d >= 256, compile-time dim): 8 lanes per row, 8 rows per pass.  A hop brings ~6 new rows; with 4 lanes per row
    // they occupy 24 of the 64 lanes and a 3.8 KB row (d = 960) takes four dependent load rounds, with 8 lanes per row 48 lanes
    // are busy and it takes two.  Lane s8 of a group owns AVX-512 lane accumulators 2 s8, 2 s8 + 1 and loads the 8 bytes at
    // 64 i + 8 s8 of every 64-byte step i, so every accumulator still sums its elements in the reference's order, and the
    // sixteen are combined in the reference's order (left to right for L2, space_l2.h:49-51; the halves tree of
    // _mm512_reduce_add_ps for IP, space_ip.h:197).
    const int s8 = lane & 7, g8 = lane >> 3;
    constexpr int B8 = D16 <= 32 ? (D16 > 0 ? D16 : 1) : (D16 % 30 == 0 ? 30 : D16 % 32 == 0 ? 32 : D16 % 24 == 0 ? 24 : 16);
    constexpr int R8 = (D16 > 0 ? D16 : 1) / B8, T8 = (D16 > 0 ? D16 : 1) % B8;
    const hs_f2 *q2 = reinterpret_cast<const hs_f2 *>(qv) + s8;
    for (uint32_t base = 0; base < cnt; base += 8) {
      const uint32_t j = base + g8;
      const bool act = j < cnt;
      const uint32_t id = nid[act ? j : base];   // idle groups re-read the pass's first row (cache hit) and discard
      const hs_f2 *row = reinterpret_cast<const hs_f2 *>(ix.vec + (size_t)id * (D16 * 16)) + s8;
      hs_f2 acc2 = {0.f, 0.f};
      hs_f2 buf[B8];
#pragma unroll 1
      for (int r = 0; r < R8; r++) {
#pragma unroll
        for (int i = 0; i < B8; i++) buf[i] = row[(r * B8 + i) * 8];
        if (base == 0 && r == 0) between();
#pragma unroll
        for (int i = 0; i < B8; i++) {
          const hs_f2 qe = q2[(r * B8 + i) * 8];
          if (METRIC == METRIC_L2) { const hs_f2 t = qe - buf[i]; const hs_f2 pp = t * t; acc2 = acc2 + pp; }
          else acc2 = __builtin_elementwise_fma(qe, buf[i], acc2);
        }
      }
      if (T8 > 0) {
#pragma unroll
        for (int i = 0; i < T8; i++) buf[i] = row[(R8 * B8 + i) * 8];
#pragma unroll
        for (int i = 0; i < T8; i++) {
          const hs_f2 qe = q2[(R8 * B8 + i) * 8];
          if (METRIC == METRIC_L2) { const hs_f2 t = qe - buf[i]; const hs_f2 pp = t * t; acc2 = acc2 + pp; }
          else acc2 = __builtin_elementwise_fma(qe, buf[i], acc2);
        }
      }
      float r;
      int own;
      if (METRIC == METRIC_L2) {
        r = acc2.x + acc2.y;
#pragma unroll
        for (int k = 1; k < 8; k++) {
          const float p = dpp_f<0x111>(r);   // row_shr:1 -- the running sum of the lane to the left (same 8-lane group)
          if (s8 == k) r = (p + acc2.x) + acc2.y;
        }
        own = 7;
      } else {
        float hx = acc2.x + dpp_f<0x104>(acc2.x), hy = acc2.y + dpp_f<0x104>(acc2.y);   // row_shl:4: accumulators j + 8
        hx = hx + dpp_f<0x102>(hx); hy = hy + dpp_f<0x102>(hy);                          // j + 4
        hx = hx + dpp_f<0x101>(hx); hy = hy + dpp_f<0x101>(hy);                          // j + 2
        r = 1.0f - (hx + hy);
        own = 0;
      }
      if (TO_REG) {
        const float got = __shfl(r, ((lane - (int)base) & 7) * 8 + own, 64);
        if ((uint32_t)lane >= base && (uint32_t)lane < base + 8 && (uint32_t)lane < cnt) out = got;
      } else {
        if (act && s8 == own) nd[j] = r;
      }
    }
    return out;
  }
  const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
  for (uint32_t base = 0; base < cnt; base += 16) {
    const uint32_t j = base + grp;
    const bool act = j < cnt;
    const uint32_t id = nid[act ? j : 0];  // idle groups re-read row 0 of the pass (cache hit) and discard
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (D16 > 0) {
      // compile-time dim: rounds of up to eight 16-byte loads per lane in flight, then the arithmetic of the round
      const float4 *row = reinterpret_cast<const float4 *>(ix.vec + (size_t)id * (D16 * 16)) + sub;
      constexpr int B = deep_buffer(D16);   // loads in flight per lane (D16 == 0 instantiates this branch too, dead)
      constexpr int R = D16 / B, T = D16 % B;                  // full rounds, tail
      float4 buf[B];
#pragma unroll
      for (int i = 0; i < B; i++) buf[i] = row[i * 4];
      if (base == 0) between();
#pragma unroll
      for (int i = 0; i < B; i++) step4<METRIC>(acc, qq[i * 4], buf[i]);
#pragma unroll 1
      for (int r = 1; r < R; r++) {
#pragma unroll
        for (int i = 0; i < B; i++) buf[i] = row[(r * B + i) * 4];
#pragma unroll
        for (int i = 0; i < B; i++) step4<METRIC>(acc, qq[(r * B + i) * 4], buf[i]);
      }
      if (T > 0) {
#pragma unroll
        for (int i = 0; i < T; i++) buf[i] = row[(R * B + i) * 4];
#pragma unroll
        for (int i = 0; i < T; i++) step4<METRIC>(acc, qq[(R * B + i) * 4], buf[i]);
      }
    } else {
      // runtime dim: the same rounds of up to eight loads in flight, trip counts known only at run time
      const uint32_t steps = ix.dim >> 4;
      const float4 *row = reinterpret_cast<const float4 *>(ix.vec + (size_t)id * ix.dim) + sub;
      for (uint32_t r0 = 0; r0 < steps; r0 += 8) {
        const uint32_t nb = min(8u, steps - r0);
        float4 buf[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
          if (i < nb) buf[i] = row[(r0 + i) * 4];
        if (base == 0 && r0 == 0) between();
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
          if (i < nb) step4<METRIC>(acc, qq[(r0 + i) * 4], buf[i]);
      }
    }
    bool owner;
    const float r = lane4_reduce<METRIC>(acc, sub, owner);
    if (TO_REG) {
      const float got = __shfl(r, ((lane - (int)base) & 15) * 4 + (METRIC == METRIC_L2 ? 3 : 0), 64);
      if ((uint32_t)lane >= base && (uint32_t)lane < base + 16 && (uint32_t)lane < cnt) out = got;
    } else {
      if (act && owner) nd[j] = r;
    }
  }
  return out;
}

// Shared prologue: stage the query, clear the visited set, entry distance, upper-layer greedy descent
// (hnswalg_slim.h:2033-2078, hnswalg.h:1385-1415).  Leaves (cur, curdist) = level-0 entry.
template <int METRIC, int D16 = 0>
__device__ __forceinline__ void descend(const DevIndex &ix, const SearchArgs &a, uint32_t qi, float *qv, Visited &vis,
                                        uint32_t *hash, uint32_t *nid, float *nd, Counters &c, uint32_t &cur,
                                        float &curdist, int lane) {
  for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
  vis_init(vis, a, qi, hash, lane);
  cur = ix.enterpoint;
  if (lane == 0) nid[0] = cur;
  wave_sync();
  wave_dists<METRIC, D16>(ix, qv, nid, nd, 1, lane);
  wave_sync();
  curdist = unif(nd[0]);
  c.n_dist = 1;
  c.n_hops = c.n_nbr = 0;
  if (a.mark_ep) vis_mark_one(vis, cur, a, lane);  // visited_array[enterpoint] = tag (hnswalg_slim.h:1919)
  uint32_t cur_b = ix.ep_base;   // up_base[cur], carried along with cur on the tiled path
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      c.n_hops++;
      if (ix.uptile) {
        // upper-level tile of (cur, lvl): {neighbour id, the neighbour's own up_base} pairs at a fixed stride -- the
        // step is tile -> rows (two dependent HBM accesses) instead of up_base -> up_ptr -> ids -> rows (four)
        if (cur_b == kNone) continue;
        uint2 pr = make_uint2(kNone, kNone);
        if ((uint32_t)lane < ix.up_stride) pr = ix.uptile[(size_t)(cur_b + lvl - 1) * ix.up_stride + lane];
        const uint32_t m = __popcll(hs_ballot(pr.x != kNone));   // ids are a prefix of the tile
        if (m == 0) continue;
        wave_sync();
        if ((uint32_t)lane < m) nid[lane] = pr.x;
        wave_sync();
        wave_dists<METRIC, D16>(ix, qv, nid, nd, m, lane);
        wave_sync();
        c.n_nbr += m;
        c.n_dist += m;
        const float mine = (uint32_t)lane < m ? nd[lane] : FLT_MAX;
        const float d = wave_min_f32(mine);
        const uint32_t l = (uint32_t)__ffsll((long long)hs_ballot((uint32_t)lane < m && mine == d)) - 1;
        if (l < m && d < curdist) {  // hnswalg_slim.h:2071-2075
          curdist = d;
          cur = __builtin_amdgcn_readlane(pr.x, l);
          cur_b = __builtin_amdgcn_readlane(pr.y, l);
          changed = true;
        }
        continue;
      }
      const uint32_t b = uni(ix.up_base[cur]);
      if (b == kNone) continue;
      const uint32_t s = uni(ix.up_ptr[b + lvl - 1]), e = uni(ix.up_ptr[b + lvl]);
      for (uint32_t base = s; base < e; base += 64) {
        const uint32_t m = min(64u, e - base);
        wave_sync();
        if ((uint32_t)lane < m) nid[lane] = ix.cols[base + lane];
        wave_sync();
        wave_dists<METRIC, D16>(ix, qv, nid, nd, m, lane);
        wave_sync();
        c.n_nbr += m;
        c.n_dist += m;
        // first index attaining the minimum == what the sequential `if (d < curdist)` scan ends on
        const float mine = (uint32_t)lane < m ? nd[lane] : FLT_MAX;
        const float d = wave_min_f32(mine);
        const uint32_t l = (uint32_t)__ffsll((long long)hs_ballot((uint32_t)lane < m && mine == d)) - 1;
        if (l < m && d < curdist) {  // hnswalg_slim.h:2071-2075
          curdist = d;
          cur = uni(nid[l]);
          changed = true;
        }
      }
    }
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
                           Pair *cand, Visited &vis, uint32_t *nid, float *nd, SState &st, Counters &c, int lane) {
  const uint32_t ef = a.ef;
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
      const uint32_t b = uni(ix.up_base[cid]);
      if (b == kNone) continue;  // neighbors == nullptr (:247-249)
      s = ix.up_ptr[b + level - 1];
      e = ix.up_ptr[b + level];
    }
    s = uni(s);
    e = uni(e);
    for (uint32_t base = s; base < e; base += 64) {
      const uint32_t m = min(64u, e - base);
      if (!vis_reserve(vis, m, a, lane)) return 1;
      if (st.cand_size + m > a.cand_cap) return 2;
      uint32_t id = 0;
      bool isnew = false;
      if ((uint32_t)lane < m) id = ix.cols[base + lane];
#ifdef HS_PROFILE
      id = __shfl(id, lane, 64);  // force the load to land before the stamp
#endif
      HS_LAP(c, 1);
      if ((uint32_t)lane < m) isnew = vis_insert(vis, id);  // :392-393
      const unsigned long long nm = hs_ballot(isnew);
      const uint32_t cnt = __popcll(nm);
      c.n_nbr += m;
      if (cnt == 0) continue;
      wave_sync();
      if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
      wave_sync();
      vis_commit(vis, cnt);
      c.n_dist += cnt;
      HS_LAP(c, 2);
      wave_dists<METRIC, -1>(ix, qv, nid, nd, cnt, lane);  // :395-396
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
  const StrictLds L = strict_layout(ix.dim, a.ef, a.fb_cand ? 0u : a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  Pair *top = reinterpret_cast<Pair *>(smem + L.off_top);
  Pair *cand = a.fb_cand ? reinterpret_cast<Pair *>(a.fb_cand) + (size_t)blockIdx.x * a.cand_cap   // last-resort pass: the heap in global memory
                         : reinterpret_cast<Pair *>(smem + L.off_cand);
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
  Visited vis;
  descend<METRIC, -1>(ix, a, qi, qv, vis, hash, nid, nd, c, cur, curdist, lane);
  HS_LAP(c, 5);

  // ---- level-0 (and threshold-level) beams ----------------------------------------------------
  const bool bare = !ix.has_deleted;  // hnswalg_slim.h:2114, hnswalg.h:1421 (no filter on this path)
  const bool ep_deleted = uni(ix.deleted[cur]) != 0;
  if (ix.kind == 0 && (bare || !ep_deleted)) c.n_dist++;  // searchBaseLayerST recomputes the entry distance (hnswalg.h:347-351)
  wave_sync();
  if (lane == 0) {
    vis_insert(vis, cur);  // visited_array[currObj] = tag (hnswalg_slim.h:2102)
    if (ix.kind == 0 && !bare && ep_deleted) {  // hnswalg.h:359-362
      cand[0].d = FLT_MAX;
      cand[0].id = cur;
    } else {
      top[0].d = curdist;  // hnswalg_slim.h:2100-2101
      top[0].id = cur;
      cand[0] = top[0];
    }
  }
  vis.n1++;
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
    rc = strict_beam<METRIC>(ix, a, lvl, /*bare=*/false, qv, top, cand, vis, nid, nd, st, c, lane);
    // next beam starts from candidate_set <- copy of top_candidates (+ make_heap) (:228-233, :327-332)
    wave_sync();
    if (lane == 0) {
      for (uint32_t i = 0; i < st.top_size; i++) cand[i] = top[i];
      make_heap(cand, (long)st.top_size, GreaterD());
    }
    st.cand_size = st.top_size;
  }
  if (rc == 0) rc = strict_beam<METRIC>(ix, a, 0, bare, qv, top, cand, vis, nid, nd, st, c, lane);
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
  for (int s = 0; s < S; s++) pos += __popcll(hs_ballot((uint32_t)(lane + 64 * s) < size && tk[s] <= d));
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

// FLAT (indexes without delete marks / filter; not the ef == k variant): the level-0 search starts WITHOUT a candidate heap.
// With nothing deleted every accepted neighbour enters the result set, so candidate_set \ top_candidates only ever holds
// entries the result set has evicted -- entries at or beyond lowerBound, which the loop condition (:340) never lets the
// reference expand unless they tie with it.  The node to expand next is then the nearest entry of the result set that has
// not been expanded: one flag bit per entry (bit 31 of its id) and a ballot replace the heap's pushes and pop, a third of
// what an expansion costs a wavefront.  What the heap's LAYOUT decides in the reference is only the order among candidates
// at EQUAL distance; the two ways that can matter are watched -- (a) the chosen node shares its distance with another
// unexpanded entry, (b) an entry is evicted at exactly the new lowerBound (still expandable in the reference) -- and when
// one shows, the reference's heap is materialised as it stands at that moment: the logged result-set insertions are its
// pushes, one byte per expansion says how many belong to each, and they are replayed with the pops in between through the
// same libstdc++ mechanics; the search then continues on the heap path.  Continuous data never leaves the flat path, integer
// data (SIFT) does in about a quarter of the queries at ef=70, and pays then what the heap would have cost it anyway.
template <int METRIC, int S, int D16, bool WB = false, bool BARE = true, bool FLAT = false>
__device__ int search_one_fast(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, unsigned char *smem) {
  static_assert(!FLAT || (BARE && !WB), "the flat start needs an index without delete marks and ef > k");
  const int lane = threadIdx.x;
  const FastLds L = fast_layout(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  CandHeap cand;
  cand.lds = reinterpret_cast<uint2 *>(smem + L.off_cand);
  cand.L = a.cand_cap + 2;  // cand_cap is even
  cand.glob = reinterpret_cast<uint2 *>(a.spill + (size_t)qi * a.spill_stride + a.spill_slots);
  const uint32_t cand_total = a.spill ? a.cand_cap + a.cand2_cap : a.cand_cap;
  // insertion log of the result set (the sequence of push_heap calls the reference makes, hnswalg_slim.h:418-423):
  // 8 bytes per accepted neighbour, written by lane 0 and read back only if a tie must be resolved
  uint2 *tlog = a.spill ? reinterpret_cast<uint2 *>(a.spill + (size_t)qi * a.spill_stride + a.spill_slots + 2 * a.cand2_cap) : nullptr;
  uint32_t n_log = 0;
  uint32_t *hash = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  uint32_t *nid = reinterpret_cast<uint32_t *>(smem + L.off_nid);
  float *nd = reinterpret_cast<float *>(smem + L.off_nd);
  const uint32_t k = a.k, ef = a.ef;
  Counters c;
#ifdef HS_PROFILE
  for (int i = 0; i < 8; i++) c.t[i] = 0;
#endif
  HS_T0();
  uint32_t cur;
  float curdist;
  Visited vis;
  if (a.phase == 2) {
    // the descent ran in an earlier launch (phase 1): take the level-0 entry from there
    for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
    vis_init(vis, a, qi, hash, lane);
    const uint4 e = a.entry[qi];
    cur = uni(e.x);
    curdist = unif(__uint_as_float(e.y));
    c.n_dist = uni(e.z);
    c.n_hops = uni(e.w);
    c.n_nbr = c.n_dist - 1;   // the descent evaluates every neighbour it reads, plus the enter point
    if (a.mark_ep) {  // hnswalg_slim.h:1919
      wave_sync();
      vis_mark_one(vis, ix.enterpoint, a, lane);
    }
    wave_sync();
  } else {
    descend<METRIC, D16>(ix, a, qi, qv, vis, hash, nid, nd, c, cur, curdist, lane);
    if (a.phase == 1) {
      if (lane == 0) a.entry[qi] = make_uint4(cur, __float_as_uint(curdist), c.n_dist, c.n_hops);
      return 0;
    }
  }
  HS_LAP(c, 5);

  constexpr bool bare = BARE;   // no delete marks / filter anywhere in the index (own instantiation, see WB)
  const bool ep_deleted = uni(ix.deleted[cur]) != 0;
  if (ix.kind == 0 && (bare || !ep_deleted)) c.n_dist++;  // hnswalg.h:347-351
  float tk[S];
  uint32_t ti[S];
#pragma unroll
  for (int s = 0; s < S; s++) { tk[s] = FLT_MAX; ti[s] = 0; }
  uint32_t top_size = 0;
  float lb;
  // the node to expand next and its distance == candidate_set[0] once every pending push is applied
  float next_d = (ix.kind == 0 && !bare && ep_deleted) ? FLT_MAX : curdist;
  uint32_t next_id = cur;
  uint32_t cand_size = 0;            // heap entries physically in LDS
  unsigned long long pending = 1ull; // accepted entries of nd/nid still to be pushed (bit j = entry j)
  // accepted entries of the previous expansion, lane j = entry j, until they are pushed (see the loop)
  float prev_d = next_d;
  uint32_t prev_id = cur;
  wave_sync();
  const bool entry_marked = vis_mark_one(vis, cur, a, lane);  // hnswalg_slim.h:2102
  if (ix.kind == 0 && !bare && ep_deleted) {  // hnswalg.h:359-362
    lb = FLT_MAX;
  } else {
    top_insert<S>(tk, ti, top_size, ef, curdist, cur, lane);  // hnswalg_slim.h:2100-2101
    if (tlog && lane == 0) tlog[0] = make_uint2(__float_as_uint(curdist), cur);
    n_log = 1;
    lb = ep_deleted ? FLT_MAX : curdist;                       // :2104-2106
  }
  const uint32_t stride = ix.tile_stride;
  int rc = entry_marked ? 0 : 1;
  // ef == k: nothing is selected at the end, so a tie ACROSS the capacity boundary decides the answer -- the reference
  // evicts the root of its heap, i.e. one of several entries with the largest key, which one depends on the heap layout.
  // Whenever an evicted key equals the last kept key the logged insertions are replayed (as for the k / k+1 tie).
  // (A separate instantiation: even dormant, this bookkeeping costs the hot kernel 4-5 % -- measured.)
  constexpr bool watch_boundary = WB;
  bool boundary_tie = false;

  // ---- accept decisions (:403-452) for the whole tile at once (indexes without delete marks) -------------------------
  // Returns the number of accepted neighbours; `pending` = their lanes; drop_tie: an entry left the set at exactly the new bound.
  uint32_t last_acc = 0;
  bool drop_tie = false;
  auto accept_bare = [&](const float my_d, const uint32_t my_id, const uint32_t cnt, const bool want_drop) {
    last_acc = 0;
    drop_tie = false;
      // ---- accept decisions (:403-452) for the whole tile at once --------------------------------------------
      // The reference scans the new neighbours in adjacency order, accepting j iff top_size < ef || lowerBound > d_j
      // with the result set updated after every acceptance.  Equivalent closed form: with T the result set before
      // this tile, j is accepted iff  #{t in T : t <= d_j} + #{i < j : d_i <= d_j}  <  ef  (every earlier neighbour
      // that is not farther is itself accepted whenever j is, and evicted entries only ever lie beyond the ef-th
      // rank).  Accepted entries then merge into the sorted set in one pass: an old entry moves up by the number of
      // accepted keys strictly below it, an accepted one lands at #{T <= d} + #{accepted before it in (d, j) order}.
      // Same set, same lowerBound sequence as far as any decision can see, a third of the vector instructions.
      const bool cand_ok = (uint32_t)lane < cnt && (top_size < ef || lb > my_d);
      const unsigned long long pm = hs_ballot(cand_ok);
      if (pm) {
        uint32_t A = 0, B = 0;
        for (unsigned long long m = pm; m; m &= m - 1) {
          const int j = __ffsll((long long)m) - 1;
          const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j));
          uint32_t a = 0;
#pragma unroll
          for (int s = 0; s < S; s++) a += __popcll(hs_ballot((uint32_t)(lane + 64 * s) < top_size && tk[s] <= dj));
          A = write_lane(A, a, j);
          B += (cand_ok && lane > j && dj <= my_d) ? 1u : 0u;
        }
        const bool acc = cand_ok && (A + B < ef);
        const unsigned long long am = hs_ballot(acc);
        const uint32_t n_acc = __popcll(am);
        uint32_t Bp = 0, shift[S];
#pragma unroll
        for (int s = 0; s < S; s++) shift[s] = 0;
        for (unsigned long long m = am; m; m &= m - 1) {
          const int j = __ffsll((long long)m) - 1;
          const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j));
          Bp += (acc && (dj < my_d || (dj == my_d && j < lane))) ? 1u : 0u;
#pragma unroll
          for (int s = 0; s < S; s++) shift[s] += tk[s] > dj ? 1u : 0u;
        }
        uint2 *stage = reinterpret_cast<uint2 *>(smem + L.off_stage);
        const uint32_t old_size = top_size;
        float dropped = FLT_MAX;   // smallest key pushed beyond the capacity (only needed when ef == k)
        if (watch_boundary || want_drop) {
          if (acc && A + Bp >= ef) dropped = my_d;
#pragma unroll
          for (int s = 0; s < S; s++) {
            const uint32_t r = lane + 64 * s;
            if (r < top_size && r + shift[s] >= ef) dropped = fminf(dropped, tk[s]);
          }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
          const uint32_t r = lane + 64 * s, nr = r + shift[s];
          if (r < top_size && nr < ef) stage[nr] = make_uint2(__float_as_uint(tk[s]), ti[s]);
        }
        if (acc && A + Bp < ef) stage[A + Bp] = make_uint2(__float_as_uint(my_d), my_id);
        wave_sync();
        top_size = min(top_size + n_acc, ef);
#pragma unroll
        for (int s = 0; s < S; s++) {
          const uint32_t r = lane + 64 * s;
          if (r < top_size) {
            const uint2 e = stage[r];
            tk[s] = __uint_as_float(e.x);
            ti[s] = e.y;
          }
        }
        wave_sync();
        lb = top_key_at<S>(tk, top_size - 1);  // :450-452
        if ((watch_boundary || want_drop) && old_size + n_acc > ef) drop_tie = hs_ballot(dropped == lb) != 0;   // (dropped keys are >= lb: the smallest equals it iff one does)
        if (watch_boundary) boundary_tie = boundary_tie || drop_tie;
        pending = am;
        last_acc = n_acc;
        if (am) {
          if (__builtin_expect(tlog != nullptr, 1) && acc) {
            const uint32_t at = n_log + __popcll(am & ((1ull << lane) - 1ull));
            if (at < a.log_cap) tlog[at] = make_uint2(__float_as_uint(my_d), my_id);
          }
          n_log += n_acc;
        }
      }
  };

  // ---- level-0 beam (hnswalg_slim.h:321-457) -----------------------------------------------------
  // The reference pushes accepted neighbours into candidate_set one by one and pops its root at the top of the next
  // iteration.  Which entry that root will be does not need the accept pass at all: a pushed entry only rises past
  // strictly larger parents, so it is the earliest neighbour with the smallest distance -- accepted iff it passes the
  // bound as it stood before this expansion, every earlier neighbour being strictly farther -- if that beats the old
  // root, else the old root.  So one expansion runs as
  //   visited lookups | row loads, under them: last expansion's pushes, then the pop (:353-354), on the LDS heap |
  //   distances | next node chosen, its adjacency tile requested | accept pass (:403-452) under that read.
  // Neither HBM round trip of an expansion waits for heap work and the heap work waits for neither.
  uint32_t id = kNone;   // adjacency tile of the node to expand: one aligned, coalesced read from its id
  if (rc == 0 && (uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];

  // ---- the same beam without the heap, until a tie (see FLAT above) ---------------------------------------------------
  constexpr uint32_t kDone = 0x80000000u;   // bit 31 of a result-set id: that entry has been expanded
  uint8_t *hoplog = (FLAT && a.spill && a.hop_cap) ? reinterpret_cast<uint8_t *>(a.spill + (size_t)qi * a.spill_stride + a.spill_slots + 2 * a.cand2_cap + 2 * a.log_cap) : nullptr;
  bool flat = FLAT && a.flat != 0 && hoplog != nullptr && tlog != nullptr && ix.n < kDone && rc == 0;
  bool flat_done = false;     // the search ended on the flat path
  if (FLAT && flat) {
    uint32_t hops_flat = 0;
    if (lane == 0) ti[0] |= kDone;   // the entry is the one element of the set and the node being expanded
    while (true) {
      if (__builtin_expect(next_d > lb, 0)) { flat_done = true; break; }   // :340 (next is the nearest unexpanded entry)
      c.n_hops++;
      const bool valid = id != kNone;
      const uint32_t m = __popcll(hs_ballot(valid));
      bool isnew = false;
      if (vis.qbits) {
        bool fail = false;
        isnew = vis_test_and_mark_q16(vis, id, valid, a, lane, fail);  // :392-393
        if (__builtin_expect(fail, 0)) { rc = 1; break; }
      } else {
        if (__builtin_expect(!vis_reserve(vis, m, a, lane), 0)) { rc = 1; break; }
        if (valid) isnew = vis_insert(vis, id);  // :392-393
      }
      const unsigned long long nm = hs_ballot(isnew);
      const uint32_t cnt = __popcll(nm);
      c.n_nbr += m;
      wave_sync();
      if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
      wave_sync();
      if (!vis.qbits) vis_commit(vis, cnt);
      c.n_dist += cnt;
      if (__builtin_expect(cnt > 0, 1)) wave_dists<METRIC, D16>(ix, qv, nid, nd, cnt, lane);  // :395-396
      wave_sync();
      const float my_d = (uint32_t)lane < cnt ? nd[lane] : FLT_MAX;
      const uint32_t my_id = (uint32_t)lane < cnt ? nid[lane] : 0;
      // nearest unexpanded entry of the set as it stands (the old "root"), against the nearest new neighbour
      bool have_next = false;
#pragma unroll
      for (int s = 0; s < S; s++) {
        const unsigned long long um = hs_ballot((uint32_t)(lane + 64 * s) < top_size && ti[s] < kDone);
        if (!have_next && um) {
          const int p = __ffsll((long long)um) - 1;
          next_d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(tk[s]), p));
          next_id = __builtin_amdgcn_readlane(ti[s], p);
          have_next = true;
        }
      }
      const float best_d = wave_min_f32(my_d);
      const bool have_best = cnt > 0 && (top_size < ef || lb > best_d);
      if (have_best && (!have_next || best_d < next_d)) {
        const int bl = __ffsll((long long)hs_ballot((uint32_t)lane < cnt && my_d == best_d)) - 1;
        next_d = best_d;
        next_id = __builtin_amdgcn_readlane(my_id, bl);
        have_next = true;
      }
      id = kNone;
      if (have_next && (uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];
      pending = 0;
      accept_bare(my_d, my_id, cnt, true);   // pending = the accepted lanes
      prev_d = my_d;
      prev_id = my_id;
      if (__builtin_expect(hops_flat >= a.hop_cap || n_log > a.log_cap, 0)) { rc = 3; break; }   // cannot be replayed: strict pass
      if (lane == 0) hoplog[hops_flat] = (uint8_t)last_acc;
      hops_flat++;
      if (!have_next) { flat_done = true; break; }
      // the chosen node among the unexpanded entries of the set as it is now: alone at its distance?
      uint32_t same = 0;
#pragma unroll
      for (int s = 0; s < S; s++) same += __popcll(hs_ballot((uint32_t)(lane + 64 * s) < top_size && ti[s] < kDone && tk[s] == next_d));
      if (__builtin_expect(drop_tie || same > 1, 0)) {
        // ---- materialise candidate_set as the reference has it now: entry pushed; per expansion: pop, then that expansion's
        //      accepted entries pushed -- all but the last expansion's, which are still pending (prev_d / prev_id / pending)
        __threadfence_block();
        cand_size = 0;
        uint32_t idx = 0, base = 0;
        uint2 e = make_uint2(0, 0);
        uint32_t hl = 0;
        for (uint32_t h = 0; h <= hops_flat; h++) {
          // pushes of "expansion h-1" (h == 0: the entry), then the pop of expansion h
          const uint32_t np = h == 0 ? 1u : (uint32_t)((__builtin_amdgcn_readlane(hl, (h - 1) & 63)));
          if (h == hops_flat) break;   // (the last expansion's accepted entries stay pending)
          if (__builtin_expect(cand_size + np > cand_total, 0)) { rc = 2; break; }
          for (uint32_t q = 0; q < np; q++) {
            if ((idx & 63u) == 0) { base = idx; e = (base + lane < n_log) ? tlog[base + lane] : make_uint2(0, 0); }
            const uint32_t dx = __builtin_amdgcn_readlane(e.x, idx & 63u), ix_ = __builtin_amdgcn_readlane(e.y, idx & 63u);
            cand_size++;
            cand_push(cand, cand_size, __uint_as_float(dx), ix_, lane);
            wave_sync();
            idx++;
          }
          if ((h & 63u) == 0) hl = (h + lane < hops_flat) ? (uint32_t)hoplog[h + lane] : 0u;
          if (lane == 0) cand_pop(cand, cand_size);
          cand_size--;
          wave_sync();
        }
#pragma unroll
        for (int s = 0; s < S; s++) ti[s] &= ~kDone;
        // the node the heap path expands next: nearest pending entry if strictly nearer than the heap's root, else the root
        const float pd = (pending >> lane) & 1ull ? my_d : FLT_MAX;
        const float pbest = wave_min_f32(pd);
        bool hn = pending != 0;
        if (cand_size > 0) {
          const uint2 root = cand.lds[1];
          next_d = unif(__uint_as_float(root.x));
          next_id = uni(root.y);
          hn = true;
        }
        if (pending != 0 && (cand_size == 0 || pbest < next_d)) {
          const int bl = __ffsll((long long)hs_ballot(((pending >> lane) & 1ull) && my_d == pbest)) - 1;
          next_d = pbest;
          next_id = __builtin_amdgcn_readlane(my_id, bl);
        }
        id = kNone;
        if (hn && (uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];
        break;
      }
      // flag the chosen node expanded where it sits in the set
#pragma unroll
      for (int s = 0; s < S; s++)
        if ((uint32_t)(lane + 64 * s) < top_size && ti[s] == next_id) ti[s] |= kDone;
    }
    if (flat_done || rc != 0) {
#pragma unroll
      for (int s = 0; s < S; s++) ti[s] &= ~kDone;
    }
  }
  while (rc == 0 && !flat_done) {
    if (__builtin_expect(cand_size == 0 && pending == 0, 0)) break;
    if (__builtin_expect(bare ? (next_d > lb) : (next_d > lb && top_size == ef), 0)) break;  // :340 / :346-347
    c.n_hops++;
    HS_LAP(c, 0);
    const bool valid = id != kNone;
    const uint32_t m = __popcll(hs_ballot(valid));
    HS_LAP(c, 1);
    if (__builtin_expect(cand_size + (uint32_t)__popcll(pending) > cand_total, 0)) { rc = 2; break; }   // room for this round's pushes
    bool isnew = false;
    if (vis.qbits) {
      bool fail = false;
      isnew = vis_test_and_mark_q16(vis, id, valid, a, lane, fail);  // :392-393
      if (__builtin_expect(fail, 0)) { rc = 1; break; }
    } else {
      if (__builtin_expect(!vis_reserve(vis, m, a, lane), 0)) { rc = 1; break; }
      if (valid) isnew = vis_insert(vis, id);  // :392-393
    }
    const unsigned long long nm = hs_ballot(isnew);
    const uint32_t cnt = __popcll(nm);
    c.n_nbr += m;
    wave_sync();
    if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;  // unvisited ids, adjacency order
    wave_sync();
    if (!vis.qbits) vis_commit(vis, cnt);
    c.n_dist += cnt;
    HS_LAP(c, 2);
    // row loads go out first; the pushes of the previous expansion (:408-411, adjacency order) and pop_heap (:353-354)
    // work on the LDS heap while they are in flight
    auto heap_hook = [&]() {
      while (pending) {
        const int j = __ffsll((long long)pending) - 1;
        pending &= pending - 1;
        cand_size++;
        cand_push(cand, cand_size, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(prev_d), j)),
                  __builtin_amdgcn_readlane(prev_id, j), lane);
        wave_sync();
      }
      if (lane == 0) cand_pop(cand, cand_size);
    };
    if (__builtin_expect(cnt > 0, 1)) {
      wave_dists<METRIC, D16>(ix, qv, nid, nd, cnt, lane, heap_hook);  // :395-396
    } else {
      heap_hook();
    }
    cand_size--;
    wave_sync();
    HS_LAP(c, 3);
    const float my_d = (uint32_t)lane < cnt ? nd[lane] : FLT_MAX;
    const uint32_t my_id = (uint32_t)lane < cnt ? nid[lane] : 0;
    {
      // root of candidate_set once this expansion's pushes are in: the nearest new neighbour if it is accepted and
      // strictly nearer than the heap's root, else that root
      const float best_d = wave_min_f32(my_d);
      const bool have_best = cnt > 0 && (top_size < ef || lb > best_d);
      bool have_next = have_best;
      if (cand_size > 0) {
        const uint2 root = cand.lds[1];
        next_d = unif(__uint_as_float(root.x));
        next_id = uni(root.y);
        have_next = true;
      }
      if (have_best && (cand_size == 0 || best_d < next_d)) {
        const int bl = __ffsll((long long)hs_ballot((uint32_t)lane < cnt && my_d == best_d)) - 1;
        next_d = best_d;
        next_id = __builtin_amdgcn_readlane(my_id, bl);
      }
      id = kNone;
      if (have_next && (uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];
    }
    if (__builtin_expect(bare, 1)) {
      accept_bare(my_d, my_id, cnt, false);
    } else {
    // accept decisions (:403-452) in adjacency order.  Once the result set is full lowerBound only
    // decreases, so entries that fail against the current bound can never pass later: skip them wholesale.
    unsigned long long todo = hs_ballot((uint32_t)lane < cnt && (top_size < ef || lb > my_d));
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j));
      if (top_size < ef || lb > d) {  // :403-404
        const uint32_t nb = __builtin_amdgcn_readlane(my_id, j);
        pending |= 1ull << j;
        if (bare || uni(ix.deleted[nb]) == 0) {
          const bool evicts = top_size == ef;
          const float evicted = evicts ? top_key_at<S>(tk, ef - 1) : 0.f;   // d < this key (strict), so it is the one dropped
          top_insert<S>(tk, ti, top_size, ef, d, nb, lane);  // :418-448
          if (watch_boundary && evicts) boundary_tie = boundary_tie || top_key_at<S>(tk, ef - 1) == evicted;
          if (tlog && lane == 0 && n_log < a.log_cap) tlog[n_log] = make_uint2(__float_as_uint(d), nb);
          n_log++;
        }
        if (top_size > 0) lb = top_key_at<S>(tk, top_size - 1);  // :450-452
      }
    }
    }  // !bare
    prev_d = my_d;
    prev_id = my_id;
    HS_LAP(c, 4);
  }
  if (__builtin_expect(rc != 0, 0)) {
    flag_query(a, qi, ST_OVERFLOW, rc - 1, lane);
    return rc;
  }
  HS_T0_RESET();
  // ---- k-selection: the k smallest distances.  A tie across the k-th boundary makes the reference's choice
  //      depend on the LAYOUT of its result heap (nth_element / pop_heap pick among equal keys by position).
  //      The traversal above is already the reference's, so its heap is rebuilt exactly by replaying the logged
  //      insertions through libstdc++'s push_heap/pop_heap mechanics -- in the visited-set area, dead by now.
  if (__builtin_expect((top_size > k && top_key_at<S>(tk, k - 1) == top_key_at<S>(tk, k)) || boundary_tie, 0)) {
    if (lane == 0) atomicAdd(a.counters + 2, 1u);
    if (!tlog || n_log > a.log_cap) return 3;  // log did not fit: the strict kernel re-runs the query
    Pair *top = reinterpret_cast<Pair *>(hash);
    __threadfence_block();
    uint32_t ts = 0;
    for (uint32_t base = 0; base < n_log; base += 64) {
      const uint32_t m = min(64u, n_log - base);
      uint2 e = make_uint2(0, 0);
      if ((uint32_t)lane < m) e = tlog[base + lane];
      for (uint32_t j = 0; j < m; j++) {
        const float d = __uint_as_float(__builtin_amdgcn_readlane(e.x, j));
        const uint32_t nb = __builtin_amdgcn_readlane(e.y, j);
        if (lane == 0) {
          top[ts].d = d;  // hnswalg_slim.h:419-423
          top[ts].id = nb;
          push_heap(top, (long)ts + 1, LessD());
          if (ts + 1 > ef) pop_heap(top, (long)ts + 1, LessD());  // :434-448
        }
        ts = min(ts + 1, ef);
      }
    }
    if (lane == 0) {
      if (a.mode == 0) {
        nth_element(top, (long)k, (long)ts, LessD());  // hnswalg_slim.h:2126-2127
      } else {
        uint32_t t2 = ts;
        while (t2 > k) { pop_heap(top, (long)t2, LessD()); t2--; }  // :2019-2022
      }
    }
    wave_sync();
    for (uint32_t i = lane; i < k; i += 64) {
      const Pair p = top[i];
      const uint64_t label = ix.labels[p.id];
      if (a.out_labels32) a.out_labels32[(size_t)qi * k + i] = (uint32_t)label;
      if (a.out_labels64) a.out_labels64[(size_t)qi * k + i] = label;
      if (a.out_dists) a.out_dists[(size_t)qi * k + i] = p.d;
    }
    if (lane == 0) {
      if (a.out_counts) a.out_counts[qi] = k;
      HS_WALL(c);
      Counters c1 = c;
      SearchArgs a1 = a;
      a1.pass_id = 1;  // answered by the tie replay
      write_stats(a1, qi, c1);
      a.status[qi] = ST_DONE;
    }
    return 0;
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
    HS_WALL(c);
    write_stats(a, qi, c);
    a.status[qi] = ST_DONE;
  }
  return 0;
}

// ---- kernels: grid-stride over the queries selected by status ----------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) strict_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  // pass 0: one query per workgroup, every query.  Re-run passes: 64 statuses per read, then the (normally zero) flagged
  // queries of the block one after the other.  (One call site, so that the search body is inlined and the kernel
  // arguments stay in scalar registers.)
  const bool scan = a.pass_id != 0;
  for (uint32_t it = blockIdx.x;; it += gridDim.x) {
    const uint32_t base = scan ? it * 64 : it;
    if (base >= a.nq) break;
    unsigned long long m = 1ull;
    if (scan) {
      const uint32_t q = base + threadIdx.x;
      m = hs_ballot(q < a.nq && ((1u << a.status[q]) & a.select_mask));
    }
    while (m) {
      const uint32_t qi = base + (uint32_t)__ffsll((long long)m) - 1;
      m &= m - 1;
      if (ix.n == 0) {  // cur_element_count == 0 (hnswalg_slim.h:2031-2032)
        if (threadIdx.x == 0) { if (a.out_counts) a.out_counts[qi] = 0; a.status[qi] = ST_DONE; }
        continue;
      }
      search_one_strict<METRIC>(ix, a, qi, smem);
      wave_sync();
    }
  }
}
// Fast kernel.  rc 3 = a tie had to be resolved but the insertion log did not fit: left to the strict pass.
// Wavefronts per SIMD the register allocation aims at.  A wavefront alone on a CU is only 25 % faster per expansion than one
// of sixteen (tools/profile_phases.py), so residency pays -- but only while it costs neither spills nor LDS.  At 96 VGPRs
// (5 waves) the short-row shapes spill 12-40 B per lane: +6 % on one 65k-query call, but once a kernel needs scratch the
// batches of several streams no longer overlap cleanly and the pipelined rate DROPS 8 % (8.4 -> 7.8 M q/s, same box,
// profiles/r02_ordered_pass_experiments.log), so the shipped value is 4 (-DHS_SHORT_WAVES=5 for the experiment).  Long rows keep
// 30 loads in flight in 168 VGPRs (3 waves), and so does the any-dim kernel with a wide result set.
#ifndef HS_SHORT_WAVES
#define HS_SHORT_WAVES 4
#endif
__host__ __device__ constexpr int fast_waves(int d16, int s) {
  return (d16 > 16 || (d16 == -1 && s >= 4)) ? 3 : ((d16 == 4 || d16 == 6 || d16 == 8 || d16 == 16) && s <= 2) ? HS_SHORT_WAVES : 4;
}
template <int METRIC, int S, int D16, bool WB = false, bool BARE = true>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(fast_waves(D16, S)))) fast_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (uint32_t it = blockIdx.x; it < a.nq; it += gridDim.x) {
    const uint32_t qi = (a.phase == 2 && a.order) ? a.order[it] : it;
    if (a.pass_id != 0 && !((1u << a.status[qi]) & a.select_mask)) continue;  // pass 0 takes every query
    if (ix.n == 0) {
      if (threadIdx.x == 0) { if (a.out_counts) a.out_counts[qi] = 0; a.status[qi] = ST_DONE; }
      continue;
    }
    const int rc = search_one_fast<METRIC, S, D16, WB, BARE, BARE && !WB>(ix, a, qi, smem);
    if (rc == 3 && threadIdx.x == 0) a.status[qi] = ST_HAZARD;
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

#if HS_TU_HAS_L2
// ---- query order for a two-launch fast pass --------------------------------------------------------------------------
// One workgroup: min / max of the entry distances, a 4096-bin histogram over that range (farthest first), its prefix
// sums, then each query takes the next free position of its bin.  Order inside a bin is whatever the atomics give: the
// order only decides when a query STARTS, never what it returns.
// (256 threads: with several batches in flight this trivial kernel must not wait for sixteen free wave slots on one CU)
constexpr uint32_t kOrderBins = 4096, kOrderThreads = 256, kOrderPer = kOrderBins / kOrderThreads;
__global__ void __launch_bounds__(kOrderThreads) order_kernel(const uint4 *entry, uint32_t *order, uint32_t nq) {
  __shared__ uint32_t bins[kOrderBins];
  __shared__ float red_lo[kOrderThreads / 64], red_hi[kOrderThreads / 64];
  __shared__ uint32_t wave_tot[kOrderThreads / 64];
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  float lo = FLT_MAX, hi = -FLT_MAX;
  auto key_of = [&](uint32_t i) -> float { return __uint_as_float(entry[i].y); };
  // (every pass over the entries takes eight loads per thread in flight: one at a time, the three passes were 30 us of dependent
  //  global-memory latency for 10k queries)
  constexpr uint32_t U = 8;
  for (uint32_t b0 = t; b0 < nq; b0 += kOrderThreads * U) {
    float dv[U];
#pragma unroll
    for (uint32_t j = 0; j < U; j++) { const uint32_t i = b0 + j * kOrderThreads; dv[j] = i < nq ? key_of(i) : __builtin_nanf(""); }
#pragma unroll
    for (uint32_t j = 0; j < U; j++) { const float d = dv[j]; if (d == d && fabsf(d) <= FLT_MAX) { lo = fminf(lo, d); hi = fmaxf(hi, d); } }
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  if (lane == 0) { red_lo[wv] = lo; red_hi[wv] = hi; }
  for (uint32_t b = t; b < kOrderBins; b += kOrderThreads) bins[b] = 0;
  __syncthreads();
  lo = red_lo[0]; hi = red_hi[0];
  for (uint32_t w = 1; w < kOrderThreads / 64; w++) { lo = fminf(lo, red_lo[w]); hi = fmaxf(hi, red_hi[w]); }
  const float scale = hi > lo ? (float)(kOrderBins - 1) / (hi - lo) : 0.f;
  auto bin_of = [&](float d) -> uint32_t {
    if (!(d == d)) return 0;                       // NaN: with the farthest
    const float x = (hi - fminf(fmaxf(d, lo), hi)) * scale;
    return min((uint32_t)x, kOrderBins - 1);
  };
  for (uint32_t b0 = t; b0 < nq; b0 += kOrderThreads * U) {
    float dv[U];
#pragma unroll
    for (uint32_t j = 0; j < U; j++) { const uint32_t i = b0 + j * kOrderThreads; dv[j] = i < nq ? key_of(i) : 0.f; }
#pragma unroll
    for (uint32_t j = 0; j < U; j++) if (b0 + j * kOrderThreads < nq) atomicAdd(&bins[bin_of(dv[j])], 1u);
  }
  __syncthreads();
  // exclusive prefix sums of the bins: kOrderPer consecutive bins per thread, wave scan, then the wave totals
  uint32_t v[kOrderPer], sum = 0;
  for (uint32_t j = 0; j < kOrderPer; j++) { v[j] = bins[kOrderPer * t + j]; sum += v[j]; }
  uint32_t incl = sum;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off);
    if (lane >= (uint32_t)off) incl += up;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t w = 0; w < wv; w++) base += wave_tot[w];
  uint32_t run = base + incl - sum;
  __syncthreads();
  for (uint32_t j = 0; j < kOrderPer; j++) { bins[kOrderPer * t + j] = run; run += v[j]; }
  __syncthreads();
  for (uint32_t b0 = t; b0 < nq; b0 += kOrderThreads * U) {
    float dv[U];
#pragma unroll
    for (uint32_t j = 0; j < U; j++) { const uint32_t i = b0 + j * kOrderThreads; dv[j] = i < nq ? key_of(i) : 0.f; }
#pragma unroll
    for (uint32_t j = 0; j < U; j++) { const uint32_t i = b0 + j * kOrderThreads; if (i < nq) order[atomicAdd(&bins[bin_of(dv[j])], 1u)] = i; }
  }
}
hipError_t launch_order(const uint4 *entry, uint32_t *order, uint32_t nq, hipStream_t stream) {
  static_assert(kOrderBins == kOrderPer * kOrderThreads, "whole bins per thread in the prefix step");
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, entry, order, nq);
  return hipGetLastError();
}
#endif

hipError_t launch_strict_l2(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream);
hipError_t launch_strict_ip(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream);
hipError_t launch_fast_l2(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream);
hipError_t launch_fast_ip(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream);
#if HS_TU_HAS_L2
hipError_t launch_strict_l2(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  return launch(strict_kernel<METRIC_L2>, ix, a, lds, stream);
}
hipError_t launch_strict(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = strict_lds_bytes(ix.dim, a.ef, a.fb_cand ? 0u : a.cand_cap, a.hash_slots);
  return ix.metric == METRIC_L2 ? launch_strict_l2(ix, a, lds, stream) : launch_strict_ip(ix, a, lds, stream);
}
hipError_t launch_fast(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = fast_lds_bytes(ix.dim, a.ef, a.cand_cap, a.hash_slots);
  return ix.metric == METRIC_L2 ? launch_fast_l2(ix, a, lds, stream) : launch_fast_ip(ix, a, lds, stream);
}
#endif
#if HS_TU_HAS_IP
hipError_t launch_strict_ip(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  return launch(strict_kernel<METRIC_IP>, ix, a, lds, stream);
}
#endif

// delete marks / filters: the variant with the reference's !bare_bone branches (runtime-dim and d=128 only)
template <int METRIC, int D16>
static hipError_t launch_fast_del(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.ef <= 64) return launch(fast_kernel<METRIC, 1, D16, false, false>, ix, a, lds, stream);
  if (a.ef <= 128) return launch(fast_kernel<METRIC, 2, D16, false, false>, ix, a, lds, stream);
  if (a.ef <= 256) return launch(fast_kernel<METRIC, 4, D16, false, false>, ix, a, lds, stream);
  return launch(fast_kernel<METRIC, 8, D16, false, false>, ix, a, lds, stream);
}
template <int METRIC, int D16>
static hipError_t launch_fast_md(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.k == a.ef) {  // nothing is selected at the end: the variant that watches ties across the capacity boundary
    if (a.ef <= 64) return launch(fast_kernel<METRIC, 1, D16, true>, ix, a, lds, stream);
    return launch(fast_kernel<METRIC, 2, D16, true>, ix, a, lds, stream);
  }
  if (a.ef <= 64) return launch(fast_kernel<METRIC, 1, D16>, ix, a, lds, stream);
  if (a.ef <= 128) return launch(fast_kernel<METRIC, 2, D16>, ix, a, lds, stream);
  if (a.ef <= 256) return launch(fast_kernel<METRIC, 4, D16>, ix, a, lds, stream);
  return launch(fast_kernel<METRIC, 8, D16>, ix, a, lds, stream);
}
#if HS_TU_HAS_L2
hipError_t launch_fast_l2(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (ix.has_deleted) {
    if (ix.dim == 128) return launch_fast_del<METRIC_L2, 8>(ix, a, lds, stream);
    return (ix.dim & 15u) ? launch_fast_del<METRIC_L2, -1>(ix, a, lds, stream) : launch_fast_del<METRIC_L2, 0>(ix, a, lds, stream);
  }
  // compile-time dims for the common shapes (the runtime-dim kernel is 1.15-1.7x slower: measured at d=64 and on DEEP-10M, d=96)
  switch (ix.dim) {
    case 128: return launch_fast_md<METRIC_L2, 8>(ix, a, lds, stream);    // SIFT
    case 96: return launch_fast_md<METRIC_L2, 6>(ix, a, lds, stream);     // DEEP
    case 960: return launch_fast_md<METRIC_L2, 60>(ix, a, lds, stream);   // GIST
    case 768: return launch_fast_md<METRIC_L2, 48>(ix, a, lds, stream);
    case 256: return launch_fast_md<METRIC_L2, 16>(ix, a, lds, stream);
    case 64: return launch_fast_md<METRIC_L2, 4>(ix, a, lds, stream);
    case 512: return launch_fast_md<METRIC_L2, 32>(ix, a, lds, stream);
    case 1024: return launch_fast_md<METRIC_L2, 64>(ix, a, lds, stream);
    // GloVe-100-like: the 4-lane recipes with the dim compiled in (the 25 steps unroll without spills; at 200 / 300 the
    // unrolled loads spill 80 / 250 B per lane, so those stay on the runtime-dim kernel)
    case 100: return launch_fast_md<METRIC_L2, -25>(ix, a, lds, stream);
    default: return (ix.dim & 15u) ? launch_fast_md<METRIC_L2, -1>(ix, a, lds, stream) : launch_fast_md<METRIC_L2, 0>(ix, a, lds, stream);
  }
}
#endif
#if HS_TU_HAS_IP
hipError_t launch_fast_ip(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (ix.has_deleted)
    return (ix.dim & 15u) ? launch_fast_del<METRIC_IP, -1>(ix, a, lds, stream) : launch_fast_del<METRIC_IP, 0>(ix, a, lds, stream);
  switch (ix.dim) {   // text / image embeddings
    case 768: return launch_fast_md<METRIC_IP, 48>(ix, a, lds, stream);    // COHERE
    case 512: return launch_fast_md<METRIC_IP, 32>(ix, a, lds, stream);
    case 1024: return launch_fast_md<METRIC_IP, 64>(ix, a, lds, stream);
    case 1536: return launch_fast_md<METRIC_IP, 96>(ix, a, lds, stream);
    case 100: return launch_fast_md<METRIC_IP, -25>(ix, a, lds, stream);   // GloVe-100-angular-like, as for L2
    default: return (ix.dim & 15u) ? launch_fast_md<METRIC_IP, -1>(ix, a, lds, stream) : launch_fast_md<METRIC_IP, 0>(ix, a, lds, stream);
  }
}
#endif

}  // namespace hs
