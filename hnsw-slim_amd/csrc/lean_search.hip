// lean_search.hip -- the one-query-per-wavefront search kernel rebuilt for residency: same algorithm, same exactness contract
// and the same visited set / candidate heap as beam_search.hip's fast kernel, but sized so that FIVE wavefronts fit a SIMD
// (90-95 VGPRs as built, resource_usage.txt; ~6.6 KB of LDS per query) instead of four -- the round-2 measurements
// show the search is bound by how many independent gather streams the chip has in flight, not by arithmetic.  Since round 3 the
// flat kernel (flat_search.hip) is the default on the shapes this one serves; it stays for HS_KERNEL=lean A/B runs and its tests:
//   * distances: 8 lanes per row, 8 rows per pass; a lane owns TWO of the sixteen AVX-512 lane accumulators and loads 8 bytes
//     of every 64-byte step, so a pass keeps the same bytes in flight per row with half the buffer registers, and the L2 recipe
//     (rounded subtract, multiply, add) is exactly one v_pk_add / v_pk_mul / v_pk_add per step; values go straight to the lane
//     that owns the candidate (no LDS round trip);
//   * result set: KEYS ONLY, 64 x S registers as per-lane columns sorted descending (order-preserving integer image of the fp32
//     distance: integer min/max, no canonicalisation).  The traversal only ever needs the set's maximum: an accepted candidate
//     replaces one instance of it (the first lane whose column tops out at it bubbles the new key in) -- the reference's
//     push_heap + pop_heap as far as any decision can see.  Ids come back at the end from the insertion log, and a query
//     whose k-subset depends on the layout of the reference's heap (equal keys across the k-th boundary) replays the log
//     through the libstdc++ mechanics, as the fast kernel does;
//   * no nd[] / staging arrays in LDS: the new distances and ids of a hop stay in two registers until they are pushed.
// Serves bare (no delete marks / filter) indexes with level-0 and upper-level tiles, threshold_level 0, dim % 16 == 0,
// k <= 64, ef <= 256; everything else runs the fast / strict kernels.
//
// Replaces (paths relative to /root/reference/third_party/hnswlib/):
//   HierarchicalNSWSlim::searchKnn 2030-2131 / 1907-2028, searchBaseLayerST<bare_bone> hnswalg_slim.h:321-457,
//   HierarchicalNSW::searchKnn / searchBaseLayerST hnswalg.h:1378-1440 / 326-479, space_l2.h:25-54, space_ip.h:146-199.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>

#include "dist_recipe.hpp"
#include "engine.hpp"
#include "heap_emul.hpp"
#include "search_common.hpp"
#include "wave_util.hpp"

namespace hs {

struct LeanLds { uint32_t off_q, off_cand, off_hash, off_nid, off_nd, total; };
__host__ __device__ inline LeanLds lean_layout(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) {
  LeanLds l;
  l.off_q = 0;
  l.off_cand = align_up(dim * 4, 16);
  l.off_hash = l.off_cand + align_up((cand_cap + 2) * 8, 16);
  const uint32_t hash_bytes = hash_slots * 4 > (ef + 1) * 8 ? hash_slots * 4 : (ef + 1) * 8;   // the tie replay rebuilds the heap there
  l.off_nid = l.off_hash + align_up(hash_bytes, 16);
  l.off_nd = l.off_nid + 64 * 4;
  l.total = l.off_nd + 64 * 4;
  return l;
}

// order-preserving image of an fp32 value in int32 (no NaNs here): comparisons and min / max become integer operations
__device__ __forceinline__ int fkey(float d) {
  const int b = (int)__float_as_uint(d);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}
static constexpr int kKeyInf = 0x7F800000;   // fkey(+inf)

__device__ __forceinline__ int wave_max_i32(int v) {
#define HS_SHR_MAXI(ctrl) v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, ctrl, 0xf, 0xf, false))
  HS_SHR_MAXI(0x111); HS_SHR_MAXI(0x112); HS_SHR_MAXI(0x114); HS_SHR_MAXI(0x118);
#undef HS_SHR_MAXI
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
  const int r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  return max(max(r0, r1), max(r2, r3));
}
__device__ __forceinline__ int wave_min_i32(int v) {
#define HS_SHR_MINI(ctrl) v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, ctrl, 0xf, 0xf, false))
  HS_SHR_MINI(0x111); HS_SHR_MINI(0x112); HS_SHR_MINI(0x114); HS_SHR_MINI(0x118);
#undef HS_SHR_MINI
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
  const int r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  return min(min(r0, r1), min(r2, r3));
}

// Distances query (LDS) -> rows nid[0..cnt) (LDS); lane j < cnt receives the distance of row j.  8 lanes per row: lane s of a
// group owns AVX-512 lane accumulators 2s, 2s+1 and loads the 8 bytes at 64 i + 8 s of every 64-byte step i, so every
// accumulator sums its elements in the reference's order (space_l2.h:36-47, space_ip.h:183-195) and the sixteen accumulators
// are combined in the reference's order too (left to right for L2 :49-51, the halves tree of _mm512_reduce_add_ps for IP :197).
// `between()` runs after the first pass's loads have been issued and before they are consumed (LDS-only work hides there).
template <int METRIC, int D16, class Hook>
__device__ __forceinline__ float wave_dists8(const float *vec, uint32_t dim, const float *qv, const uint32_t *nid, uint32_t cnt, int lane,
                                             Hook between) {
  const int s = lane & 7, grp = lane >> 3;
  const uint32_t steps = dim >> 4;
  const hs_f2 *qq = reinterpret_cast<const hs_f2 *>(qv) + s;
  float out = FLT_MAX;
  for (uint32_t base = 0; base < cnt; base += 8) {
    const uint32_t j = base + grp;
    const bool act = j < cnt;
    const uint32_t id = nid[act ? j : base];   // idle groups re-read the pass's first row (cache hit) and discard
    const hs_f2 *row = reinterpret_cast<const hs_f2 *>(vec + (size_t)id * dim) + s;
    hs_f2 acc = {0.f, 0.f};
    if (D16 > 0) {   // compile-time dim (D16 = dim / 16 <= 8): every load of the row in flight at once, fully unrolled
      constexpr int B = D16 > 0 ? D16 : 1;
      hs_f2 buf[B];
#pragma unroll
      for (int i = 0; i < B; i++) buf[i] = row[i * 8];
      if (base == 0) between();
#pragma unroll
      for (int i = 0; i < B; i++) {
        const hs_f2 q2 = qq[i * 8];
        if (METRIC == METRIC_L2) {
          const hs_f2 t = q2 - buf[i];
          const hs_f2 p = t * t;
          acc = acc + p;
        } else {
          acc = __builtin_elementwise_fma(q2, buf[i], acc);
        }
      }
    } else
    for (uint32_t r0 = 0; r0 < steps; r0 += 8) {
      const uint32_t nb = min(8u, steps - r0);
      hs_f2 buf[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; i++)
        if (i < nb) buf[i] = row[(r0 + i) * 8];
      if (base == 0 && r0 == 0) between();
#pragma unroll
      for (uint32_t i = 0; i < 8; i++)
        if (i < nb) {
          const hs_f2 q2 = qq[(r0 + i) * 8];
          if (METRIC == METRIC_L2) {
            const hs_f2 t = q2 - buf[i];
            const hs_f2 p = t * t;
            acc = acc + p;
          } else {
            acc = __builtin_elementwise_fma(q2, buf[i], acc);
          }
        }
    }
    float r;
    int owner;
    if (METRIC == METRIC_L2) {
      r = acc.x + acc.y;
#pragma unroll
      for (int k = 1; k < 8; k++) {
        const float p = dpp_f<0x111>(r);   // row_shr:1 -- the running sum of the lane to the left
        if (s == k) r = (p + acc.x) + acc.y;
      }
      owner = 7;
    } else {
      float hx = acc.x + dpp_f<0x104>(acc.x), hy = acc.y + dpp_f<0x104>(acc.y);   // row_shl:4: accumulators j + 8
      hx = hx + dpp_f<0x102>(hx); hy = hy + dpp_f<0x102>(hy);                     // j + 4
      hx = hx + dpp_f<0x101>(hx); hy = hy + dpp_f<0x101>(hy);                     // j + 2
      r = 1.0f - (hx + hy);
      owner = 0;
    }
    const float got = __shfl(r, ((lane - (int)base) & 7) * 8 + owner, 64);
    if ((uint32_t)lane >= base && (uint32_t)lane < base + 8 && (uint32_t)lane < cnt) out = got;
  }
  return out;
}
struct NoHook8 { __device__ __forceinline__ void operator()() const {} };

template <int METRIC, int S, int D16>
__device__ int search_one_lean(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;
  const uint32_t dim = D16 > 0 ? (uint32_t)D16 * 16u : ix.dim;
  const LeanLds L = lean_layout(dim, a.ef, a.cand_cap, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  CandHeap cand;
  cand.lds = reinterpret_cast<uint2 *>(smem + L.off_cand);
  cand.L = a.cand_cap + 2;
  cand.glob = reinterpret_cast<uint2 *>(a.spill + (size_t)qi * a.spill_stride + a.spill_slots);
  const uint32_t cand_total = a.cand_cap + a.cand2_cap;
  uint2 *tlog = reinterpret_cast<uint2 *>(a.spill + (size_t)qi * a.spill_stride + a.spill_slots + 2 * a.cand2_cap);
  uint32_t *hash = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  uint32_t *nid = reinterpret_cast<uint32_t *>(smem + L.off_nid);
  float *nd = reinterpret_cast<float *>(smem + L.off_nd);
  const uint32_t k = a.k, ef = a.ef;
  uint32_t n_dist = 1, n_hops = 0, n_nbr = 0;

  // ---- stage the query, clear the visited set, entry distance, upper-layer greedy descent (hnswalg_slim.h:2033-2078) ----
  for (uint32_t i = lane; i < dim; i += 64) qv[i] = a.queries[(size_t)qi * dim + i];
  Visited vis;
  vis_init(vis, a, qi, hash, lane);
  uint32_t cur = ix.enterpoint;
  float curdist = 0.f;
  if (a.phase == 2) {
    // the descent ran in an earlier launch (SearchArgs::phase): its result and its counters come from entry[]
    const uint4 e = a.entry[qi];
    cur = uni(e.x);
    curdist = unif(__uint_as_float(e.y));
    n_dist = uni(e.z);
    n_hops = uni(e.w);
    n_nbr = n_dist - 1;
  } else {
  if (lane == 0) nid[0] = cur;
  wave_sync();
  curdist = unif(wave_dists8<METRIC, D16>(ix.vec, dim, qv, nid, 1, lane, NoHook8()));
  uint32_t cur_b = ix.ep_base;
  for (int lvl = ix.maxlevel; lvl > 0; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      n_hops++;
      if (cur_b == kNone) continue;
      uint2 pr = make_uint2(kNone, kNone);
      if ((uint32_t)lane < ix.up_stride) pr = ix.uptile[(size_t)(cur_b + lvl - 1) * ix.up_stride + lane];
      const uint32_t m = __popcll(hs_ballot(pr.x != kNone));   // ids are a prefix of the tile
      if (m == 0) continue;
      wave_sync();
      if ((uint32_t)lane < m) nid[lane] = pr.x;
      wave_sync();
      const float mine = wave_dists8<METRIC, D16>(ix.vec, dim, qv, nid, m, lane, NoHook8());
      n_nbr += m;
      n_dist += m;
      const float d = wave_min_f32(mine);
      const uint32_t l = (uint32_t)__ffsll((long long)hs_ballot((uint32_t)lane < m && mine == d)) - 1;
      if (l < m && d < curdist) {   // first index attaining the minimum == where the sequential scan ends (:2071-2075)
        curdist = d;
        cur = __builtin_amdgcn_readlane(pr.x, l);
        cur_b = __builtin_amdgcn_readlane(pr.y, l);
        changed = true;
      }
    }
  }
  if (a.phase == 1) {
    if (lane == 0) a.entry[qi] = make_uint4(cur, __float_as_uint(curdist), n_dist, n_hops);
    return 0;
  }
  }
  if (ix.kind == 0) n_dist++;   // searchBaseLayerST recomputes the entry distance (hnswalg.h:347-351)

  // ---- level-0 beam (hnswalg_slim.h:321-457, bare_bone) -----------------------------------------------------------------
  int tk[S];
#pragma unroll
  for (int s = 0; s < S; s++) tk[s] = (uint32_t)(s * 64 + lane) < ef ? kKeyInf : INT_MIN;
  int lb_key = kKeyInf;        // max over the columns, empties included: accept iff key < lb_key
  int rmax_key = INT_MIN;      // max real key while the set is not full (the reference's lowerBound then)
  uint32_t n_acc = 0;          // accepted so far == entries of the insertion log
  bool btie = false;
  const bool watch = ef == k;
  uint32_t cand_size = 0;
  unsigned long long pending = 1ull;   // accepted entries (bits index my_d / my_id of the previous hop) still to be pushed
  float my_d = curdist;                // lane j: distance / id of new neighbour j of the hop before
  uint32_t my_id = cur;
  float next_d = curdist;
  uint32_t next_id = cur;
  wave_sync();
  // visited_array[enterpoint] = tag of the (q,k) overloads (:1919; nothing reads the set before level 0, so it is marked here)
  // and visited_array[currObj] = tag (:2100-2102), in ONE place: lane 0 marks the entry, lane 1 the enter point.  (Three
  // inlined copies of the 16-bit set's code make this compiler emit an illegal V_CMP against src_shared_base for the kernel.)
  bool entry_marked = true;
  {
    const bool mine = lane == 0 || (lane == 1 && a.mark_ep);
    const uint32_t mid = lane == 0 ? cur : ix.enterpoint;
    if (vis.qbits) {
      bool fail = false;
      vis_test_and_mark_q16(vis, mid, mine, a, lane, fail);
      entry_marked = !fail;
    } else {
      if (mine) vis_insert(vis, mid);
      vis.n1 += a.mark_ep ? 2u : 1u;
    }
  }
  if (lane == 0) tlog[0] = make_uint2(__float_as_uint(curdist), cur);
  {   // top_candidates = {(curdist, cur)}
    const int kc = fkey(curdist);
    if (lane == 0) {
      int c = kc;
#pragma unroll
      for (int s = 0; s + 1 < S; s++) { const int nx = tk[s + 1]; tk[s] = max(nx, c); c = min(nx, c); }
      tk[S - 1] = c;
    }
    rmax_key = kc;
    n_acc = 1;
    lb_key = wave_max_i32(tk[0]);
  }
  const uint32_t stride = ix.tile_stride;
  int rc = entry_marked ? 0 : 1;
  while (rc == 0) {
    if (__builtin_expect(cand_size == 0 && pending == 0, 0)) break;
    if (__builtin_expect(fkey(next_d) > (n_acc >= ef ? lb_key : rmax_key), 0)) break;   // :340 candidate distance > lowerBound
    uint32_t id = kNone;
    if ((uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];   // the node's whole level-0 list: one aligned tile
    n_hops++;
    // pending pushes of the previous expansion (:408-411), in adjacency order, under the tile read
    while (pending) {
      const int j = __ffsll((long long)pending) - 1;
      pending &= pending - 1;
      cand_size++;
#ifdef HS_LEAN_NO_T2
      cand_push_t<false>(cand, cand_size, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j)), __builtin_amdgcn_readlane(my_id, j), lane);
#else
      cand_push(cand, cand_size, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j)), __builtin_amdgcn_readlane(my_id, j), lane);
#endif
      wave_sync();
    }
    const bool valid = id != kNone;
    const uint32_t m = __popcll(hs_ballot(valid));
    if (__builtin_expect(cand_size + m > cand_total, 0)) { rc = 2; break; }
    bool isnew = false;
    if (vis.qbits) {
      bool fail = false;
      isnew = vis_test_and_mark_q16(vis, id, valid, a, lane, fail);   // :392-393
      if (__builtin_expect(fail, 0)) { rc = 1; break; }
    } else {
      if (__builtin_expect(!vis_reserve(vis, m, a, lane), 0)) { rc = 1; break; }
      if (valid) isnew = vis_insert(vis, id);   // :392-393
    }
    const unsigned long long nm = hs_ballot(isnew);
    const uint32_t cnt = __popcll(nm);
    n_nbr += m;
    wave_sync();
    if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;   // unvisited ids, adjacency order
    wave_sync();
    if (!vis.qbits) vis_commit(vis, cnt);
    n_dist += cnt;
    // row loads go out first; pop_heap (:353-354) re-heapifies the LDS array while they are in flight
    auto pop_hook = [&]() {
#ifdef HS_LEAN_NO_T2
      if (lane == 0) cand_pop_t<false>(cand, cand_size);
#else
      if (lane == 0) cand_pop(cand, cand_size);
#endif
    };
    if (__builtin_expect(cnt > 0, 1)) {
      my_d = wave_dists8<METRIC, D16>(ix.vec, dim, qv, nid, cnt, lane, pop_hook);   // :395-396
    } else {
      pop_hook();
    }
    cand_size--;
    wave_sync();
    my_id = (uint32_t)lane < cnt ? nid[lane] : 0;
    // ---- accept (:403-452), adjacency order.  lowerBound only falls, so a candidate that fails it now never passes later.
    const int my_key = (uint32_t)lane < cnt ? fkey(my_d) : kKeyInf;
    unsigned long long todo = hs_ballot(my_key < lb_key);
    float best_d = FLT_MAX;
    uint32_t best_id = 0;
    bool have_best = false;
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int kj = __builtin_amdgcn_readlane(my_key, j);
      if (kj < lb_key) {
        const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_d), j));
        const uint32_t idj = __builtin_amdgcn_readlane(my_id, j);
        pending |= 1ull << j;
        if (!have_best || dj < best_d) { best_d = dj; best_id = idj; have_best = true; }
        if (lane == 0 && n_acc < a.log_cap) tlog[n_acc] = make_uint2(__float_as_uint(dj), idj);   // insertion log (:418-423)
        rmax_key = max(rmax_key, kj);
        // replace one instance of the set's maximum by the new key (push_heap + pop_heap of the reference, keys only)
        const int holder = __ffsll((long long)hs_ballot(tk[0] == lb_key)) - 1;
        if (lane == holder) {
          int c = kj;
#pragma unroll
          for (int s = 0; s + 1 < S; s++) { const int nx = tk[s + 1]; tk[s] = max(nx, c); c = min(nx, c); }
          tk[S - 1] = c;
        }
        const int nlb = wave_max_i32(tk[0]);
        btie = btie || (n_acc >= ef && nlb == lb_key);   // an evicted key equals the last kept key
        lb_key = nlb;
        n_acc++;
      }
    }
    // root of candidate_set once the pending pushes are applied: a pushed entry only passes strictly larger parents
    if (cand_size > 0) {
      const uint2 root = cand.lds[1];
      next_d = unif(__uint_as_float(root.x));
      next_id = uni(root.y);
      if (have_best && best_d < next_d) { next_d = best_d; next_id = best_id; }
    } else if (have_best) {
      next_d = best_d;
      next_id = best_id;
    }
  }
  if (__builtin_expect(rc != 0, 0)) {
    flag_query(a, qi, ST_OVERFLOW, rc - 1, lane);
    return rc;
  }

  // ---- k-selection: keys from the columns, ids from the insertion log ------------------------------------------------------
  const uint32_t top_size = min(n_acc, ef), valid_n = min(top_size, k);
  if (__builtin_expect(n_acc > a.log_cap, 0)) return 3;   // log did not fit: the strict kernel re-runs the query
  // columns hold +inf (empty), the keys descending, INT_MIN (slots beyond ef): bring every lane's smallest real key to its last slot
  for (int it = 0; it < S; it++) {
    const bool sh = tk[S - 1] == INT_MIN;
    if (!hs_ballot(sh)) break;
    if (sh) {
#pragma unroll
      for (int s = S - 1; s > 0; s--) tk[s] = tk[s - 1];
      tk[0] = kKeyInf;
    }
  }
  int kth = INT_MIN;
  for (uint32_t i = 0; i < valid_n; i++) {
    const int mn = wave_min_i32(tk[S - 1]);
    const int holder = __ffsll((long long)hs_ballot(tk[S - 1] == mn)) - 1;
    if (lane == holder) {
#pragma unroll
      for (int s = S - 1; s > 0; s--) tk[s] = tk[s - 1];
      tk[0] = kKeyInf;
    }
    kth = mn;
  }
  bool replay = (top_size > k && wave_min_i32(tk[S - 1]) == kth) || (watch && btie);
  __threadfence_block();   // the log was written by this wave
  uint32_t nm = 0;
  for (uint32_t base = 0; base < n_acc; base += 64) {
    const uint32_t idx = base + lane;
    uint2 e = make_uint2(0, 0);
    if (idx < n_acc) e = tlog[idx];
    const bool mt = idx < n_acc && fkey(__uint_as_float(e.x)) <= kth;
    const unsigned long long mb = hs_ballot(mt);
    const uint32_t at = nm + __popcll(mb & ((1ull << lane) - 1ull));
    if (mt && at < 64) { nd[at] = __uint_as_float(e.x); nid[at] = e.y; }
    nm += __popcll(mb);
  }
  if (nm != valid_n) replay = true;   // equal keys across the boundary
  wave_sync();
  if (__builtin_expect(!replay, 1)) {
    float md = FLT_MAX;
    uint32_t mid = 0;
    if ((uint32_t)lane < valid_n) { md = nd[lane]; mid = nid[lane]; }
    uint32_t pos = 0;   // rank inside the selection: ascending distance, log order among equal keys
    for (uint32_t j = 0; j < valid_n; j++) {
      const float dj = nd[j];
      pos += (dj < md || (dj == md && j < (uint32_t)lane)) ? 1u : 0u;
    }
    if ((uint32_t)lane < valid_n) {
      const uint64_t label = ix.labels[mid];
      if (a.out_labels32) a.out_labels32[(size_t)qi * k + pos] = (uint32_t)label;
      if (a.out_labels64) a.out_labels64[(size_t)qi * k + pos] = label;
      if (a.out_dists) a.out_dists[(size_t)qi * k + pos] = md;
    }
    if ((uint32_t)lane >= valid_n && (uint32_t)lane < k) {
      if (a.out_labels32) a.out_labels32[(size_t)qi * k + lane] = 0xFFFFFFFFu;
      if (a.out_labels64) a.out_labels64[(size_t)qi * k + lane] = ~0ull;
      if (a.out_dists) a.out_dists[(size_t)qi * k + lane] = __builtin_inff();
    }
  } else {
#ifdef HS_LEAN_NO_REPLAY
    return 3;
#endif
    // the reference's result heap rebuilt exactly: the logged insertions replayed through libstdc++'s push_heap / pop_heap
    // (hnswalg_slim.h:419-448) and the final nth_element (:2126) or pop_heap loop (:2019-2022), in the (dead by now) visited-set area
    if (lane == 0) atomicAdd(a.counters + 2, 1u);
    Pair *top = reinterpret_cast<Pair *>(hash);
    uint32_t ts = 0;
    for (uint32_t base = 0; base < n_acc; base += 64) {
      const uint32_t mm = min(64u, n_acc - base);
      uint2 e = make_uint2(0, 0);
      if ((uint32_t)lane < mm) e = tlog[base + lane];
      for (uint32_t j = 0; j < mm; j++) {
        const float d = __uint_as_float(__builtin_amdgcn_readlane(e.x, j));
        const uint32_t nb = __builtin_amdgcn_readlane(e.y, j);
        if (lane == 0) {
          top[ts].d = d;
          top[ts].id = nb;
          push_heap(top, (long)ts + 1, LessD());
          if (ts + 1 > ef) pop_heap(top, (long)ts + 1, LessD());
        }
        ts = min(ts + 1, ef);
      }
    }
    if (lane == 0) {
      if (a.mode == 0) {
        if (ts >= k) nth_element(top, (long)k, (long)ts, LessD());
      } else {
        uint32_t t2 = ts;
        while (t2 > k) { pop_heap(top, (long)t2, LessD()); t2--; }
      }
    }
    wave_sync();
    for (uint32_t i = lane; i < k; i += 64) {
      const bool v = i < valid_n;
      const Pair p = v ? top[i] : Pair{__builtin_inff(), 0};
      const uint64_t label = v ? ix.labels[p.id] : ~0ull;
      if (a.out_labels32) a.out_labels32[(size_t)qi * k + i] = v ? (uint32_t)label : 0xFFFFFFFFu;
      if (a.out_labels64) a.out_labels64[(size_t)qi * k + i] = label;
      if (a.out_dists) a.out_dists[(size_t)qi * k + i] = p.d;
    }
  }
  if (lane == 0) {
    if (a.out_counts) a.out_counts[qi] = valid_n;
    if (a.stats) {
      a.stats[qi * 4 + 0] = n_dist;
      a.stats[qi * 4 + 1] = n_hops;
      a.stats[qi * 4 + 2] = n_nbr;
      a.stats[qi * 4 + 3] = replay ? 1u : a.pass_id;
    }
    a.status[qi] = ST_DONE;
  }
  return 0;
}

#ifndef HS_LEAN_WAVES
#define HS_LEAN_WAVES 4
#endif
template <int METRIC, int S, int D16>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HS_LEAN_WAVES))) lean_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  for (uint32_t it = blockIdx.x; it < a.nq; it += gridDim.x) {
    const uint32_t qi = (a.phase == 2 && a.order) ? a.order[it] : it;
    if (a.pass_id != 0 && !((1u << a.status[qi]) & a.select_mask)) continue;   // pass 0 takes every query
    const int rc = search_one_lean<METRIC, S, D16>(ix, a, qi, smem);
    if (rc == 3 && threadIdx.x == 0) a.status[qi] = ST_HAZARD;
    wave_sync();
  }
}

template <typename K>
static hipError_t lean_launch(K kern, const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;
  if (lds > 64 * 1024) {   // (a large user cand_cap, or the candidate share after hs_search_check doubled it)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(std::max(1u, std::min(a.grid, a.nq))), dim3(64), lds, stream, ix, a);
  return hipGetLastError();
}
template <int METRIC, int D16>
static hipError_t lean_launch_d(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = lean_layout(ix.dim, a.ef, a.cand_cap, a.hash_slots).total;
  if (a.ef <= 64) return lean_launch(lean_kernel<METRIC, 1, D16>, ix, a, lds, stream);
  if (a.ef <= 128) return lean_launch(lean_kernel<METRIC, 2, D16>, ix, a, lds, stream);
  return lean_launch(lean_kernel<METRIC, 4, D16>, ix, a, lds, stream);
}
template <int METRIC>
static hipError_t lean_launch_s(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  if (METRIC == METRIC_L2 && ix.dim == 128) return lean_launch_d<METRIC, 8>(ix, a, stream);
  if (METRIC == METRIC_L2 && ix.dim == 96) return lean_launch_d<METRIC, 6>(ix, a, stream);
  return lean_launch_d<METRIC, 0>(ix, a, stream);
}

hipError_t launch_lean_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
hipError_t launch_lean_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 0
bool lean_supported(const DevIndex &ix, uint32_t ef, uint32_t k) {
  return ix.tile0 != nullptr && (ix.maxlevel == 0 || ix.uptile != nullptr) && ix.threshold_level == 0 && !ix.has_deleted && ix.n > 0 &&
         (ix.dim & 15u) == 0 && ef >= k && ef <= 256 && k <= 64;
}
// Where it is the better kernel: measured only on the shapes it is compiled for with a compile-time dim (L2, d = 96 / 128); every
// other shape would take its runtime-dim distance loop against the fast kernel's compiled-in one.
size_t lean_lds_bytes(uint32_t dim, uint32_t ef, uint32_t cand_cap, uint32_t hash_slots) { return lean_layout(dim, ef, cand_cap, hash_slots).total; }
hipError_t launch_lean_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) { return lean_launch_s<METRIC_L2>(ix, a, stream); }
hipError_t launch_lean(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  return ix.metric == METRIC_L2 ? launch_lean_l2(ix, a, stream) : launch_lean_ip(ix, a, stream);
}
#endif
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 1
hipError_t launch_lean_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) { return lean_launch_s<METRIC_IP>(ix, a, stream); }
#endif

}  // namespace hs
