// group_search.hip -- the hot path on gfx950, second generation: FOUR queries per wavefront.
//
// A wavefront is four rows of 16 lanes (the DPP row); each row ("group") runs one query's whole searchKnn -- upper-layer
// greedy descent, level-0 best-first beam, k-selection -- and the four queries share every instruction: one visited-set
// probe sequence, one heap sift, one accept iteration serve four queries, and a distance pass (4 lanes per row, 16 rows)
// always has its four rows per query, so the scalar bookkeeping that bounded the one-query-per-wave kernel
// (beam_search.hip: ~100 VALU + 66 SALU per distance evaluation) is amortised fourfold.  Groups run in lock step but
// not in phase: each group is a small state machine (INIT -> DESC -> BEAM -> finish), and a group that finishes pulls the
// next query from a device-wide queue (persistent grid), so neither the tail of a launch nor a long query strands lanes.
//
// Replaces (paths relative to /root/reference/third_party/hnswlib/), for the shapes group_supported() admits:
//   HierarchicalNSWSlim::searchKnn            hnswalg_slim.h:1907-2028, 2030-2131
//   HierarchicalNSWSlim::searchBaseLayerST    hnswalg_slim.h:321-457   (bare_bone)
//   HierarchicalNSW::searchKnn / searchBaseLayerST  hnswalg.h:1378-1440 / 326-479   (bare_bone)
//   L2SqrSIMD16ExtAVX512 / InnerProductSIMD16ExtAVX512   space_l2.h:25-54 / space_ip.h:146-199
//   VisitedList                                visited_list_pool.h:10-31
//
// Same exactness contract as beam_search.hip's fast kernel: distances bit-identical (dist_recipe.hpp), the traversal is
// the reference's step for step (candidate heap = libstdc++ push_heap/pop_heap mechanics on the raw array, so the expansion
// order among equal-distance candidates is the reference's; counters equal), the k-subset is the reference's, output sorted
// by distance.  The result set is kept as KEYS ONLY (16 x SP registers per query): the traversal needs nothing but its
// maximum; ids come back at the end from the insertion log, and a query whose k-subset could depend on the layout of the
// reference's result heap (equal distances across the k-th boundary) replays that log through the libstdc++ mechanics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

#include "dist_recipe.hpp"
#include "engine.hpp"
#include "heap_emul.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr uint32_t kNoneG = 0xFFFFFFFFu;
static constexpr uint32_t kGroupScratch = 32;   // nid / nub / nd entries per group (a tile holds at most 32 ids)

__host__ __device__ inline uint32_t g_align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// Per-group LDS block; the four blocks of a wavefront lie `stride` bytes apart, stride = 64 (mod 256) so that the four
// groups' accesses to the same field fall on different banks.
struct GroupLds { uint32_t off_hash, off_cand, off_nid, off_nub, off_nd, off_q, stride; };
__host__ __device__ inline GroupLds group_layout(uint32_t dim_in_lds, uint32_t cand_cap, uint32_t hash_slots) {
  GroupLds l;
  l.off_hash = 0;
  l.off_cand = hash_slots * 4;                       // hash_slots % 4 == 0
  l.off_nid = l.off_cand + (cand_cap + 2) * 8;       // cand_cap even
  l.off_nub = l.off_nid + kGroupScratch * 4;
  l.off_nd = l.off_nub + kGroupScratch * 4;
  l.off_q = l.off_nd + kGroupScratch * 4;
  uint32_t total = g_align_up(l.off_q + dim_in_lds * 4, 64);
  while ((total & 255u) != 64u) total += 64;
  l.stride = total;
  return l;
}

enum : uint32_t { G_IDLE = 0, G_INIT = 1, G_DESC = 2, G_BEAM = 3 };

// Diagnostic build only (make gprof): shader-clock laps per phase, summed over the wavefronts of a launch into
// counters[16..48) as 16 x u64 (printed by hs_search_check under HS_GPROF=1).  The product build compiles these to nothing.
#ifdef HS_GPROF
#define GP_DECL() unsigned long long _gp[16] = {0}; unsigned long long _gt = clock64()
#define GP_LAP(i) do { const unsigned long long _t1 = clock64(); _gp[i] += _t1 - _gt; _gt = _t1; } while (0)
#define GP_CNT(i, v) do { _gp[i] += (v); } while (0)
#define GP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GP_FLUSH(ctr) do { if (threadIdx.x == 0) for (int _i = 0; _i < 16; _i++) atomicAdd(reinterpret_cast<unsigned long long *>((ctr) + 16) + _i, _gp[_i]); } while (0)
#else
#define GP_DECL() do {} while (0)
#define GP_LAP(i) do {} while (0)
#define GP_CNT(i, v) do {} while (0)
#define GP_DRAIN() do {} while (0)
#define GP_FLUSH(ctr) do {} while (0)
#endif

// ---- row (16-lane group) primitives ----------------------------------------------------------------------------------
// the 16 ballot bits of this lane's group
__device__ __forceinline__ uint32_t gbits(unsigned long long b, int lane) { return (uint32_t)(b >> (lane & 48)) & 0xFFFFu; }
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// all-lanes reductions over the row: rotate by 1, 2, 4, 8 (row_ror:n = 0x120 + n)
__device__ __forceinline__ float row_max_f32(float v) {
  v = fmaxf(v, __uint_as_float(dpp_u32<0x121>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u32<0x122>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u32<0x124>(__float_as_uint(v))));
  v = fmaxf(v, __uint_as_float(dpp_u32<0x128>(__float_as_uint(v))));
  return v;
}
__device__ __forceinline__ float row_min_f32(float v) {
  v = fminf(v, __uint_as_float(dpp_u32<0x121>(__float_as_uint(v))));
  v = fminf(v, __uint_as_float(dpp_u32<0x122>(__float_as_uint(v))));
  v = fminf(v, __uint_as_float(dpp_u32<0x124>(__float_as_uint(v))));
  v = fminf(v, __uint_as_float(dpp_u32<0x128>(__float_as_uint(v))));
  return v;
}
// value of lane `src` (0..15, row-uniform) of the row, in every lane of the row
__device__ __forceinline__ uint32_t row_bcast(uint32_t v, uint32_t src, int lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 48) | src) << 2), (int)v);
}
// wave-uniform maximum of a row-uniform value
__device__ __forceinline__ uint32_t wave_max_rows(uint32_t v) {
  const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

// ---- visited set: 4-slot buckets of 32-bit ids, one ds_read_b128 per probe (as beam_search.hip, tier 1 only) ----------
__device__ __forceinline__ int g_bucket_scan(const uint4 &w, uint32_t id) {  // -2 found, -1 full, else first free slot
  const uint32_t hit = min(min(w.x ^ id, w.y ^ id), min(w.z ^ id, w.w ^ id));
  const int used = 4 + (((int)w.x >> 31) + ((int)w.y >> 31) + ((int)w.z >> 31) + ((int)w.w >> 31));
  return hit == 0 ? -2 : (used < 4 ? used : -1);
}
// Tier 1 only (the common case: no group of the wavefront has spilled).
__device__ __forceinline__ bool g_vis_insert(uint32_t *tab, uint32_t nb, uint32_t id) {
  uint32_t b = __umulhi(id * 2654435761u, nb);
  while (true) {
    const uint4 w = reinterpret_cast<const uint4 *>(tab)[b];
    const int e = g_bucket_scan(w, id);
    if (e == -2) return false;
    if (e >= 0) {
      const uint32_t old = atomicCAS(&tab[b * 4 + e], kNoneG, id);
      if (__builtin_expect(old == kNoneG, 1)) return true;
      if (old == id) return false;
      continue;  // another lane of this group took the slot: look at the bucket again
    }
    if (++b == nb) b = 0;
  }
}
// Two tiers (beam_search.hip `Visited`): a group whose tier 1 reached its fill limit freezes it (read-only from then on) and
// continues in its query's tier-2 table in global memory -- a long query degrades to L2-latency probes instead of being re-run.
__device__ __forceinline__ bool g_vis_insert2(uint32_t *tab, uint32_t nb, bool frozen, uint32_t *tab2, uint32_t nb2, uint32_t id) {
  const uint32_t h = id * 2654435761u;
  uint32_t b = __umulhi(h, nb);
  while (true) {
    const uint4 w = reinterpret_cast<const uint4 *>(tab)[b];
    const int e = g_bucket_scan(w, id);
    if (e == -2) return false;
    if (e >= 0) {
      if (frozen) break;   // the id is not in tier 1
      const uint32_t old = atomicCAS(&tab[b * 4 + e], kNoneG, id);
      if (old == kNoneG) return true;
      if (old == id) return false;
      continue;
    }
    if (++b == nb) b = 0;
  }
  b = __umulhi(h, nb2);
  while (true) {
    const uint4 w = reinterpret_cast<const uint4 *>(tab2)[b];
    const int e = g_bucket_scan(w, id);
    if (e == -2) return false;
    if (e >= 0) {
      const uint32_t old = atomicCAS(&tab2[b * 4 + e], kNoneG, id);
      if (old == kNoneG) return true;
      if (old == id) return false;
      continue;
    }
    if (++b == nb2) b = 0;
  }
}

// ---- candidate min-heap (libstdc++ mechanics, element i at slot i+1: the children of a node share one 16-byte read) ----
// std::push_heap by the whole group in one read and one write round: lane t reads ancestor t+1 of the new slot, a ballot
// finds where the rise stops, the passed ancestors move one level down (beam_search.hip cand_push_t, per row).
// Slots [0, L) live in LDS (L even, so a child pair never straddles), the rest in the query's region of global memory
// (T2 = true; T2 = false: the heap is known to fit its LDS share -> LDS-only code, no vmcnt waits).
struct GHeap { uint2 *lds, *glob; uint32_t L; };
template <bool T2> __device__ __forceinline__ uint2 gh_get(const GHeap &h, uint32_t s) {
  if (!T2) return h.lds[s];
  return s < h.L ? h.lds[s] : h.glob[s - h.L];
}
template <bool T2> __device__ __forceinline__ void gh_set(const GHeap &h, uint32_t s, uint2 v) {
  if (!T2 || s < h.L) h.lds[s] = v;
  else h.glob[s - h.L] = v;
}
template <bool T2> __device__ __forceinline__ uint4 gh_get2(const GHeap &h, uint32_t s /*even*/) {
  if (!T2) return *reinterpret_cast<const uint4 *>(&h.lds[s]);
  return s < h.L ? *reinterpret_cast<const uint4 *>(&h.lds[s]) : *reinterpret_cast<const uint4 *>(&h.glob[s - h.L]);
}
template <bool T2>
__device__ __forceinline__ void g_cand_push(const GHeap &heap, uint32_t n /*size incl. new*/, float d, uint32_t id, bool act, int lane) {
  const int l = lane & 15;
  const uint32_t anc = n >> (l + 1);
  const bool has = act && anc != 0;
  uint2 p = make_uint2(0, 0);
  if (has) p = gh_get<T2>(heap, anc);
  const uint32_t rises = gbits(hs_ballot(has && __uint_as_float(p.x) > d), lane);
  const uint32_t r = __ffs(~rises) - 1;   // consecutive ancestors passed (< 16: the heap holds fewer than 65536 entries)
  if (act) {
    if ((uint32_t)l < r) gh_set<T2>(heap, n >> l, p);
    if ((uint32_t)l == r) gh_set<T2>(heap, n >> r, make_uint2(__float_as_uint(d), id));
  }
}
// std::pop_heap by one lane (beam_search.hip cand_pop_t); the root was read by the caller beforehand
template <bool T2>
__device__ __forceinline__ void g_cand_pop(const GHeap &heap, uint32_t n /*size before the pop*/) {
  if (n <= 1) return;
  const uint2 v = gh_get<T2>(heap, n);
  const uint32_t len = n - 1;
  uint32_t hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    const uint4 two = gh_get2<T2>(heap, child);   // a[child-1], a[child]
    const bool left = __uint_as_float(two.z) > __uint_as_float(two.x);
    gh_set<T2>(heap, hole + 1, left ? make_uint2(two.x, two.y) : make_uint2(two.z, two.w));
    child = left ? child - 1 : child;
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    gh_set<T2>(heap, hole + 1, gh_get<T2>(heap, child));
    hole = child - 1;
  }
  const float vd = __uint_as_float(v.x);
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const uint2 p = gh_get<T2>(heap, parent + 1);
    if (!(__uint_as_float(p.x) > vd)) break;
    gh_set<T2>(heap, hole + 1, p);
    hole = parent;
  }
  gh_set<T2>(heap, hole + 1, v);
}

// ---- distances: per group, rows nid[0..cnt) -> nd[0..cnt); 4 lanes per row, 4 rows per group and pass, two passes of
//      loads in flight.  QREG: the lane's query chunks live in registers (D16 <= 8), else in the group's LDS block.
template <int METRIC, int D16>
struct QRegs { float4 v[D16 > 0 ? D16 : 1]; };

template <int METRIC, int D16>
__device__ __forceinline__ void g_row_dist(const float *vec, uint32_t dim, const QRegs<METRIC, D16> &q, const float *qlds,
                                           uint32_t id, bool act, int sub, float (&acc)[4]) {
  acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
  if (D16 > 0) {
    float4 buf[D16 > 0 ? D16 : 1];
#pragma unroll
    for (int i = 0; i < D16; i++) buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
      const float4 *row = reinterpret_cast<const float4 *>(vec + (size_t)id * (D16 * 16)) + sub;
#pragma unroll
      for (int i = 0; i < D16; i++) buf[i] = row[i * 4];
    }
#pragma unroll
    for (int i = 0; i < D16; i++) step4<METRIC>(acc, q.v[i], buf[i]);
  } else {
    const uint32_t steps = dim >> 4;
    const float4 *row = reinterpret_cast<const float4 *>(vec + (size_t)id * dim) + sub;
    const float4 *qq = reinterpret_cast<const float4 *>(qlds) + sub;
    for (uint32_t r0 = 0; r0 < steps; r0 += 8) {
      const uint32_t nb = min(8u, steps - r0);
      float4 buf[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; i++) {
        buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act && i < nb) buf[i] = row[(r0 + i) * 4];
      }
#pragma unroll
      for (uint32_t i = 0; i < 8; i++)
        if (i < nb) step4<METRIC>(acc, qq[(r0 + i) * 4], buf[i]);
    }
  }
}

template <int METRIC, int D16>
__device__ __forceinline__ void g_dists(const float *vec, uint32_t dim, const QRegs<METRIC, D16> &q, const float *qlds,
                                        const uint32_t *nid, float *nd, uint32_t cnt /*row-uniform*/, uint32_t maxcnt /*wave-uniform*/,
                                        int lane) {
  const int sub = lane & 3, rs = (lane >> 2) & 3;
  if (D16 > 0) {
    for (uint32_t p = 0; p < maxcnt; p += 8) {
      const uint32_t j0 = p + rs, j1 = p + 4 + rs;
      const bool a0 = j0 < cnt, a1 = j1 < cnt;
      const bool two = p + 4 < maxcnt;   // wave-uniform
      const uint32_t id0 = a0 ? nid[j0] : 0u, id1 = a1 ? nid[j1] : 0u;
      float4 b0[D16 > 0 ? D16 : 1], b1[D16 > 0 ? D16 : 1];
#pragma unroll
      for (int i = 0; i < D16; i++) { b0[i] = make_float4(0.f, 0.f, 0.f, 0.f); b1[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
      if (a0) {
        const float4 *row = reinterpret_cast<const float4 *>(vec + (size_t)id0 * (D16 * 16)) + sub;
#pragma unroll
        for (int i = 0; i < D16; i++) b0[i] = row[i * 4];
      }
      if (two && a1) {
        const float4 *row = reinterpret_cast<const float4 *>(vec + (size_t)id1 * (D16 * 16)) + sub;
#pragma unroll
        for (int i = 0; i < D16; i++) b1[i] = row[i * 4];
      }
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < D16; i++) step4<METRIC>(acc, q.v[i], b0[i]);
      bool owner;
      float r = lane4_reduce<METRIC>(acc, sub, owner);
      if (a0 && owner) nd[j0] = r;
      if (two) {
        acc[0] = acc[1] = acc[2] = acc[3] = 0.f;
#pragma unroll
        for (int i = 0; i < D16; i++) step4<METRIC>(acc, q.v[i], b1[i]);
        r = lane4_reduce<METRIC>(acc, sub, owner);
        if (a1 && owner) nd[j1] = r;
      }
    }
  } else {
    for (uint32_t p = 0; p < maxcnt; p += 4) {
      const uint32_t j0 = p + rs;
      const bool a0 = j0 < cnt;
      const uint32_t id0 = a0 ? nid[j0] : 0u;
      float acc[4];
      g_row_dist<METRIC, D16>(vec, dim, q, qlds, id0, a0, sub, acc);
      bool owner;
      const float r = lane4_reduce<METRIC>(acc, sub, owner);
      if (a0 && owner) nd[j0] = r;
    }
  }
}

// =====================================================================================================================
// The kernel.  SP = result-set key slots per lane (ef <= 16 SP).  D16 = dim / 16 compiled in (query in registers), or 0.
// =====================================================================================================================
template <int METRIC, int D16, int SP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) group_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const float kInf = __builtin_inff();
  const uint32_t dim = D16 > 0 ? (uint32_t)D16 * 16u : ix.dim;
  const GroupLds L = group_layout(D16 > 0 ? 0u : ix.dim, a.cand_cap, a.hash_slots);
  unsigned char *gmem = smem + (size_t)g * L.stride;
  uint32_t *hash = reinterpret_cast<uint32_t *>(gmem + L.off_hash);
  GHeap cand;
  cand.lds = reinterpret_cast<uint2 *>(gmem + L.off_cand);
  cand.L = a.cand_cap + 2;   // cand_cap is even
  cand.glob = nullptr;
  const uint32_t cand_total = a.cand_cap + a.cand2_cap;
  uint32_t *nid = reinterpret_cast<uint32_t *>(gmem + L.off_nid);
  uint32_t *nub = reinterpret_cast<uint32_t *>(gmem + L.off_nub);
  float *nd = reinterpret_cast<float *>(gmem + L.off_nd);
  float *qlds = reinterpret_cast<float *>(gmem + L.off_q);
  const uint32_t ef = a.ef, k = a.k;
  const uint32_t nbuckets = a.hash_slots >> 2;
  const uint32_t vis_limit = a.hash_slots - (a.hash_slots >> 3);   // 87.5 % fill of 4-slot buckets
  const uint32_t nbuckets2 = a.spill_slots >> 2, vis_limit2 = a.spill_slots - (a.spill_slots >> 2);
  const uint32_t stride0 = ix.tile_stride, ustride = ix.up_stride;
  const bool watch = ef == k;   // nothing is selected at the end: ties ACROSS the capacity boundary decide the answer

  // ---- per-group state (row-uniform registers) -----------------------------------------------------------------------
  uint32_t st = G_IDLE, qi = 0;
  QRegs<METRIC, D16> q;
  float tk[SP];              // result-set keys, a per-lane column sorted descending; +inf = empty, -inf = beyond ef
  float lb_eff = kInf;       // max over the columns (empties included): accept iff d < lb_eff
  float rmax = 0.f;          // max real key while the set is not full (the reference's lowerBound then)
  uint32_t n_acc = 0;        // accepted so far == entries in the insertion log
  uint32_t cur = 0, cur_b = kNoneG;
  int lvl = 0;
  float curdist = 0.f;       // DESC: distance of cur; BEAM: distance of the node to expand next
  uint32_t cand_size = 0, pending = 0, n_vis = 0;
  uint32_t n_dist = 0, n_hops = 0, n_nbr = 0;
  bool btie = false;
  uint32_t t0a = kNoneG, t0b = kNoneG;   // level-0 tile ids of the node to expand (adjacency positions 2l, 2l+1 / l)
  uint2 ut0 = make_uint2(kNoneG, kNoneG), ut1 = make_uint2(kNoneG, kNoneG);   // upper-level tile entries l, 16 + l
  bool have_tile = false;    // DESC: a tile was requested for this round
  uint2 *tlog = nullptr;
  uint32_t *vis2 = nullptr;  // tier-2 visited set of the query (global memory)
  bool spilled = false;      // tier 1 frozen, inserts go to tier 2
  uint32_t n_vis2 = 0;
  bool queue_empty = false;
  GP_DECL();

  while (true) {
    // ================= refill: idle groups pull the next query ========================================================
    if (hs_ballot(st == G_IDLE) && !queue_empty) {
      uint32_t got = 0;
      if (st == G_IDLE && l == 0) got = atomicAdd(a.queue, 1u);
      got = row_bcast(got, 0, lane);
      const bool take = st == G_IDLE && got < a.nq;
      if (hs_ballot(st == G_IDLE && got >= a.nq)) queue_empty = true;
      if (take) {
        qi = got;
        const float *qsrc = a.queries + (size_t)qi * dim;
        if (D16 > 0) {
#pragma unroll
          for (int i = 0; i < D16; i++) q.v[i] = reinterpret_cast<const float4 *>(qsrc)[i * 4 + (lane & 3)];
        } else {
          for (uint32_t i = l; i < dim; i += 16) qlds[i] = qsrc[i];
        }
        for (uint32_t i = l; i < nbuckets; i += 16) reinterpret_cast<uint4 *>(hash)[i] = make_uint4(kNoneG, kNoneG, kNoneG, kNoneG);
#pragma unroll
        for (int s = 0; s < SP; s++) tk[s] = (uint32_t)(s * 16 + l) < ef ? kInf : -kInf;
        lb_eff = kInf;
        rmax = -kInf;
        n_acc = 0; cand_size = 0; pending = 0; n_vis = 0;
        n_dist = 0; n_hops = 0; n_nbr = 0;
        btie = false;
        cur = ix.enterpoint; cur_b = ix.ep_base; lvl = ix.maxlevel;
        vis2 = a.spill + (size_t)qi * a.spill_stride;
        cand.glob = reinterpret_cast<uint2 *>(vis2 + a.spill_slots);
        tlog = reinterpret_cast<uint2 *>(vis2 + a.spill_slots + 2 * a.cand2_cap);
        spilled = false;
        n_vis2 = 0;
        st = G_INIT;
      }
    }
    if (!hs_ballot(st != G_IDLE)) break;
    wave_sync();
    GP_LAP(0);
    GP_CNT(10, 1);

    // ================= BEAM groups: pending pushes (hnswalg_slim.h:408-411) and the pop (:353-354), LDS only, while
    //                   the tile of the node being expanded is in flight =================================================
    {
      const bool beam = st == G_BEAM;
      while (hs_ballot(beam && pending != 0)) {
        const bool act = beam && pending != 0;
        const uint32_t j = act ? (uint32_t)__ffs(pending) - 1u : 0u;
        float d = 0.f;
        uint32_t id = 0;
        if (act) { d = nd[j]; id = nid[j]; pending &= pending - 1; cand_size++; }
        if (__builtin_expect(hs_ballot(act && cand_size >= cand.L) == 0, 1)) g_cand_push<false>(cand, cand_size, d, id, act, lane);
        else g_cand_push<true>(cand, cand_size, d, id, act, lane);
        wave_sync();
        GP_CNT(11, 1);
      }
      GP_LAP(1);
      if (__builtin_expect(hs_ballot(beam && cand_size >= cand.L) == 0, 1)) {
        if (beam && l == 0) g_cand_pop<false>(cand, cand_size);
      } else {
        if (beam && l == 0) g_cand_pop<true>(cand, cand_size);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
      if (beam) { cand_size--; n_hops++; }
      wave_sync();
      GP_LAP(2);
    }
    GP_DRAIN();
    GP_LAP(3);

    // ================= ids to evaluate this round ======================================================================
    uint32_t cnt = 0;
    uint32_t fail = 0;   // 1 visited set full, 2 candidate heap full
    if (st == G_INIT) {
      if (l == 0) nid[0] = cur;
      cnt = 1;
      if (a.mark_ep && l == 0) g_vis_insert(hash, nbuckets, cur);   // visited_array[enterpoint] = tag (hnswalg_slim.h:1919)
      if (a.mark_ep) n_vis = 1;
    } else if (st == G_DESC) {
      n_hops++;
      if (have_tile) {
        const bool v0 = ut0.x != kNoneG, v1 = ut1.x != kNoneG;
        if (v0) { nid[l] = ut0.x; nub[l] = ut0.y; }
        if (v1) { nid[16 + l] = ut1.x; nub[16 + l] = ut1.y; }
        cnt = __popc(gbits(hs_ballot(v0), lane)) + __popc(gbits(hs_ballot(v1), lane));
        n_nbr += cnt;
      }
    } else if (st == G_BEAM) {
      // the node's whole level-0 list is one aligned tile (hnswalg_slim.h:363-369); test-and-mark every id (:392-393)
      const bool v0 = t0a != kNoneG, v1 = t0b != kNoneG;
      const uint32_t m = __popc(gbits(hs_ballot(v0), lane)) + __popc(gbits(hs_ballot(v1), lane));
      n_nbr += m;
      if (__builtin_expect(!spilled && n_vis + m > vis_limit, 0)) {
        // tier 1 is full: clear the query's tier-2 table and continue there
        for (uint32_t i = l; i < nbuckets2; i += 16) reinterpret_cast<uint4 *>(vis2)[i] = make_uint4(kNoneG, kNoneG, kNoneG, kNoneG);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        spilled = true;
        if (l == 0) atomicAdd(a.counters + 3, 1u);
      }
      fail = (spilled && n_vis2 + m > vis_limit2) ? 1u : ((cand_size + m > cand_total) ? 2u : 0u);
      if (!fail) {
        bool new0 = false, new1 = false;
        if (__builtin_expect(hs_ballot(spilled) == 0, 1)) {
          if (v0) new0 = g_vis_insert(hash, nbuckets, t0a);
          if (v1) new1 = g_vis_insert(hash, nbuckets, t0b);
        } else {
          if (v0) new0 = g_vis_insert2(hash, nbuckets, spilled, vis2, nbuckets2, t0a);
          if (v1) new1 = g_vis_insert2(hash, nbuckets, spilled, vis2, nbuckets2, t0b);
        }
        const uint32_t b0 = gbits(hs_ballot(new0), lane), b1 = gbits(hs_ballot(new1), lane);
        const uint32_t below = (1u << l) - 1u;
        const uint32_t pre = __popc(b0 & below) + __popc(b1 & below);   // unvisited ids before adjacency position 2l (resp. l)
        if (new0) nid[pre] = t0a;
        if (new1) nid[pre + (new0 ? 1u : 0u)] = t0b;
        cnt = __popc(b0) + __popc(b1);
        if (spilled) n_vis2 += cnt;
        else n_vis += cnt;
      }
    }
    if (__builtin_expect(hs_ballot(fail != 0) != 0, 0)) {
      if (fail) {   // scratch exhausted: the query is re-run by the one-query-per-wave kernels (tier 2 / whole-CU passes)
        if (l == 0) {
          a.status[qi] = ST_OVERFLOW;
          atomicAdd(a.counters + (fail - 1), 1u);
        }
        st = G_IDLE;
      }
    }
    wave_sync();
    GP_LAP(4);

    // ================= distances (hnswalg_slim.h:395-396, 2033-2035, 2064-2070) ========================================
    const uint32_t maxcnt = wave_max_rows(cnt);
    if (maxcnt) g_dists<METRIC, D16>(ix.vec, dim, q, qlds, nid, nd, cnt, maxcnt, lane);
    n_dist += cnt;
    wave_sync();
    GP_LAP(5);
    GP_CNT(13, (maxcnt + 3) / 4);

    // ================= INIT -> DESC ======================================================================================
    bool to_beam = false;   // this round's descent ended on level 0: seed the beam below
    if (st == G_INIT) {
      curdist = nd[0];
      st = G_DESC;
      have_tile = false;
      if (lvl <= ix.threshold_level) to_beam = true;
    } else if (st == G_DESC) {
      // first index attaining the minimum == what the sequential `if (d < curdist)` scan ends on (hnswalg_slim.h:2064-2075)
      bool moved = false;
      if (cnt) {
        const float d0 = (uint32_t)l < cnt ? nd[l] : kInf, d1 = (uint32_t)(16 + l) < cnt ? nd[16 + l] : kInf;
        const float mn = row_min_f32(fminf(d0, d1));
        if (mn < curdist) {
          const uint32_t e0 = gbits(hs_ballot(d0 == mn), lane), e1 = gbits(hs_ballot(d1 == mn), lane);
          const uint32_t j = e0 ? (uint32_t)__ffs(e0) - 1u : 16u + (uint32_t)__ffs(e1) - 1u;
          curdist = mn;
          cur = nid[j];
          cur_b = nub[j];
          moved = true;
        }
      }
      if (!moved) lvl--;
      if (lvl <= ix.threshold_level) to_beam = true;
    }
    wave_sync();
    // ---- seed the level-0 beam: top_candidates = candidate_set = {(curdist, cur)}, visited[cur] (hnswalg_slim.h:2100-2106;
    //      hnswalg.h:346-358 recomputes the entry distance) -- expressed as one pending acceptance of scratch entry 0
    uint32_t todo = 0;
    if (to_beam) {
      if (l == 0) { nd[0] = curdist; nid[0] = cur; g_vis_insert(hash, nbuckets, cur); }
      n_vis++;
      if (ix.kind == 0) n_dist++;
      st = G_BEAM;
      cand_size = 0;
      todo = 1u;
    } else if (st == G_BEAM && cnt) {
      // candidates of this tile that can pass `top_size < ef || lowerBound > d` (:403-404); lowerBound only falls
      const float d0 = (uint32_t)l < cnt ? nd[l] : kInf, d1 = (uint32_t)(16 + l) < cnt ? nd[16 + l] : kInf;
      todo = gbits(hs_ballot(d0 < lb_eff), lane) | (gbits(hs_ballot(d1 < lb_eff), lane) << 16);
    }
    wave_sync();

    // ================= accept loop (hnswalg_slim.h:403-452), adjacency order, one candidate per group and iteration ======
    float best_d = kInf;
    uint32_t best_id = 0;
    GP_LAP(6);
    while (hs_ballot(todo != 0)) {
      GP_CNT(12, 1);
      const bool act = todo != 0;
      const uint32_t j = act ? (uint32_t)__ffs(todo) - 1u : 0u;
      todo &= todo - 1;
      const float d = nd[j];
      const uint32_t id = nid[j];
      const bool ok = act && d < lb_eff;
      if (ok) {
        if (l == 0 && n_acc < a.log_cap) tlog[n_acc] = make_uint2(__float_as_uint(d), id);   // insertion log (:418-423)
        pending |= 1u << j;
        if (d < best_d) { best_d = d; best_id = id; }
        rmax = fmaxf(rmax, d);
      }
      // replace one instance of the set's maximum by d (push_heap + pop_heap of the reference, keys only): the first lane
      // whose column tops out at lb_eff drops that entry and bubbles d into its sorted column
      const uint32_t hb = gbits(hs_ballot(ok && tk[0] == lb_eff), lane);
      if (ok && (uint32_t)l == (uint32_t)__ffs(hb) - 1u) {
        float c = d;
#pragma unroll
        for (int s = 0; s + 1 < SP; s++) {
          const float nx = tk[s + 1];
          tk[s] = fmaxf(nx, c);
          c = fminf(nx, c);
        }
        tk[SP - 1] = c;
      }
      const float nlb = row_max_f32(tk[0]);
      if (ok) {
        btie = btie || (n_acc >= ef && nlb == lb_eff);   // an evicted key equals the last kept key
        n_acc++;
        lb_eff = nlb;
      }
    }

    GP_LAP(7);
    // ================= next node: root of candidate_set once the pending pushes are applied ==============================
    bool finish = false;
    if (st == G_BEAM) {
      float next_d = kInf;
      uint32_t next_id = 0;
      bool any = false;
      if (cand_size > 0) {
        const uint2 root = cand.lds[1];
        next_d = __uint_as_float(root.x);
        next_id = root.y;
        any = true;
      }
      if (pending != 0 && (!any || best_d < next_d)) { next_d = best_d; next_id = best_id; any = true; }
      const float lb_stop = n_acc >= ef ? lb_eff : rmax;
      finish = !any || next_d > lb_stop;   // candidate_set empty, or :340 `candidate distance > lowerBound`
      cur = next_id;
      curdist = next_d;
    }

    // ================= request the next round's tile =====================================================================
    t0a = t0b = kNoneG;
    ut0 = ut1 = make_uint2(kNoneG, kNoneG);
    have_tile = false;
    if (st == G_BEAM && !finish) {
      if (stride0 > 16) {
        const uint2 t = reinterpret_cast<const uint2 *>(ix.tile0 + (size_t)cur * 32)[l];
        t0a = t.x; t0b = t.y;
      } else {
        t0a = ix.tile0[(size_t)cur * 16 + l];
      }
    } else if (st == G_DESC && cur_b != kNoneG) {
      const uint2 *tp = ix.uptile + (size_t)(cur_b + (uint32_t)lvl - 1u) * ustride;
      ut0 = tp[l];
      if (ustride > 16) ut1 = tp[16 + l];
      have_tile = true;
    }

    GP_LAP(8);
    // ================= finish: k-selection and output =====================================================================
    if (__builtin_expect(hs_ballot(finish) != 0, 0)) {
      GP_CNT(14, 1);
      wave_sync();
      const uint32_t top_size = min(n_acc, ef);
      const uint32_t valid_n = min(top_size, k);
      bool hazard = false, replay = false;
      float kth = -kInf;     // the valid_n-th smallest key
      if (finish) {
        // columns: move the -inf "beyond ef" slots off the bottom (as +inf on top) so that every lane's minimum is its last slot
        for (int it = 0; it < SP; it++) {
          const bool sh = tk[SP - 1] == -kInf;
          if (!hs_ballot(sh)) break;
          if (sh) {
#pragma unroll
            for (int s = SP - 1; s > 0; s--) tk[s] = tk[s - 1];
            tk[0] = kInf;
          }
        }
        for (uint32_t i = 0; i < valid_n; i++) {
          const float mn = row_min_f32(tk[SP - 1]);
          const uint32_t hb = gbits(hs_ballot(tk[SP - 1] == mn), lane);
          if ((uint32_t)l == (uint32_t)__ffs(hb) - 1u) {
#pragma unroll
            for (int s = SP - 1; s > 0; s--) tk[s] = tk[s - 1];
            tk[0] = kInf;
          }
          kth = mn;
        }
        const float nxt = row_min_f32(tk[SP - 1]);
        replay = (top_size > k && nxt == kth) || (watch && btie);
        hazard = n_acc > a.log_cap;
      }
      // ids: scan the insertion log for the entries with key <= kth (exactly valid_n of them unless a tie crosses the boundary)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the log was written by this wave: stores before loads
      uint32_t nm = 0;
      if (finish && !hazard) {
        for (uint32_t base = 0; base < n_acc; base += 16) {
          const uint32_t idx = base + l;
          uint2 e = make_uint2(0, 0);
          if (idx < n_acc) e = tlog[idx];
          const bool mt = idx < n_acc && __uint_as_float(e.x) <= kth;
          const uint32_t mb = gbits(hs_ballot(mt), lane);
          const uint32_t at = nm + __popc(mb & ((1u << l) - 1u));
          if (mt && at < kGroupScratch) { nd[at] = __uint_as_float(e.x); nid[at] = e.y; }
          nm += __popc(mb);
        }
        if (nm != valid_n) replay = true;
      }
      wave_sync();
      if (finish && !hazard && !replay) {
        // rank inside the selection: ascending distance, log order among equal keys
        float md = kInf;
        uint32_t mid = 0;
        if ((uint32_t)l < valid_n) { md = nd[l]; mid = nid[l]; }
        uint32_t pos = 0;
        for (uint32_t j = 0; j < valid_n; j++) {
          const float dj = nd[j];
          pos += (dj < md || (dj == md && j < (uint32_t)l)) ? 1u : 0u;
        }
        if ((uint32_t)l < valid_n) {
          const uint64_t label = ix.labels[mid];
          if (a.out_labels32) a.out_labels32[(size_t)qi * k + pos] = (uint32_t)label;
          if (a.out_labels64) a.out_labels64[(size_t)qi * k + pos] = label;
          if (a.out_dists) a.out_dists[(size_t)qi * k + pos] = md;
        }
        if ((uint32_t)l >= valid_n && (uint32_t)l < k) {
          if (a.out_labels32) a.out_labels32[(size_t)qi * k + l] = 0xFFFFFFFFu;
          if (a.out_labels64) a.out_labels64[(size_t)qi * k + l] = ~0ull;
          if (a.out_dists) a.out_dists[(size_t)qi * k + l] = kInf;
        }
      }
      if (__builtin_expect(hs_ballot(finish && !hazard && replay) != 0, 0)) {
        // the reference's result heap rebuilt exactly: the logged insertions replayed through libstdc++'s push_heap /
        // pop_heap (hnswalg_slim.h:419-448) and the final nth_element (:2126) or pop_heap loop (:2019-2022), in the
        // (dead by now) visited-set area
        const bool rp = finish && !hazard && replay;
        Pair *top = reinterpret_cast<Pair *>(hash);
        uint32_t ts = 0;
        const uint32_t nlog = rp ? n_acc : 0u;
        for (uint32_t base = 0; hs_ballot(base < nlog); base += 16) {
          uint2 e = make_uint2(0, 0);
          if (base + l < nlog) e = tlog[base + l];
          wave_sync();
          if (base < nlog) { nd[l] = __uint_as_float(e.x); nid[l] = e.y; }
          wave_sync();
          if (base < nlog && l == 0) {
            const uint32_t m = min(16u, nlog - base);
            for (uint32_t j = 0; j < m; j++) {
              top[ts].d = nd[j];
              top[ts].id = nid[j];
              push_heap(top, (long)ts + 1, LessD());
              if (ts + 1 > ef) pop_heap(top, (long)ts + 1, LessD());
              ts = min(ts + 1, ef);
            }
          }
          wave_sync();
        }
        if (rp && l == 0) {
          if (a.mode == 0) {
            if (ts >= k) nth_element(top, (long)k, (long)ts, LessD());
          } else {
            uint32_t t2 = ts;
            while (t2 > k) { pop_heap(top, (long)t2, LessD()); t2--; }
          }
        }
        wave_sync();
        if (rp && (uint32_t)l < k) {
          const bool v = (uint32_t)l < valid_n;
          const Pair p = v ? top[l] : Pair{kInf, 0};
          const uint64_t label = v ? ix.labels[p.id] : ~0ull;
          if (a.out_labels32) a.out_labels32[(size_t)qi * k + l] = v ? (uint32_t)label : 0xFFFFFFFFu;
          if (a.out_labels64) a.out_labels64[(size_t)qi * k + l] = label;
          if (a.out_dists) a.out_dists[(size_t)qi * k + l] = p.d;
        }
        if (rp && l == 0) atomicAdd(a.counters + 2, 1u);
      }
      if (finish && l == 0) {
        if (hazard) {
          a.status[qi] = ST_HAZARD;   // log did not fit: the strict kernel re-runs the query
        } else {
          if (a.out_counts) a.out_counts[qi] = valid_n;
          if (a.stats) {
            a.stats[qi * 4 + 0] = n_dist;
            a.stats[qi * 4 + 1] = n_hops;
            a.stats[qi * 4 + 2] = n_nbr;
            a.stats[qi * 4 + 3] = replay ? 1u : 0u;
          }
          a.status[qi] = ST_DONE;
        }
      }
      if (finish) st = G_IDLE;
      wave_sync();
      GP_LAP(9);
    }
  }
  GP_FLUSH(a.counters);
}

// ---- host side ----------------------------------------------------------------------------------------------------

#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 0
bool group_supported(const DevIndex &ix, uint32_t ef, uint32_t k) {
  return ix.tile0 != nullptr && ix.tile_stride <= 32 && (ix.maxlevel == 0 || (ix.uptile != nullptr && ix.up_stride <= 32)) &&
         ix.threshold_level == 0 && !ix.has_deleted && ix.n > 0 && (ix.dim & 15u) == 0 && ef >= k && ef <= 256 && k <= 16;
}
size_t group_lds_bytes(uint32_t dim, uint32_t cand_cap, uint32_t hash_slots, bool q_in_regs) {
  return (size_t)group_layout(q_in_regs ? 0u : dim, cand_cap, hash_slots).stride * 4;
}
bool group_q_in_regs(int metric, uint32_t dim) { return metric == METRIC_L2 ? (dim == 128 || dim == 96 || dim == 64) : false; }
#endif

// Persistent grid: as many wavefronts as the device holds at once (each pulls queries until the queue is empty).  The
// residency of a (kernel, LDS size) pair is asked once per host thread and device.
template <typename K>
static hipError_t g_launch(K kern, const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  struct Memo { const void *fn; size_t lds; int dev; uint32_t resident; };
  static thread_local Memo memo[8] = {};
  static thread_local int memo_n = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  uint32_t resident = 0;
  for (int i = 0; i < memo_n; i++)
    if (memo[i].fn == reinterpret_cast<const void *>(kern) && memo[i].lds == lds && memo[i].dev == dev) resident = memo[i].resident;
  if (!resident) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int occ = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds);
    if (e != hipSuccess) return e;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    resident = (uint32_t)std::max(1, occ) * (uint32_t)std::max(1, cus);
    memo[memo_n % 8] = Memo{reinterpret_cast<const void *>(kern), lds, dev, resident};
    memo_n = std::min(memo_n + 1, 8);
  }
  const uint32_t grid = std::max(1u, std::min(resident, (a.nq + 3) / 4));
  static const bool dbg = getenv("HS_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[hs] group_kernel: nq=%u ef=%u lds=%zu B/wave (cand_cap=%u hash_slots=%u) resident=%u grid=%u\n", a.nq, a.ef, lds, a.cand_cap, a.hash_slots, resident, grid);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream, ix, a);
  return hipGetLastError();
}

template <int METRIC, int D16>
static hipError_t g_launch_sp(const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.ef <= 32) return g_launch(group_kernel<METRIC, D16, 2>, ix, a, lds, stream);
  if (a.ef <= 64) return g_launch(group_kernel<METRIC, D16, 4>, ix, a, lds, stream);
  if (a.ef <= 96) return g_launch(group_kernel<METRIC, D16, 6>, ix, a, lds, stream);
  if (a.ef <= 128) return g_launch(group_kernel<METRIC, D16, 8>, ix, a, lds, stream);
  return g_launch(group_kernel<METRIC, D16, 16>, ix, a, lds, stream);
}

hipError_t launch_group_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
hipError_t launch_group_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 0
hipError_t launch_group_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const bool qr = group_q_in_regs(METRIC_L2, ix.dim);
  const size_t lds = group_lds_bytes(ix.dim, a.cand_cap, a.hash_slots, qr);
  switch (qr ? ix.dim : 0u) {
    case 128: return g_launch_sp<METRIC_L2, 8>(ix, a, lds, stream);
    case 96: return g_launch_sp<METRIC_L2, 6>(ix, a, lds, stream);
    case 64: return g_launch_sp<METRIC_L2, 4>(ix, a, lds, stream);
    default: return g_launch_sp<METRIC_L2, 0>(ix, a, lds, stream);
  }
}
hipError_t launch_group(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  return ix.metric == METRIC_L2 ? launch_group_l2(ix, a, stream) : launch_group_ip(ix, a, stream);
}
#endif
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 1
hipError_t launch_group_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = group_lds_bytes(ix.dim, a.cand_cap, a.hash_slots, false);
  return g_launch_sp<METRIC_IP, 0>(ix, a, lds, stream);
}
#endif

}  // namespace hs
