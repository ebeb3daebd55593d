// flat_search.hip -- the "lazy heap" search kernel: one query per wavefront, the reference's level-0 beam
// (hnswalg_slim.h:321-457 bare_bone, hnswalg.h:326-479) WITHOUT maintaining candidate_set as long as the heap's layout cannot
// matter, and with the heap brought up to date from a log exactly when it can.
//
//   * Result set = top_candidates as a SORTED register array of (key, id | expanded-flag), rank r in lane r / S, slot r % S
//     (interleaved: inserting at rank p is one DPP wave_shr:1 plus per-slot selects, no carries between slots).
//   * With no delete marks every accepted neighbour enters the result set, so the entries of candidate_set that the reference
//     can still pop (key <= lowerBound) are exactly the unexpanded entries of the result set -- plus, while lowerBound does
//     not move, unexpanded entries that were evicted at a key EQUAL to it ("ghosts").  The reference pops the minimum of
//     candidate_set; when that minimum is attained by ONE entry it is the nearest unexpanded entry of the result set, whatever
//     the heap looks like: a ballot + ffs instead of push_heap x accepted + pop_heap.
//   * When the minimum is NOT unique (two unexpanded entries at the same distance, or a ghost at lowerBound), the heap's layout
//     decides, and the layout is history: every hop's accepted entries are logged (8 B each -- the log that also resolves a tie
//     at the k-th boundary at the end) with a per-hop count, and the heap is replayed, from where the last replay stopped,
//     through the libstdc++ push_heap / pop_heap mechanics (search_common.hpp) up to now; its root is popped; the search goes
//     on without the heap.  Replayed pops need no record: a past hop's pop was forced (unique minimum) or came from the heap
//     itself.  So the traversal is the reference's on every input, ties included, and a query that never ties never touches
//     the heap.
//   * Visited set: 16-byte buckets of 7 x 15-bit remainders of a bijective hash + a 16-bit fill count, any number of buckets
//     (bucket = h mod nb, remainder = h div nb).  The ids of one expansion are distinct, so a lookup is ONE ds_read_b128 +
//     packed 16-bit compares, and an insert is ONE ds_add_rtn (slot allocation) + ONE ds_write_b16: no compare-and-swap loop.
//     A full bucket sends its later arrivals to a short overflow list in LDS that the whole wave scans.
//   * Distances: 8 lanes per row, 8 rows per pass (lean_search.hip's mapping); the left-to-right sum of the sixteen AVX-512 lane
//     accumulators is seven (v_add_f32 with a DPP row_shr:1 operand, v_add_f32) pairs; the result stays in the group's last lane.
//
// Serves bare indexes (no delete marks / filter) with level-0 tiles, threshold_level 0, dim % 16 == 0, ef <= 512, k <= 64.
// Replaces (paths relative to /root/reference/third_party/hnswlib/): HierarchicalNSWSlim::searchKnn 2030-2131 / 1907-2028,
// searchBaseLayerST<bare_bone> hnswalg_slim.h:321-457, HierarchicalNSW::searchKnn / searchBaseLayerST hnswalg.h:1378-1440 /
// 326-479, space_l2.h:25-54, space_ip.h:146-199, visited_list_pool.h:10-31.
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

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef uint32_t hs_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) hs_u4 lds_u128;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) hs_f2 lds_f2;
typedef unsigned short hs_us2 __attribute__((ext_vector_type(2)));

static constexpr uint32_t kFlatOvfCap = 64;      // ids of full buckets (scanned by the wave)
static constexpr uint32_t kFlatBucketTags = 7;

struct FlatLds { uint32_t off_q, off_vis, off_ovf, off_nid, total; };
__host__ __device__ inline FlatLds flat_layout(uint32_t dim, uint32_t ef, uint32_t nb) {
  FlatLds l;
  l.off_q = 0;
  l.off_vis = align_up(dim * 4, 16);
  const uint32_t vis_bytes = nb * 16 > (ef + 1) * 8 ? nb * 16 : (ef + 1) * 8;   // the k-th boundary replay rebuilds the result heap there
  l.off_ovf = l.off_vis + align_up(vis_bytes, 16);
  l.off_nid = l.off_ovf + kFlatOvfCap * 4;
  l.total = l.off_nid + 64 * 4;
  return l;
}

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(hs_us2, a), __builtin_bit_cast(hs_us2, b)));
}
// order-preserving image of an fp32 distance in int32 (no NaNs): L2 distances are >= +0, whose bit patterns already order
template <int METRIC>
__device__ __forceinline__ int dkey(float d) {
  const int b = (int)__float_as_uint(d);
  return METRIC == METRIC_L2 ? b : (b ^ ((b >> 31) & 0x7FFFFFFF));
}
template <int METRIC>
__device__ __forceinline__ float key_dist(int k) {
  return __uint_as_float((uint32_t)(METRIC == METRIC_L2 ? k : (k ^ ((k >> 31) & 0x7FFFFFFF))));
}
static constexpr int kFKeyInf = 0x7F800000;       // dkey(+inf)
static constexpr uint32_t kFDone = 0x80000000u;   // bit 31 of a result-set id: expanded (or padding)

__device__ __forceinline__ int wave_min_i32_f(int v) {
#define HS_SHR_MINI(ctrl) v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, ctrl, 0xf, 0xf, false))
  HS_SHR_MINI(0x111); HS_SHR_MINI(0x112); HS_SHR_MINI(0x114); HS_SHR_MINI(0x118);
#undef HS_SHR_MINI
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
  const int r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  return min(min(r0, r1), min(r2, r3));
}

// loads in flight per lane for a compile-time dim of D16 sixteen-float steps
__host__ __device__ constexpr int flat_row_buffer(int d16) {
#ifdef HS_FLAT_LONG_B   // A/B knob (make flatvar): loads in flight per round for rows beyond 512 B
  return d16 <= 0 ? 1 : d16 <= 8 ? d16 : (d16 < HS_FLAT_LONG_B ? d16 : HS_FLAT_LONG_B);
#else
  return d16 <= 0 ? 1 : d16 <= 8 ? d16 : d16 % 30 == 0 ? 30 : d16 % 32 == 0 ? 32 : d16 % 24 == 0 ? 24 : 16;
#endif
}
// Wavefronts per SIMD a shape is compiled for: 5 (96 VGPRs, 8 KiB of LDS each) for the common shapes; result sets beyond 256
// entries (S = 6, 8: twelve / sixteen more registers) and long rows (a 16..32-deep load buffer) get 128 .. 256 VGPRs.
__host__ __device__ constexpr int flat_waves(int s, int d16) {
#ifdef HS_FLAT_WAVES
  return HS_FLAT_WAVES;
#else
#ifdef HS_FLAT_LONG_WAVES
  return d16 > 8 ? HS_FLAT_LONG_WAVES : d16 < 0 ? 3 : s > 4 ? 4 : 5;
#else
  return d16 > 8 ? 2 : d16 < 0 ? 3 : s > 4 ? 4 : 5;
#endif
#endif
}

// One pass of distances: the 8 lanes of group g = lane >> 3 work on row `rowid`; the group's lane OWN (7 for L2, 0 for IP)
// returns the distance in the reference's summation order (space_l2.h:36-51, space_ip.h:183-197); other lanes: unspecified.
template <int METRIC, int D16>
__device__ __forceinline__ float flat_dist8(const float *vec, uint32_t dim, const lds_u8 *qv, uint32_t rowid, int s) {
  const hs_f2 *row = reinterpret_cast<const hs_f2 *>(vec + (size_t)rowid * dim) + s;
  const lds_f2 *qq = reinterpret_cast<const lds_f2 *>(qv) + s;
  hs_f2 acc = {0.f, 0.f};
  auto step = [&](const hs_f2 q2, const hs_f2 x) {
    if (METRIC == METRIC_L2) {
      const hs_f2 t = q2 - x;
      const hs_f2 p = t * t;
      acc = acc + p;
    } else {
      acc = __builtin_elementwise_fma(q2, x, acc);
    }
  };
  if (D16 > 0) {
    // compile-time dim: the whole row in flight up to d = 128; longer rows in rounds of up to 32 eight-byte loads per lane.  d = 960
    // (two rounds of 30) is compiled for 2 wavefronts per SIMD (flat_waves()): with 256 VGPRs the compiler issues both rounds'
    // loads before the first use, and -- what decides -- keeps the hop's state out of scratch memory: the 3-wave build (168
    // VGPRs, 270 B of scratch per lane, reloads between the loads) took 5.55 ms for a 1000-query launch at ef = 384, this one
    // 3.38 ms, a 1-wave build with spills to AGPRs 3.20 ms (profiles/r03_longrow_ab.log, tools/r03_longrow_ab.sh).
    constexpr int B = flat_row_buffer(D16);
    constexpr int R = (D16 > 0 ? D16 : 1) / B, T = (D16 > 0 ? D16 : 1) % B;
    hs_f2 buf[B];
#pragma unroll
    for (int i = 0; i < B; i++) buf[i] = row[i * 8];
#pragma unroll
    for (int i = 0; i < B; i++) {
      if (B > 8 && i % 8 == 0) __builtin_amdgcn_sched_barrier(0);   // (keeps the query's LDS reads from being hoisted sixty deep)
      step(qq[i * 8], buf[i]);
    }
#pragma unroll 1
    for (int r = 1; r < R; r++) {
#pragma unroll
      for (int i = 0; i < B; i++) buf[i] = row[(r * B + i) * 8];
#pragma unroll
      for (int i = 0; i < B; i++) step(qq[(r * B + i) * 8], buf[i]);
    }
    if (T > 0) {
#pragma unroll
      for (int i = 0; i < T; i++) buf[i] = row[(R * B + i) * 8];
#pragma unroll
      for (int i = 0; i < T; i++) step(qq[(R * B + i) * 8], buf[i]);
    }
  } else {
    // runtime dim: rounds of 8 (D16 == 0) or 16 (D16 == -1: rows beyond 1 KB) loads in flight
    constexpr uint32_t RB = D16 == 0 ? 8u : 16u;
    const uint32_t steps = dim >> 4;
    for (uint32_t r0 = 0; r0 < steps; r0 += RB) {
      const uint32_t nb = min(RB, steps - r0);
      hs_f2 buf[RB];
#pragma unroll
      for (uint32_t i = 0; i < RB; i++)
        if (i < nb) buf[i] = row[(r0 + i) * 8];
#pragma unroll
      for (uint32_t i = 0; i < RB; i++)
        if (i < nb) step(qq[(r0 + i) * 8], buf[i]);
    }
  }
  if (METRIC == METRIC_L2) {
    // ((a0 + a1) + a2) + ... + a15: lane s adds its two accumulators to the running sum of lane s - 1.  Every lane runs every
    // step; lane s holds the right prefix after step s and is read (by lane s + 1) in step s + 1, before it is overwritten.
    float r = acc.x + acc.y;
#pragma unroll
    for (int k = 1; k < 8; k++) {
      float t;
      asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(r), "v"(acc.x));
      r = t + acc.y;
    }
    return r;
  } else {
    float hx = acc.x + dpp_f<0x104>(acc.x), hy = acc.y + dpp_f<0x104>(acc.y);   // row_shl:4: accumulators j + 8
    hx = hx + dpp_f<0x102>(hx); hy = hy + dpp_f<0x102>(hy);                     // j + 4
    hx = hx + dpp_f<0x101>(hx); hy = hy + dpp_f<0x101>(hy);                     // j + 2
    return 1.0f - (hx + hy);
  }
}

// The candidate min-heap of search_common.hpp (element i at slot i + 1; same sift decisions as std::push_heap / pop_heap with
// compare_by_first_rev, hnswalg_slim.h:177-183) with the address spaces spelled out: slots [0, L) in LDS, the rest in the
// query's scratch region.  (A generic pointer made from an LDS pointer sends this compiler into an illegal V_CMP against
// src_shared_base -- see wave_util.hpp::lds_cas.)
struct FlatHeap {
  lds_u8 *lds;
  uint2 *glob;   // slot s >= L at glob[s]
  uint32_t L;
};
typedef uint32_t hs_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) hs_u2 lds_u64;
// T2 = false: every slot touched is known to lie in LDS -> straight-line LDS code (no address tests, no global access)
template <bool T2>
__device__ __forceinline__ uint2 fh_get(const FlatHeap &h, uint32_t s) {
  if (!T2 || s < h.L) { const hs_u2 v = *reinterpret_cast<lds_u64 *>(h.lds + s * 8); return make_uint2(v.x, v.y); }
  return h.glob[s];
}
template <bool T2>
__device__ __forceinline__ void fh_set(const FlatHeap &h, uint32_t s, uint2 v) {
  if (!T2 || s < h.L) *reinterpret_cast<lds_u64 *>(h.lds + s * 8) = hs_u2{v.x, v.y};
  else h.glob[s] = v;
}
template <bool T2>
__device__ __forceinline__ uint4 fh_get2(const FlatHeap &h, uint32_t s /*even*/) {
  if (!T2 || s < h.L) { const hs_u4 v = *reinterpret_cast<lds_u128 *>(h.lds + s * 8); return make_uint4(v.x, v.y, v.z, v.w); }
  return *reinterpret_cast<const uint4 *>(&h.glob[s]);
}
// std::push_heap: all ancestors of the new slot are known up front, so the whole wave does it in one read and one write round
template <bool T2>
__device__ __forceinline__ void fh_push(const FlatHeap &h, uint32_t n /*size incl. new*/, float d, uint32_t id, int lane) {
  const uint32_t anc = n >> (lane + 1);
  const bool has = anc != 0 && lane < 31;
  uint2 p = make_uint2(0, 0);
  if (has) p = fh_get<T2>(h, anc);
  const unsigned long long rises = hs_ballot(has && __uint_as_float(p.x) > d);
  const uint32_t r = __ffsll((long long)~rises) - 1;
  if ((uint32_t)lane < r) fh_set<T2>(h, n >> lane, p);
  if ((uint32_t)lane == r) fh_set<T2>(h, n >> r, make_uint2(__float_as_uint(d), id));
}
// std::pop_heap (one lane); the popped root was read by the caller beforehand
template <bool T2>
__device__ __forceinline__ void fh_pop(const FlatHeap &h, uint32_t n /*size before pop*/) {
  if (n <= 1) return;
  const uint2 v = fh_get<T2>(h, n);
  const uint32_t len = n - 1;
  uint32_t hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    const uint4 two = fh_get2<T2>(h, child);
    const bool left = __uint_as_float(two.z) > __uint_as_float(two.x);
    fh_set<T2>(h, hole + 1, left ? make_uint2(two.x, two.y) : make_uint2(two.z, two.w));
    child = left ? child - 1 : child;
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    fh_set<T2>(h, hole + 1, fh_get<T2>(h, child));
    hole = child - 1;
  }
  const float vd = __uint_as_float(v.x);
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const uint2 p = fh_get<T2>(h, parent + 1);
    if (!(__uint_as_float(p.x) > vd)) break;
    fh_set<T2>(h, hole + 1, p);
    hole = parent;
  }
  fh_set<T2>(h, hole + 1, v);
}

// std::pop_heap by the whole wave, for a heap that lies in LDS: __adjust_heap's walk to the bottom picks, at every node, one of
// its two children, so lanes 0..6 read the child pairs of the hole, of its children and of its grandchildren in ONE LDS round,
// a scalar walk over the seven pick bits resolves three levels, and the (up to three) moves go out as one masked write.  A heap
// of 200 entries is popped in three LDS round trips instead of eight dependent ones.  Same decisions as fh_pop / std::pop_heap.
__device__ __forceinline__ void fh_pop_wave(const FlatHeap &h, uint32_t n /*size before pop*/, int lane) {
  if (n <= 1) return;
  const uint2 v = fh_get<false>(h, n);   // a[n-1] (uniform address: broadcast)
  const uint32_t len = n - 1;
  uint32_t hole = 0;
  const uint32_t dl = lane < 1 ? 0u : (lane < 3 ? 1u : 2u);   // depth of this lane's node below the hole
  const uint32_t ol = (uint32_t)lane + 1u - (1u << dl);
  while (true) {
    const uint32_t node = ((hole + 1u) << dl) - 1u + ol;
    const bool two = lane < 7 && 2u * node + 2u < len;   // the node has both children
    uint4 pr = make_uint4(0, 0, 0, 0);
    if (two) pr = fh_get2<false>(h, 2u * node + 2u);   // a[2 node + 1], a[2 node + 2]
    const bool right = !(__uint_as_float(pr.z) > __uint_as_float(pr.x));   // comp(a[child], a[child-1]) false: take a[child]
    const unsigned long long vm = hs_ballot(two), rm = hs_ballot(two && right);
    uint32_t l = 0, path = 0, last = 0;
    bool open = true;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      const bool ok = open && ((vm >> l) & 1ull);
      if (ok) { path |= 1u << l; last = l; l = 2u * l + 1u + (uint32_t)((rm >> l) & 1ull); }
      open = ok;
    }
    if (path == 0) break;
    const uint32_t child = 2u * node + 1u + (right ? 1u : 0u);
    if (lane < 7 && ((path >> lane) & 1u)) fh_set<false>(h, node + 1u, right ? make_uint2(pr.z, pr.w) : make_uint2(pr.x, pr.y));
    hole = __builtin_amdgcn_readlane(child, last);
    if (!open) break;
  }
  if ((len & 1u) == 0 && hole == (len - 2u) / 2u) {   // a last node with a left child only
    const uint2 c = fh_get<false>(h, 2u * hole + 2u);
    if (lane == 0) fh_set<false>(h, hole + 1u, c);
    hole = 2u * hole + 1u;
  }
  wave_sync();
  fh_push<false>(h, hole + 1u, __uint_as_float(v.x), v.y, lane);   // __push_heap(first, hole, top = 0, value)
}

// The lazy candidate heap brought up to date and its root popped (see the file header): state = the reference's candidate_set
// after `hp` pops and the pushes of `hq` hops; it rests in the query's scratch region in global memory (slot i + 1 = element i),
// and for the replay the head of the visited-set area (up to 8 KiB) is parked in the scratch region and the heap's first slots
// work in LDS.
struct FlatSync {
  lds_u8 *vis;
  uint32_t *scratch;
  uint32_t heap_lds_slots, off_heap, off_log, off_hop, hop_cap, cand_total;
  uint32_t n_log, hop0, hl, lb_bits;
  uint32_t cand_size, hp, hq, h_idx;
};
struct FlatSyncOut { uint32_t cand_size, hp, hq, h_idx, rc, stop, next_id; };
#ifdef HS_FLAT_SYNC_CALL   // A/B knob (make flatvar): a real call costs the kernel a stack in scratch memory; measured slower
#define HS_FLAT_SYNC_ATTR __attribute__((noinline))
#else
#define HS_FLAT_SYNC_ATTR __forceinline__
#endif
template <int METRIC>
__device__ HS_FLAT_SYNC_ATTR FlatSyncOut flat_heap_sync(FlatSync f) {
  const int lane = threadIdx.x;
  // (arguments of a device function arrive in vector registers: everything that is the same in all lanes goes back to scalar ones,
  //  so that the loops below are scalar loops)
  lds_u8 *vis = (lds_u8 *)(uintptr_t)uni((uint32_t)(uintptr_t)f.vis);
  const uint64_t sbits = (uint64_t)(uintptr_t)f.scratch;
  uint32_t *scratch = reinterpret_cast<uint32_t *>((uintptr_t)(((uint64_t)uni((uint32_t)(sbits >> 32)) << 32) | uni((uint32_t)sbits)));
  uint2 *heap_store = reinterpret_cast<uint2 *>(scratch + uni(f.off_heap));
  const uint2 *tlog = reinterpret_cast<const uint2 *>(scratch + uni(f.off_log));
  const uint32_t *hoplog = scratch + uni(f.off_hop);
  hs_u4 *park = reinterpret_cast<hs_u4 *>(scratch + uni(f.off_hop) + uni(f.hop_cap));
  const uint32_t heap_lds_slots = uni(f.heap_lds_slots), n_log = uni(f.n_log), hop0 = uni(f.hop0), hl = f.hl, cand_total = uni(f.cand_total);
  const int lb = (int)uni(f.lb_bits);
  uint32_t cand_size = uni(f.cand_size), hp = uni(f.hp), hq = uni(f.hq), h_idx = uni(f.h_idx), rc = 0, next_id = 0;
  __threadfence_block();
  for (uint32_t i = lane; i * 2u < heap_lds_slots; i += 64) park[i] = *reinterpret_cast<lds_u128 *>(vis + i * 16);
  FlatHeap cand;
  cand.lds = vis;
  cand.L = heap_lds_slots;
  cand.glob = heap_store;
  wave_sync();
  for (uint32_t sl = lane; sl <= cand_size && sl < heap_lds_slots; sl += 64) {
    const uint2 e = heap_store[sl];
    *reinterpret_cast<lds_u64 *>(vis + sl * 8) = hs_u2{e.x, e.y};
  }
  wave_sync();
  uint32_t ebase = ~0u;
  uint2 eb = make_uint2(0, 0);
  auto log_at = [&](uint32_t i) -> uint2 {
    if ((i & ~63u) != ebase) {
      ebase = i & ~63u;
      eb = (ebase + lane < n_log) ? tlog[ebase + lane] : make_uint2(0, 0);
    }
    return make_uint2(__builtin_amdgcn_readlane(eb.x, i & 63u), __builtin_amdgcn_readlane(eb.y, i & 63u));
  };
  if (h_idx == 0) {   // candidate_set = {entry} (hnswalg_slim.h:327-332)
    const uint2 e = log_at(0);
    cand_size = 1;
    fh_push<false>(cand, cand_size, __uint_as_float(e.x), e.y, lane);
    wave_sync();
    h_idx = 1;
  }
  const uint32_t ring_base = hop0 & ~63u;
  uint32_t hbase = ~0u, hcnt = 0;   // per-hop counts, 64 at a time (the unflushed block is still in `hl`)
  while (hq < hop0) {   // per past hop: pop (:353-354), then that hop's pushes (:408-411)
    if (hp == hq) {
      if (cand_size < heap_lds_slots) fh_pop_wave(cand, cand_size, lane);
      else if (lane == 0) fh_pop<true>(cand, cand_size);
      cand_size--;
      hp++;
      wave_sync();
    }
    if ((hq & ~63u) != hbase) {
      hbase = hq & ~63u;
      hcnt = hbase == ring_base ? hl : hoplog[hbase + lane];
    }
    const uint32_t np = (uint32_t)__builtin_amdgcn_readlane(hcnt, hq & 63u);
    if (__builtin_expect(cand_size + np > cand_total, 0)) { rc = 2; break; }
    const bool in_lds = cand_size + np < heap_lds_slots;
    for (uint32_t q = 0; q < np; q++) {
      const uint2 e = log_at(h_idx);
      cand_size++;
      if (in_lds) fh_push<false>(cand, cand_size, __uint_as_float(e.x), e.y, lane);
      else fh_push<true>(cand, cand_size, __uint_as_float(e.x), e.y, lane);
      wave_sync();
      h_idx++;
    }
    hq++;
  }
  bool stop = rc != 0 || cand_size == 0;
  if (!stop) {
    const uint2 root = fh_get<false>(cand, 1);
    const uint32_t rkey_bits = uni(root.x);
    if (dkey<METRIC>(__uint_as_float(rkey_bits)) > lb) {   // :340
      stop = true;
    } else {
      next_id = uni(root.y);
      if (cand_size < heap_lds_slots) fh_pop_wave(cand, cand_size, lane);
      else if (lane == 0) fh_pop<true>(cand, cand_size);
      cand_size--;
      hp++;
      wave_sync();
    }
  }
  // the heap goes back to its store, the visited set back to LDS
  for (uint32_t sl = lane; sl <= cand_size && sl < heap_lds_slots; sl += 64) {
    const hs_u2 e = *reinterpret_cast<lds_u64 *>(vis + sl * 8);
    heap_store[sl] = make_uint2(e.x, e.y);
  }
  wave_sync();
  __threadfence_block();
  for (uint32_t i = lane; i * 2u < heap_lds_slots; i += 64) *reinterpret_cast<lds_u128 *>(vis + i * 16) = park[i];
  wave_sync();
  FlatSyncOut o;
  o.cand_size = cand_size; o.hp = hp; o.hq = hq; o.h_idx = h_idx; o.rc = rc; o.stop = stop ? 1u : 0u; o.next_id = next_id;
  return o;
}

template <int S>
__device__ __forceinline__ int rank_key(const int (&tk)[S], uint32_t r) {   // key at rank r (uniform)
  const uint32_t l = r / S, sl = r % S;
  int v = __builtin_amdgcn_readlane(tk[0], l);
#pragma unroll
  for (int s = 1; s < S; s++) {
    const int t = __builtin_amdgcn_readlane(tk[s], l);
    v = sl == (uint32_t)s ? t : v;
  }
  return v;
}
template <int S>
__device__ __forceinline__ uint32_t rank_id(const uint32_t (&ti)[S], uint32_t r) {
  const uint32_t l = r / S, sl = r % S;
  uint32_t v = __builtin_amdgcn_readlane(ti[0], l);
#pragma unroll
  for (int s = 1; s < S; s++) {
    const uint32_t t = __builtin_amdgcn_readlane(ti[s], l);
    v = sl == (uint32_t)s ? t : v;
  }
  return v;
}

template <int METRIC, int S, int D16>
__device__ int search_one_flat(const DevIndex &ix, const SearchArgs &a, const uint32_t qi, lds_u8 *smem) {
  const int lane = threadIdx.x;
  constexpr int OWN = METRIC == METRIC_L2 ? 7 : 0;
  const int s8 = lane & 7, grp = lane >> 3;
  const uint32_t dim = D16 > 0 ? (uint32_t)D16 * 16u : ix.dim;
  const FlatLds L = flat_layout(dim, a.ef, a.fl_nb);
  lds_u8 *qv = smem + L.off_q;
  lds_u8 *vis = smem + L.off_vis;
  lds_u32 *ovf = reinterpret_cast<lds_u32 *>(smem + L.off_ovf);
  lds_u32 *nid = reinterpret_cast<lds_u32 *>(smem + L.off_nid);
  const uint32_t k = a.k, ef = a.ef;
  uint32_t *scratch = a.spill + (size_t)qi * a.spill_stride;
  uint2 *tlog = reinterpret_cast<uint2 *>(scratch + a.spill_slots + 2 * a.cand2_cap);
  uint32_t *hoplog = scratch + a.spill_slots + 2 * a.cand2_cap + 2 * a.log_cap;   // one word per level-0 hop: entries accepted
  const uint32_t hop_cap = a.hop_cap / 4;
  uint32_t n_dist = 1, n_hops = 0, n_nbr = 0;
#ifdef HS_FLAT_DIAG   // diagnostic build (make flatdiag): per-query wall clock and replay counts behind the stats block
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
  uint32_t diag_syncs = 0, diag_replayed = 0, diag_sync_ticks = 0, diag_pre = 0;
  uint32_t diag_ph[6] = {0, 0, 0, 0, 0, 0};   // shader cycles: select | tile wait | visited + compaction | rows + distances | accept (+ pre-select) | hop end
  unsigned long long diag_tp = __builtin_amdgcn_s_memtime();
#define HS_DIAG_LAP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); diag_ph[i] += (uint32_t)(t_ - diag_tp); diag_tp = t_; }
#else
#define HS_DIAG_LAP(i)
#endif

  // ---- stage the query, clear the visited set ---------------------------------------------------------------------------
  for (uint32_t i = lane; i < dim; i += 64) reinterpret_cast<lds_f32 *>(qv)[i] = a.queries[(size_t)qi * dim + i];
  for (uint32_t i = lane; i < a.fl_nb; i += 64) *reinterpret_cast<lds_u128 *>(vis + i * 16) = hs_u4{~0u, ~0u, ~0u, 0x0000FFFFu};
  uint32_t ovf_n = 0;
  const uint32_t vmask = a.vis_bits >= 32 ? 0xFFFFFFFFu : (1u << a.vis_bits) - 1u;
  const uint32_t nb = a.fl_nb, vmul = a.fl_mul, vsh = a.fl_sh;
  wave_sync();

  // Test-and-mark of one id per lane (`valid` lanes; the ids of one call are distinct).  True in the lanes whose id was new.
  // A bucket that is full sends its later arrivals to the overflow list in LDS, and once that is full too, to the query's
  // tier-2 table in global memory (search_common.hpp: 4-slot buckets, linear probing, cleared on first use) -- only the few
  // longest queries of a batch ever get there.  `vfail` (uniform) is raised when even that is exhausted.
  bool vfail = false;
  Visited v2;
  v2.t2 = scratch;
  v2.slots2 = a.spill_slots;
  v2.limit2 = a.spill_slots - (a.spill_slots >> 2);
  v2.n2 = 0;
  v2.spilled = false;
  auto vis_mark = [&](const uint32_t id, const bool valid) -> bool {
    const uint32_t h32 = id * 2654435761u;
    const uint32_t h = h32 & vmask;
    const uint32_t tag = __umulhi(h, vmul) >> vsh;           // h div nb
    const uint32_t b = h - tag * nb;                         // h mod nb
    lds_u8 *bp = vis + b * 16;
    const hs_u4 w = *reinterpret_cast<lds_u128 *>(bp);
    const uint32_t t2 = tag | (tag << 16);
    const uint32_t m = pk_min_u16(pk_min_u16(w.x ^ t2, w.y ^ t2), pk_min_u16(w.z ^ t2, (w.w ^ tag) | 0xFFFF0000u));
    const bool hit = (m & 0xFFFFu) == 0 || (m >> 16) == 0;
    bool isnew = valid && !hit;
    bool done = false;   // test-and-mark completed in tier 2
    // a full bucket's later arrivals live in the overflow list / tier 2
    unsigned long long lm = hs_ballot(isnew && (w.w >> 16) >= kFlatBucketTags);
    if (__builtin_expect(lm != 0 && ovf_n != 0, 0)) {
      const uint32_t mine = (uint32_t)lane < ovf_n ? ovf[lane] : kNone;
      unsigned long long todo = lm;
      while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t idj = __builtin_amdgcn_readlane(id, j);
        const bool found = hs_ballot(mine == idj) != 0;
        if (found && lane == j) isnew = false;
      }
      if (v2.spilled) {
        const bool mine2 = isnew && ((lm >> lane) & 1ull);
        if (mine2) { isnew = vis_insert_t2(v2, id, h32); done = true; }
        v2.n2 += __popcll(hs_ballot(mine2 && isnew));
      }
    }
    uint32_t e = 0;
    if (isnew && !done) {
      const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<lds_u32 *>(bp + 12), 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      e = old >> 16;
      if (e < kFlatBucketTags) *reinterpret_cast<lds_u16 *>(bp + e * 2) = (unsigned short)tag;
    }
    const bool over = isnew && !done && e >= kFlatBucketTags;
    const unsigned long long om = hs_ballot(over);
    if (__builtin_expect(om != 0, 0)) {
      const uint32_t at = ovf_n + __popcll(om & ((1ull << lane) - 1ull));
      const bool to_list = over && at < kFlatOvfCap;
      if (to_list) ovf[at] = id;
      const uint32_t room = kFlatOvfCap - ovf_n;
      const uint32_t want = __popcll(om);
      ovf_n += min(want, room);
      if (want > room) {   // the list is full: the rest (ids that are new by now) go to tier 2
        if (!v2.spilled) {
          for (uint32_t i = lane; i < v2.slots2; i += 64) v2.t2[i] = kEmpty;
          __threadfence_block();
          v2.spilled = true;
          if (lane == 0) atomicAdd(a.counters + 3, 1u);
        }
        if (over && !to_list) vis_insert_t2(v2, id, h32);
        v2.n2 += want - room;
      }
      if (v2.n2 > v2.limit2) vfail = true;
      wave_sync();
    }
    return isnew;
  };

  // ---- level-0 entry: from the descent launch (phase 2) or the upper-layer greedy descent here (hnswalg_slim.h:2033-2078) ------
  uint32_t cur = ix.enterpoint;
  float curdist = 0.f;
  if (a.phase == 2) {
    const uint4 e = a.entry[qi];
    cur = uni(e.x);
    curdist = unif(__uint_as_float(e.y));
    n_dist = uni(e.z);
    n_hops = uni(e.w);
    n_nbr = n_dist - 1;
  } else {
    {
      const float d = flat_dist8<METRIC, D16>(ix.vec, dim, qv, cur, s8);
      curdist = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), OWN));
    }
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
        n_nbr += m;
        n_dist += m;
        // the sequential scan (:2067-2076) ends on the first index attaining the minimum, if that beats curdist
        int best_key = INT_MAX;
        uint32_t best_at = 0;
        for (uint32_t base = 0; base < m; base += 8) {
          const uint32_t j = base + grp;
          const uint32_t rid = __shfl(pr.x, (int)(j < m ? j : base), 64);
          const float d = flat_dist8<METRIC, D16>(ix.vec, dim, qv, rid, s8);
          const int ky = (s8 == OWN && j < m) ? dkey<METRIC>(d) : INT_MAX;
          const int mn = wave_min_i32_f(ky);
          if (mn < best_key) {
            best_key = mn;
            best_at = base + ((uint32_t)(__ffsll((long long)hs_ballot(ky == mn)) - 1) >> 3);
          }
        }
        if (best_key < dkey<METRIC>(curdist)) {
          curdist = key_dist<METRIC>(best_key);
          cur = __builtin_amdgcn_readlane(pr.x, best_at);
          cur_b = __builtin_amdgcn_readlane(pr.y, best_at);
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

  // ---- level-0 beam ----------------------------------------------------------------------------------------------------
  int tk[S];
  uint32_t ti[S];
  unsigned long long inr[S];   // lanes whose slot-s rank is below ef
#pragma unroll
  for (int s = 0; s < S; s++) {
    tk[s] = kFKeyInf;
    ti[s] = kFDone;
    inr[s] = hs_ballot((uint32_t)(lane * S + s) < ef);
  }
  const int laneS = lane * S;
  const bool exact_fit = ef == 64u * S;   // no spare ranks behind ef - 1
  // visited_array[enterpoint] = tag of the (q,k) overloads (:1919; nothing reads the set before level 0) and
  // visited_array[currObj] = tag (:2100-2102): lane 0 marks the entry, lane 1 the enter point
  {
    const bool mine = lane == 0 || (lane == 1 && a.mark_ep && ix.enterpoint != cur);
    vis_mark(lane == 0 ? cur : ix.enterpoint, mine);
  }
  // top_candidates = {(curdist, cur)} (:2100-2101); candidate_set likewise (:327-332): log entry 0
  if (lane == 0) {
    tk[0] = dkey<METRIC>(curdist);
    ti[0] = cur;
    tlog[0] = make_uint2(__float_as_uint(curdist), cur);
  }
  int lb = rank_key<S>(tk, ef - 1);   // lowerBound as far as any decision sees it: key of rank ef-1 (+inf while the set is not full)
  uint32_t n_log = 1;                 // accepted so far == entries of the insertion log
  uint32_t hop0 = 0;                  // level-0 expansions so far
  uint32_t hl = 0;                    // ring: lane h & 63 = entries accepted by level-0 hop h (the unflushed block)
  int ghost_key = INT_MIN;            // an unexpanded entry was evicted at this key while it was the bound
  bool btie = false;                  // an evicted key equalled the last kept key (decides the answer when ef == k)
  // lazy candidate heap (search_common.hpp): state = the reference's candidate_set after `hp` pops and the pushes of `hq` hops.
  // Between two replays it rests in the query's scratch region in global memory (slot i + 1 = element i, as CandHeap has it);
  // for a replay the first 8 KiB of the visited-set area are parked in the scratch region and the heap's first 1024 slots work in LDS.
  uint32_t cand_size = 0, hp = 0, hq = 0, h_idx = 0;
  const uint32_t stride = ix.tile_stride;
  int rc = vfail ? 1 : 0;

  // nearest unexpanded entry of the result set (ranks < ef): {found, its lane and slot, key, id, how many unexpanded entries share the key}
  struct Near { bool any; int p, slot, key; uint32_t id, same; };
  auto nearest = [&]() -> Near {
    Near r;
    unsigned long long um[S], any = 0;
#pragma unroll
    for (int s = 0; s < S; s++) {
      um[s] = hs_ballot((int)ti[s] >= 0) & inr[s];
      any |= um[s];
    }
    r.any = any != 0;
    r.p = 0; r.slot = 0; r.key = kFKeyInf; r.id = 0; r.same = 0;
    if (any) {
      const int p = __ffsll((long long)any) - 1;
      int u_key = __builtin_amdgcn_readlane(tk[S - 1], p);
      uint32_t u_id = __builtin_amdgcn_readlane(ti[S - 1], p);
      int u_slot = S - 1;
#pragma unroll
      for (int s = S - 2; s >= 0; s--) {
        const bool here = (um[s] >> p) & 1ull;
        const int kk = __builtin_amdgcn_readlane(tk[s], p);
        const uint32_t ii = __builtin_amdgcn_readlane(ti[s], p);
        u_key = here ? kk : u_key;
        u_id = here ? ii : u_id;
        u_slot = here ? s : u_slot;
      }
      uint32_t same = 0;
#pragma unroll
      for (int s = 0; s < S; s++) same += __popcll(hs_ballot(tk[s] == u_key) & um[s]);
      r.p = p; r.slot = u_slot; r.key = u_key; r.id = u_id; r.same = same;
    }
    return r;
  };
  // The node of the NEXT expansion is usually known before this expansion's accept pass has run (see the pass loop): its tile
  // is then already on its way (pre_tile) while the accepted entries are inserted.
  bool have_pre = false;
  uint32_t pre_id = 0, pre_tile = kNone;
  int pre_key = 0;

  while (rc == 0) {
    // ---- the node to expand: minimum of candidate_set (:335-354) -------------------------------------------------------
    uint32_t next_id = 0, id = kNone;
    bool need_heap = false;
    if (have_pre && !(ghost_key == lb && pre_key == lb)) {   // (an entry evicted at the very key of the chosen node: the heap decides)
      next_id = pre_id;
      id = pre_tile;
    } else {
      if (have_pre) {   // take the choice back: the node is still an unexpanded entry of the set
#pragma unroll
        for (int s = 0; s < S; s++) ti[s] = ti[s] == (pre_id | kFDone) ? pre_id : ti[s];
      }
      const Near u = nearest();
      const bool any = u.any;
      need_heap = ghost_key == lb;
      if (any) {
        need_heap = u.same > 1 || (need_heap && u.key == lb);
        next_id = u.id;
      } else if (!need_heap) {
        break;   // nothing left that the reference could pop with dist <= lowerBound
      }
    if (__builtin_expect(need_heap, 0)) {
      // ---- bring the heap up to date and pop its root (a real call: the replay keeps its own registers, see flat_heap_sync) ----
#ifdef HS_FLAT_DIAG
      diag_syncs += any == 0 ? 0x10000u : (ghost_key == lb ? 0x100u : 1u);
      diag_replayed += hop0 - hq;
      const unsigned long long diag_s0 = __builtin_amdgcn_s_memrealtime();
#endif
      FlatSync fs;
      fs.vis = vis; fs.heap_lds_slots = min(a.fl_nb * 2u, 1024u) & ~1u; fs.scratch = scratch; fs.off_heap = a.spill_slots;
      fs.off_log = a.spill_slots + 2 * a.cand2_cap; fs.off_hop = fs.off_log + 2 * a.log_cap; fs.hop_cap = hop_cap;
      fs.cand_total = a.cand2_cap - 2; fs.n_log = n_log; fs.hop0 = hop0; fs.hl = hl; fs.lb_bits = (uint32_t)lb;
      fs.cand_size = cand_size; fs.hp = hp; fs.hq = hq; fs.h_idx = h_idx;
      const FlatSyncOut fo = flat_heap_sync<METRIC>(fs);
      cand_size = fo.cand_size; hp = fo.hp; hq = fo.hq; h_idx = fo.h_idx;
#ifdef HS_FLAT_DIAG
      diag_sync_ticks += (uint32_t)(__builtin_amdgcn_s_memrealtime() - diag_s0);
#endif
      if (fo.rc != 0) { rc = (int)fo.rc; break; }
      if (fo.stop) break;
      next_id = fo.next_id;
    }
      // flag the node expanded where it sits in the result set
#pragma unroll
      for (int s = 0; s < S; s++) ti[s] = ti[s] == next_id ? (next_id | kFDone) : ti[s];
      // ---- its level-0 list is one aligned tile (:358-369)
      if ((uint32_t)lane < stride) id = ix.tile0[(size_t)next_id * stride + lane];
    }
#ifdef HS_FLAT_DIAG
    diag_pre += have_pre ? 1u : 0u;
    HS_DIAG_LAP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HS_DIAG_LAP(1);
#endif
    have_pre = false;
    n_hops++;
    const bool valid = id != kNone;
    const uint32_t m = __popcll(hs_ballot(valid));
    const bool isnew = vis_mark(id, valid);   // :392-393
    if (__builtin_expect(vfail, 0)) { rc = 1; break; }
    const unsigned long long nm = hs_ballot(isnew);
    const uint32_t cnt = __popcll(nm);
    n_nbr += m;
    n_dist += cnt;
    if (isnew) nid[__popcll(nm & ((1ull << lane) - 1ull))] = id;   // unvisited ids, adjacency order
    wave_sync();
    HS_DIAG_LAP(2);
    uint32_t n_acc_hop = 0;
    for (uint32_t base = 0; base < cnt; base += 8) {
      const uint32_t j = base + grp;
      const bool act = j < cnt;
      const uint32_t rid = nid[act ? j : base];   // idle groups re-read the pass's first row (cache hit) and discard
      const float d = flat_dist8<METRIC, D16>(ix.vec, dim, qv, rid, s8);   // :395-396
      const int my_key = (s8 == OWN && act) ? dkey<METRIC>(d) : kFKeyInf;
#ifdef HS_FLAT_DIAG
      asm volatile("s_nop 0" : : "v"(my_key) : "memory");
      HS_DIAG_LAP(3);
#endif
      // ---- the next node, before the accept pass (last pass of the hop): candidate_set's minimum once this hop's pushes are in
      // is the nearest unexpanded entry u of the set as it stands, or the nearest new neighbour b if that one is accepted --
      // it is iff it passes the bound as it stands now: earlier neighbours of the pass are strictly farther and cannot take
      // the bound down to it -- and strictly nearer than u.  u must still be in the set afterwards: at most the new keys below
      // u's move it up.  Ties (b's key twice, b == u, u's key twice) and a ghost at the bound are left to the top of the loop.
      int b_lane = -1;
      if (base + 8 >= cnt && ghost_key != lb) {
        const Near u = nearest();
        int mk = min(my_key, __builtin_amdgcn_update_dpp(kFKeyInf, my_key, 0x118, 0xf, 0xf, false));   // row_shr:8: owner lanes 8 apart
        const int b_key = min(min(__builtin_amdgcn_readlane(mk, OWN + 8), __builtin_amdgcn_readlane(mk, OWN + 24)),
                              min(__builtin_amdgcn_readlane(mk, OWN + 40), __builtin_amdgcn_readlane(mk, OWN + 56)));
        const bool b_ok = b_key < lb;
        if (b_ok && (!u.any || b_key < u.key)) {
          const unsigned long long bm = hs_ballot(my_key == b_key);
          if (__popcll(bm) == 1) {
            b_lane = __ffsll((long long)bm) - 1;
            have_pre = true;
            pre_key = b_key;
            pre_id = __builtin_amdgcn_readlane(rid, b_lane);
          }
        } else if (u.any && u.same == 1 && (!b_ok || u.key < b_key)) {
          const uint32_t below = __popcll(hs_ballot(my_key < u.key));
          if ((uint32_t)(u.p * S + u.slot) + below < ef) {
            have_pre = true;
            pre_key = u.key;
            pre_id = u.id;
#pragma unroll
            for (int s = 0; s < S; s++) ti[s] = ti[s] == pre_id ? (pre_id | kFDone) : ti[s];
          }
        }
        pre_tile = kNone;
        if (have_pre && (uint32_t)lane < stride) pre_tile = ix.tile0[(size_t)pre_id * stride + lane];
      }
      // ---- accept (:403-452), adjacency order (= lane order of the owner lanes).  lowerBound only falls.
      unsigned long long todo = hs_ballot(my_key < lb);
      unsigned long long am = 0;
      while (todo) {
        const int jl = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int kj = __builtin_amdgcn_readlane(my_key, jl);
        if (kj < lb) {
          const uint32_t idj = __builtin_amdgcn_readlane(rid, jl) | (jl == b_lane ? kFDone : 0u);   // (the chosen one enters flagged)
          am |= 1ull << jl;
          uint32_t pos = 0;   // entries with key <= kj stay in front
#pragma unroll
          for (int s = 0; s < S; s++) pos += __popcll(hs_ballot(tk[s] <= kj));
          // without spare ranks behind ef - 1 the entry this insertion pushes out is gone at once: look at it now
          // (with spare ranks the evicted entries stay in the array and are looked at once per hop, below)
          const int ek = lb;
          uint32_t ei = 0;
          if (exact_fit) ei = rank_id<S>(ti, ef - 1);
          const int upk = __builtin_amdgcn_update_dpp(0, tk[S - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
          const uint32_t upi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ti[S - 1], 0x138, 0xf, 0xf, false);
#pragma unroll
          for (int s = S - 1; s >= 1; s--) {
            const bool gt = (uint32_t)(laneS + s) > pos;
            tk[s] = gt ? tk[s - 1] : tk[s];
            ti[s] = gt ? ti[s - 1] : ti[s];
          }
          {
            const bool gt = (uint32_t)laneS > pos;
            tk[0] = gt ? upk : tk[0];
            ti[0] = gt ? upi : ti[0];
          }
          {   // the new entry: one v_writelane per register, in the slot the rank falls in
            const uint32_t pl = pos / S, ps = pos % S;
#pragma unroll
            for (int s = 0; s < S; s++)
              if (ps == (uint32_t)s) {
                tk[s] = (int)write_lane((uint32_t)tk[s], (uint32_t)kj, pl);
                ti[s] = write_lane(ti[s], idj, pl);
              }
          }
          lb = rank_key<S>(tk, ef - 1);   // :450-452
          if (exact_fit && ek != kFKeyInf && lb == ek) {   // an entry left the set at exactly the new bound
            btie = true;
            if ((int)ei >= 0) ghost_key = ek;   // ... unexpanded: the reference may still pop it
          }
        }
      }
      // insertion log (:408-423: the pushes of candidate_set and top_candidates), adjacency order
      if (am) {
        const uint32_t at = n_log + __popcll(am & ((1ull << lane) - 1ull));
        if (((am >> lane) & 1ull) && at < a.log_cap) tlog[at] = make_uint2(__float_as_uint(d), rid);
        const uint32_t na = __popcll(am);
        n_log += na;
        n_acc_hop += na;
      }
      HS_DIAG_LAP(4);
    }
    if (!exact_fit && n_acc_hop != 0 && lb != kFKeyInf) {
      // entries evicted at exactly the bound sit right behind rank ef - 1 for as long as the bound does not move: one of them
      // decides the answer when ef == k (btie), an unexpanded one is still a candidate of the reference (ghost)
      unsigned long long eq = 0, un = 0;
#pragma unroll
      for (int s = 0; s < S; s++) {
        const unsigned long long e = hs_ballot(tk[s] == lb) & ~inr[s];
        eq |= e;
        un |= e & hs_ballot((int)ti[s] >= 0);
      }
      if (eq) btie = true;
      if (un) ghost_key = lb;
      // (an array whose very last entry still has the bound's key may have lost such an entry off its end)
      if (__builtin_expect(rank_key<S>(tk, 64 * S - 1) == lb, 0)) { rc = 3; break; }
    }
    hl = write_lane(hl, uni(n_acc_hop), uni(hop0 & 63u));
    hop0++;
    if (__builtin_expect((hop0 & 63u) == 0, 0)) {
      if (hop0 > hop_cap) { rc = 3; break; }
      hoplog[hop0 - 64 + lane] = hl;
    }
    if (__builtin_expect(n_log > a.log_cap, 0)) { rc = 3; break; }
    HS_DIAG_LAP(5);
  }
  if (__builtin_expect(rc == 1 || rc == 2, 0)) {
    flag_query(a, qi, ST_OVERFLOW, rc - 1, lane);
    return rc;
  }
  if (__builtin_expect(rc == 3, 0)) return 3;   // a log did not fit: the strict kernel re-runs the query

  // ---- k-selection (hnswalg_slim.h:2126-2130 / 2019-2027): the set is sorted, the k nearest are ranks 0..k-1 -- unless equal
  //      keys straddle the boundary: then the reference's choice depends on the layout of its result heap, which is rebuilt by
  //      replaying the insertion log through the libstdc++ mechanics (in the visited-set area, dead by now)
  const uint32_t top_size = min(n_log, ef), valid_n = min(top_size, k);
  const bool watch = ef == k;
  const bool replay = (top_size > k && rank_key<S>(tk, k - 1) == rank_key<S>(tk, k)) || (watch && btie);
  if (__builtin_expect(!replay, 1)) {
#pragma unroll
    for (int s = 0; s < S; s++) {
      const uint32_t r = laneS + s;
      if (r < k) {
        const bool v = r < valid_n;
        const uint64_t label = v ? ix.labels[ti[s] & ~kFDone] : ~0ull;
        if (a.out_labels32) a.out_labels32[(size_t)qi * k + r] = v ? (uint32_t)label : 0xFFFFFFFFu;
        if (a.out_labels64) a.out_labels64[(size_t)qi * k + r] = label;
        if (a.out_dists) a.out_dists[(size_t)qi * k + r] = v ? key_dist<METRIC>(tk[s]) : __builtin_inff();
      }
    }
  } else {
    if (lane == 0) atomicAdd(a.counters + 2, 1u);
    __threadfence_block();
    Pair *top = reinterpret_cast<Pair *>((unsigned char *)smem + L.off_vis);
    uint32_t ts = 0;
    for (uint32_t base = 0; base < n_log; base += 64) {
      const uint32_t mm = min(64u, n_log - base);
      uint2 e = make_uint2(0, 0);
      if ((uint32_t)lane < mm) e = tlog[base + lane];
      for (uint32_t j = 0; j < mm; j++) {
        const float d = __uint_as_float(__builtin_amdgcn_readlane(e.x, j));
        const uint32_t nbj = __builtin_amdgcn_readlane(e.y, j);
        if (lane == 0) {
          top[ts].d = d;
          top[ts].id = nbj;
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
#ifdef HS_FLAT_DIAG
      uint32_t *dg = a.stats + (size_t)a.nq * 4 + qi * 16;
      dg[4] = (uint32_t)diag_t0;                                          // start, 100 MHz ticks (low word)
      dg[5] = diag_sync_ticks;                                            // ticks spent inside heap replays
      dg[6] = diag_pre;                                                   // hops whose node was chosen before the accept pass
      for (int i = 0; i < 6; i++) dg[8 + i] = diag_ph[i];
      dg[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - diag_t0);   // 100 MHz ticks
      dg[1] = diag_syncs;
      dg[2] = diag_replayed;
      dg[3] = ovf_n | (v2.spilled ? 0x10000u : 0u) | (v2.n2 << 17);
#endif
    }
    a.status[qi] = ST_DONE;
  }
  return 0;
}

template <int METRIC, int S, int D16>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(flat_waves(S, D16)))) flat_kernel(DevIndex ix, SearchArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  lds_u8 *smem = (lds_u8 *)smem_raw;
  for (uint32_t it = blockIdx.x; it < a.nq; it += gridDim.x) {
    const uint32_t qi = (a.phase == 2 && a.order) ? a.order[it] : it;
    if (a.pass_id != 0 && !((1u << a.status[qi]) & a.select_mask)) continue;   // pass 0 takes every query
    const int rc = search_one_flat<METRIC, S, D16>(ix, a, qi, smem);
    if (rc == 3 && threadIdx.x == 0) a.status[qi] = ST_HAZARD;
    wave_sync();
  }
}

template <typename K>
static hipError_t flat_launch(K kern, const DevIndex &ix, const SearchArgs &a, size_t lds, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(std::max(1u, std::min(a.grid, a.nq))), dim3(64), lds, stream, ix, a);
  return hipGetLastError();
}
template <int METRIC, int D16>
static hipError_t flat_launch_d(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  const size_t lds = flat_layout(ix.dim, a.ef, a.fl_nb).total;
  if (a.ef <= 64) return flat_launch(flat_kernel<METRIC, 1, D16>, ix, a, lds, stream);
  if (a.ef <= 128) return flat_launch(flat_kernel<METRIC, 2, D16>, ix, a, lds, stream);
  if (a.ef <= 192) return flat_launch(flat_kernel<METRIC, 3, D16>, ix, a, lds, stream);   // (three slots instead of four: ef = 160 +5.5 %, 192 +3.5 %)
  if (a.ef <= 256) return flat_launch(flat_kernel<METRIC, 4, D16>, ix, a, lds, stream);
  if (a.ef <= 384) return flat_launch(flat_kernel<METRIC, 6, D16>, ix, a, lds, stream);
  return flat_launch(flat_kernel<METRIC, 8, D16>, ix, a, lds, stream);
}
template <int METRIC>
static hipError_t flat_launch_s(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  if (ix.dim == 128) return flat_launch_d<METRIC, 8>(ix, a, stream);
  if (ix.dim == 96) return flat_launch_d<METRIC, 6>(ix, a, stream);
  if (ix.dim == 960) return flat_launch_d<METRIC, 60>(ix, a, stream);
  if (ix.dim > 256) return flat_launch_d<METRIC, -1>(ix, a, stream);
  return flat_launch_d<METRIC, 0>(ix, a, stream);
}

// Parity/debug: a sequence of candidate_set operations through this file's heap code (one wavefront; 1024 slots in LDS, the rest
// in `spill`).  ops: 3 words each {0 = push | 1 = pop, key bits, id}; out_heap: the final array (2 words per element); out_pops:
// the root each pop removed; out_n[0] = final size, out_n[1] = pops.
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 0
__global__ void __launch_bounds__(64) flat_heap_ops_kernel(const uint32_t *ops, uint32_t n_ops, uint2 *spill, uint2 *out_heap, uint2 *out_pops,
                                                            uint32_t *out_n, int wave_pop, uint32_t lds_slots) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  FlatHeap h;
  h.lds = (lds_u8 *)smem_raw;
  h.L = lds_slots;
  h.glob = spill;
  uint32_t size = 0, pops = 0;
  for (uint32_t i = 0; i < n_ops; i++) {
    const uint32_t kind = uni(ops[3 * i]), key = uni(ops[3 * i + 1]), id = uni(ops[3 * i + 2]);
    if (kind == 0) {
      size++;
      if (size < h.L) fh_push<false>(h, size, __uint_as_float(key), id, lane);
      else fh_push<true>(h, size, __uint_as_float(key), id, lane);
    } else if (size > 0) {
      const uint2 root = fh_get<true>(h, 1);
      if (lane == 0) out_pops[pops] = root;
      pops++;
      if (wave_pop && size < h.L) fh_pop_wave(h, size, lane);
      else if (lane == 0) { if (size < h.L) fh_pop<false>(h, size); else fh_pop<true>(h, size); }
      size--;
    }
    wave_sync();
  }
  for (uint32_t s = lane; s < size; s += 64) out_heap[s] = fh_get<true>(h, s + 1);
  if (lane == 0) { out_n[0] = size; out_n[1] = pops; }
}
hipError_t flat_heap_ops(const uint32_t *d_ops, uint32_t n_ops, uint2 *d_spill, uint2 *d_heap, uint2 *d_pops, uint32_t *d_n, int wave_pop,
                         uint32_t lds_slots, hipStream_t stream) {
  hipLaunchKernelGGL(flat_heap_ops_kernel, dim3(1), dim3(64), (size_t)lds_slots * 8, stream, d_ops, n_ops, d_spill, d_heap, d_pops, d_n, wave_pop, lds_slots);
  return hipGetLastError();
}
#endif

hipError_t launch_flatk_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
hipError_t launch_flatk_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream);
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 0
bool flatk_supported(const DevIndex &ix, uint32_t ef, uint32_t k) {
  return ix.tile0 != nullptr && (ix.maxlevel == 0 || ix.uptile != nullptr) && ix.threshold_level == 0 && !ix.has_deleted && ix.n > 0 &&
         ix.n < kFDone && (ix.dim & 15u) == 0 && ef >= k && ef <= 512 && k <= 64;
}
// wavefronts per CU the shape's kernel is resident with (the LDS share of a wave follows from it: capi.cpp plan_flat)
uint32_t flatk_waves_per_cu(uint32_t dim, uint32_t ef) {
  const int s = ef <= 64 ? 1 : ef <= 128 ? 2 : ef <= 192 ? 3 : ef <= 256 ? 4 : ef <= 384 ? 6 : 8;
  const int d16 = dim == 128 ? 8 : dim == 96 ? 6 : dim == 960 ? 60 : dim > 256 ? -1 : 0;
  return 4u * (uint32_t)flat_waves(s, d16);
}
size_t flatk_lds_bytes(uint32_t dim, uint32_t ef, uint32_t nb) { return flat_layout(dim, ef, nb).total; }
hipError_t launch_flatk_l2(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) { return flat_launch_s<METRIC_L2>(ix, a, stream); }
hipError_t launch_flatk(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) {
  return ix.metric == METRIC_L2 ? launch_flatk_l2(ix, a, stream) : launch_flatk_ip(ix, a, stream);
}
#endif
#if !defined(HS_TU_METRIC) || HS_TU_METRIC == 1
hipError_t launch_flatk_ip(const DevIndex &ix, const SearchArgs &a, hipStream_t stream) { return flat_launch_s<METRIC_IP>(ix, a, stream); }
#endif

}  // namespace hs
