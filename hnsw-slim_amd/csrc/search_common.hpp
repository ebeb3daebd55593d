// search_common.hpp -- device structures shared by the one-query-per-wavefront search kernels (beam_search.hip, lean_search.hip):
// the exact two-tier visited set and the candidate min-heap with libstdc++ sift mechanics.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr uint32_t kNone = 0xFFFFFFFFu;
static constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__host__ __device__ inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// Visited set (visited_list_pool.h semantics: test-and-mark, exact).  Tier 1: open addressing in LDS,
// filled to 75 %.  A query that visits more than that freezes tier 1 (read-only from then on) and
// continues in a per-query tier-2 table in global memory (cleared lazily by the wave itself), so a long
// query degrades to L2-latency probes instead of being thrown away and re-run.
struct Visited {
  uint32_t *t1, *t2;
  uint32_t slots1, limit1, slots2, limit2;
  uint32_t n1, n2;
  bool spilled;
  // 16-bit (quotient) form of tier 1, see vis_test_and_mark: qbits = width of the id space, 0 = the 32-bit form above
  uint32_t qbits, qshift, qmask;
};
__device__ __forceinline__ void vis_init(Visited &v, const SearchArgs &a, uint32_t qi, uint32_t *lds_tab, int lane) {
  v.t1 = lds_tab;
  v.slots1 = a.hash_slots;
  v.limit1 = a.hash_slots - (a.hash_slots >> (a.hash_fill_shift ? a.hash_fill_shift : 2));   // 75 % by default, 87.5 % for the lean kernel
  v.t2 = a.fb_spill ? a.fb_spill + (size_t)blockIdx.x * a.spill_slots : (a.spill ? a.spill + (size_t)qi * a.spill_stride : nullptr);
  v.slots2 = a.spill_slots;
  v.limit2 = a.spill_slots - (a.spill_slots >> 2);
  v.n1 = v.n2 = 0;
  v.spilled = false;
  v.qbits = a.vis_bits;
  v.qshift = a.vis_bits ? a.vis_bits - (uint32_t)(__ffs((int)(a.hash_slots >> 2)) - 1) : 0;   // buckets: a power of two
  v.qmask = (1u << v.qshift) - 1u;
  for (uint32_t i = lane; i < a.hash_slots; i += 64) lds_tab[i] = kEmpty;
}
// Call (wave-uniformly) before up to m inserts.  Returns false when even tier 2 is exhausted.
__device__ __forceinline__ bool vis_reserve(Visited &v, uint32_t m, const SearchArgs &a, int lane) {
  if (__builtin_expect(!v.spilled, 1)) {
    if (__builtin_expect(v.n1 + m <= v.limit1, 1)) return true;
    if (!v.t2) return false;
    for (uint32_t i = lane; i < v.slots2; i += 64) v.t2[i] = kEmpty;
    __threadfence_block();
    v.spilled = true;
    if (lane == 0) atomicAdd(a.counters + 3, 1u);
  }
  return v.n2 + m <= v.limit2;
}
__device__ __forceinline__ void vis_commit(Visited &v, uint32_t cnt) {
  if (v.spilled) v.n2 += cnt;
  else v.n1 += cnt;
}
// Both tiers are arrays of 4-slot buckets probed with one 16-byte read: an id sits in the first bucket,
// in probe order from its home bucket, that had a free slot when it arrived (nothing is ever deleted), so
// a lookup ends at the first bucket that contains the id or still has a free slot.
// Slots of a bucket fill in order (an insert always takes the first free one), so the free slots are a suffix and
// -- ids being below 2^31, kEmpty = 0xFFFFFFFF the only value with the sign bit set -- their number is minus the sum
// of the four arithmetic sign extensions.  A dozen vector instructions instead of a compare/select ladder.
__device__ __forceinline__ int bucket_scan(const uint4 &w, uint32_t id) {  // -2 found, -1 full, else free slot
  const uint32_t hit = min(min(w.x ^ id, w.y ^ id), min(w.z ^ id, w.w ^ id));
  const int used = 4 + (((int)w.x >> 31) + ((int)w.y >> 31) + ((int)w.z >> 31) + ((int)w.w >> 31));
  return hit == 0 ? -2 : (used < 4 ? used : -1);
}
__device__ __forceinline__ bool vis_insert_t2(const Visited &v, uint32_t id, uint32_t h) {
  const uint32_t nb2 = v.slots2 >> 2;
  uint32_t b = __umulhi(h, nb2);
  while (true) {
    const uint4 w = reinterpret_cast<const uint4 *>(v.t2)[b];
    const int e = bucket_scan(w, id);
    if (e == -2) return false;
    if (e >= 0) {
      const uint32_t old = atomicCAS(&v.t2[b * 4 + e], kEmpty, id);
      if (old == kEmpty) return true;
      if (old == id) return false;
      continue;
    }
    if (++b == nb2) b = 0;
  }
}
__device__ __forceinline__ bool vis_insert(const Visited &v, uint32_t id) {
  const uint32_t h = id * 2654435761u;
  const uint32_t nb1 = v.slots1 >> 2;
  uint32_t b = __umulhi(h, nb1);
  while (true) {
    const uint4 w = reinterpret_cast<const uint4 *>(v.t1)[b];
    const int e = bucket_scan(w, id);
    if (e == -2) return false;
    if (e >= 0) {
      if (__builtin_expect(v.spilled, 0)) break;  // tier 1 is frozen: the id is not in it
      const uint32_t old = lds_cas(&v.t1[b * 4 + e], kEmpty, id);
      if (__builtin_expect(old == kEmpty, 1)) return true;
      if (old == id) return false;
      continue;  // another lane of this wave took the slot: look at the bucket again
    }
    if (++b == nb1) b = 0;
  }
  return vis_insert_t2(v, id, h);
}

// ---- tier 1 in 16-bit slots ------------------------------------------------------------------------------------------
// The longest queries are the ones that outgrow tier 1, and they are also the ones a launch ends on: once tier 1 is frozen
// every lookup of a new id walks its probe sequence to a bucket with a free slot and then goes to global memory (measured:
// the longest query of a batch runs 26 % faster with a visited set that never spills).  This form holds twice the ids in the
// same LDS.  With the id space B bits wide, h = id * odd mod 2^B is a bijection; the bucket is the top bits of h and only the
// remaining B - log2(buckets) <= 16 bits are stored, eight to a 16-byte bucket, 0xFFFF = free.  There is no probing: a
// bucket that is full sends ITS later arrivals to tier 2 (the 32-bit table in global memory) and nobody else's, and a lookup
// reads one bucket, plus tier 2 only if that bucket is full.  Exact like the other form: an id is in exactly one place.
static constexpr uint32_t kQ16SpillCount = 128;
__device__ __forceinline__ uint32_t zero_halves(uint32_t x) { return (x - 0x00010001u) & ~x & 0x80008000u; }
// Wave-level test-and-mark of one id per lane (valid lanes).  Returns true in the lanes whose id was new; `fail` is set
// (wave-uniformly) when tier 2 is needed and missing or exhausted.
__device__ __forceinline__ bool vis_test_and_mark_q16(Visited &v, uint32_t id, bool valid, const SearchArgs &a, int lane, bool &fail) {
  const uint32_t h32 = id * 2654435761u;
  const uint32_t h = h32 & (v.qbits >= 32 ? 0xFFFFFFFFu : (1u << v.qbits) - 1u);
  const uint32_t b = h >> v.qshift, tag = h & v.qmask;
  const uint32_t tag2 = tag | (tag << 16);
  bool isnew = false;
  bool t2 = valid && tag == 0xFFFFu;   // the one remainder that looks like a free slot (only when 16 bits are stored)
  bool todo = valid && !t2;
  while (todo) {
    const uint4 w = reinterpret_cast<const uint4 *>(v.t1)[b];
    const uint32_t hit = zero_halves(w.x ^ tag2) | zero_halves(w.y ^ tag2) | zero_halves(w.z ^ tag2) | zero_halves(w.w ^ tag2);
    if (hit) break;   // already in the set
    // slots fill in order, so the free ones are a suffix and their number is the number of 0xFFFF halves
    const uint32_t nfree = __popc(zero_halves(~w.x)) + __popc(zero_halves(~w.y)) + __popc(zero_halves(~w.z)) + __popc(zero_halves(~w.w));
    if (nfree == 0) { t2 = true; break; }
    const uint32_t e = 8u - nfree;
    const uint32_t wd = (e >> 1) == 0 ? w.x : (e >> 1) == 1 ? w.y : (e >> 1) == 2 ? w.z : w.w;
    const uint32_t nw = (e & 1u) ? ((wd & 0x0000FFFFu) | (tag << 16)) : ((wd & 0xFFFF0000u) | tag);
    const uint32_t got = lds_cas(&v.t1[b * 4 + (e >> 1)], wd, nw);
    if (got == wd) { isnew = true; break; }
    // another lane of this wave changed the word: look at the bucket again
  }
  const unsigned long long m2 = hs_ballot(t2);
  if (__builtin_expect(m2 != 0, 0)) {
    if (!v.spilled) {   // (the launches that use this form always come with a tier-2 region: SearchArgs::spill)
      for (uint32_t i = lane; i < v.slots2; i += 64) v.t2[i] = kEmpty;
      __threadfence_block();
      v.spilled = true;
    }
    if (v.n2 + (uint32_t)__popcll(m2) > v.limit2) { fail = true; return false; }
    if (t2) isnew = vis_insert_t2(v, id, h32);
    // (the spill counter feeds hs_search_check's sizing rule: here a query counts once a real share of its ids lives in tier 2,
    //  not when one full bucket sends its first id there)
    const uint32_t before = v.n2;
    v.n2 += __popcll(hs_ballot(t2 && isnew));
    if (before < kQ16SpillCount && v.n2 >= kQ16SpillCount && lane == 0) atomicAdd(a.counters + 3, 1u);
  }
  return isnew;
}

// Wave-level: mark one id (the same value in every lane) visited.  False: the 16-bit form needed a tier 2 that is missing / full.
__device__ __forceinline__ bool vis_mark_one(Visited &v, uint32_t id, const SearchArgs &a, int lane) {
  if (v.qbits) {
    bool fail = false;
    vis_test_and_mark_q16(v, id, lane == 0, a, lane, fail);
    return !fail;
  }
  if (lane == 0) vis_insert(v, id);
  v.n1++;
  return true;
}

// Candidate min-heap, element i stored at slot i+1 so that the two children of any node share one aligned
// 16-byte read.  Slots [0, L) live in LDS (L even, so a child pair never straddles), the rest in a per-query
// region of global memory: a query whose heap outgrows its LDS share keeps going (its deepest heap levels
// pay L2 latency) instead of being thrown away.  Same sift decisions as std::push_heap / pop_heap with
// compare_by_first_rev (hnswalg_slim.h:177-183).
struct CandHeap {
  uint2 *lds, *glob;
  uint32_t L;  // LDS slots
};
// T2 = false: the heap is known to fit its LDS share -> LDS-only code (no global access, so no vmcnt wait
// is generated and the row loads in flight around the pop stay in flight).
template <bool T2>
__device__ __forceinline__ uint2 ch_get(const CandHeap &h, uint32_t s) {
  if (!T2) return h.lds[s];
  return s < h.L ? h.lds[s] : h.glob[s - h.L];
}
template <bool T2>
__device__ __forceinline__ void ch_set(const CandHeap &h, uint32_t s, uint2 v) {
  if (!T2 || s < h.L) h.lds[s] = v;
  else h.glob[s - h.L] = v;
}
template <bool T2>
__device__ __forceinline__ uint4 ch_get2(const CandHeap &h, uint32_t s /*even*/) {
  if (!T2) return *reinterpret_cast<const uint4 *>(&h.lds[s]);
  return s < h.L ? *reinterpret_cast<const uint4 *>(&h.lds[s]) : *reinterpret_cast<const uint4 *>(&h.glob[s - h.L]);
}

// std::push_heap == std::__push_heap(first, hole = n-1, top = 0, value): the value rises past every
// consecutive ancestor that compares strictly greater.  All ancestors of slot n-1 are known up front
// ((n >> t) - 1 for t = 1, 2, ...), so the whole wave does it in one read and one write round: lane t-1
// reads ancestor t, a ballot finds where the rise stops, the passed ancestors each move one level down.
template <bool T2>
__device__ __forceinline__ void cand_push_t(const CandHeap &h, uint32_t n /*size incl. new*/, float d, uint32_t id, int lane) {
  const uint32_t anc = n >> (lane + 1);          // (index + 1) of this lane's ancestor; 0 = beyond the root
  const bool has = anc != 0 && lane < 31;
  uint2 p = make_uint2(0, 0);
  if (has) p = ch_get<T2>(h, anc);               // element anc-1 lives at slot anc
  const unsigned long long rises = hs_ballot(has && __uint_as_float(p.x) > d);
  const uint32_t r = __ffsll((long long)~rises) - 1;  // number of consecutive ancestors passed
  if ((uint32_t)lane < r) ch_set<T2>(h, n >> lane, p);  // ancestor t moves to the path node below it ((n >> (t-1)) - 1)
  if ((uint32_t)lane == r) ch_set<T2>(h, n >> r, make_uint2(__float_as_uint(d), id));
}
__device__ __forceinline__ void cand_push(const CandHeap &h, uint32_t n, float d, uint32_t id, int lane) {
  if (__builtin_expect(n < h.L, 1)) cand_push_t<false>(h, n, d, id, lane);  // slot n is the deepest slot touched
  else cand_push_t<true>(h, n, d, id, lane);
}
// std::pop_heap (one lane).  Returns nothing; the popped root was read by the caller beforehand.
template <bool T2>
__device__ __forceinline__ void cand_pop_t(const CandHeap &h, uint32_t n /*size before pop*/) {
  if (n <= 1) return;
  const uint2 v = ch_get<T2>(h, n);  // a[n-1]
  const uint32_t len = n - 1;
  uint32_t hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    const uint4 two = ch_get2<T2>(h, child);  // a[child-1], a[child]
    const bool left = __uint_as_float(two.z) > __uint_as_float(two.x);  // comp(a[child], a[child-1])
    ch_set<T2>(h, hole + 1, left ? make_uint2(two.x, two.y) : make_uint2(two.z, two.w));
    child = left ? child - 1 : child;
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    ch_set<T2>(h, hole + 1, ch_get<T2>(h, child));  // a[child-1]
    hole = child - 1;
  }
  const float vd = __uint_as_float(v.x);
  while (hole > 0) {
    const uint32_t parent = (hole - 1) >> 1;
    const uint2 p = ch_get<T2>(h, parent + 1);
    if (!(__uint_as_float(p.x) > vd)) break;
    ch_set<T2>(h, hole + 1, p);
    hole = parent;
  }
  ch_set<T2>(h, hole + 1, v);
}
__device__ __forceinline__ void cand_pop(const CandHeap &h, uint32_t n) {
  if (__builtin_expect(n < h.L, 1)) cand_pop_t<false>(h, n);
  else cand_pop_t<true>(h, n);
}

__device__ __forceinline__ void flag_query(const SearchArgs &a, uint32_t qi, uint32_t status, uint32_t counter, int lane) {
  if (lane == 0) {
    a.status[qi] = status;
    atomicAdd(a.counters + counter, 1u);
  }
}


}  // namespace hs
