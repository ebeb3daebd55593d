// slimq_search.hip -- HNSW-SlimQ search on gfx950: one wavefront per query.
//
// Path (reference, relative to /root/reference/third_party/):
//   HierarchicalNSWSlimQ::searchKnn(q, k, result)     hnswlib/hnswalg_slimq.h:1810-1924
//     rotator_->rotate                                rabitqlib/utils/rotator.hpp:370-423
//     SplitSingleQuery ctor                           rabitqlib/index/query.hpp:112-156
//     q_to_centroids                                  hnswlib/hnswalg_slimq.h:1822-1848
//     greedy descent on estimated distances           hnswlib/hnswalg_slimq.h:1850-1901
//     searchBaseLayerST<true>(.., k, ..)              hnswlib/hnswalg_slimq.h:688-759
//       SearchBuffer insert / is_full / pop           hnswlib/hnswalg_slimq.h:80-151
//       get_bin_est -> split_single_estdist           hnswlib/hnswalg_slimq.h:408-440, rabitqlib/index/estimator.hpp:164-188
//       exact rerank + k-bounded max-heap             hnswlib/hnswalg_slimq.h:747-757
//
// Two kernels, one wavefront per query each:
//   * slimq_prep_kernel: query preparation in LDS.  The Hadamard butterflies are element-wise (bit-exact in any lane
//     order); the float reductions are plain left-to-right sums (the definition rabitq_host.hpp documents), one
//     reduction per lane, all of them in one loop.  Output: a small per-query record (delta, vl, k1xsumq, g_add per
//     cluster, bit planes).  Kept apart so that the search kernel's registers and LDS are sized by the traversal.
//   * slimq_kernel: the traversal.
//     - The SearchBuffer (sorted array, capacity ef) lives in registers: rank r in lane r/S, slot r%S, +inf/checked
//       padding.  Lower-bound position = popcount of a ballot per slot, the memmove = one DPP wave_shr:1 of the last slot +
//       per-slot selects, the new entry two v_writelane; cur_ and the last key are scalars.
//     - Level 0 and the upper levels are read from FUSED tiles: slot j of a node's tile is neighbour j's whole RaBitQ
//       record (16 B of factors + padded/8 B of sign code, 32 B at d=128) with the neighbour id in the header -- one
//       dependent HBM access per expansion.  One lane per neighbour; the estimator is AND + popcount against the
//       query's 4 bit planes (SGPRs for codes of <= 2 words, LDS otherwise; compile-time code lengths load the whole
//       code up front).  Candidates that survive the (monotone) is_full pre-test are inserted in adjacency order.
//     - Exact re-ranks are batched 16 at a time (4 lanes per row, the distance recipe of dist_recipe.hpp) and pushed
//       into the k-heap (registers, element j in lane j) in expansion order, so the heap array ends up as the
//       reference's; a push that provably leaves the array untouched is skipped.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "heap_emul.hpp"
#include "rabitq_est.hpp"
#include "slimq_engine.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr uint32_t kNoneQ = 0xFFFFFFFFu;
static constexpr uint32_t kChecked = 0x80000000u;
// re-rank: 16-byte loads in flight per lane and round.  Long codes (d = 768: 12 words) already hold ~115 VGPRs for the estimator's
// loads, so their re-rank takes twelve at a time for free (d = 768: 4 dependent rounds per flush instead of 12)
template <int NBLK> struct RerankDepth { static constexpr int v = NBLK >= 8 ? 12 : 4; };

__host__ __device__ inline uint32_t al16(uint32_t x) { return (x + 15u) & ~15u; }

struct SlimQLds { uint32_t off_q, off_planes, off_red, off_hash, off_heap, off_pend, off_pd, total; };
__host__ __device__ inline SlimQLds slimq_layout(uint32_t dim, uint32_t padded, uint32_t ncl, uint32_t k, uint32_t hash_slots) {
  SlimQLds l;
  uint32_t o = 0;
  l.off_q = o; o += al16(dim * 4);
  l.off_planes = o; o += al16(padded / 64 * 4 * 8);
  l.off_red = o; o += al16(ncl * 4);         // g_add of cluster c
  l.off_hash = o; o += al16(hash_slots * 4);
  l.off_heap = o; o += al16((k + 1) * 8);
  l.off_pend = o; o += 64;
  l.off_pd = o; o += 64;
  l.total = o;
  return l;
}
size_t slimq_lds_bytes(uint32_t dim, uint32_t padded, uint32_t ncl, uint32_t k, uint32_t hash_slots) {
  return slimq_layout(dim, padded, ncl, k, hash_slots).total;
}
bool slimq_supported(uint32_t pool_cap) { return pool_cap >= 1 && pool_cap <= 1024; }

// ---- expanded-node set: open addressing (linear probing) in LDS; the per-lane lookup of the estimate pre-test.  The pop's own
//      test-and-add reads 64 slots of the probe sequence in one round (slimq_one) ------------------------------------------
__device__ __forceinline__ uint32_t hash_of(uint32_t id, uint32_t mask) { return (id * 2654435761u) >> 7 & mask; }
__device__ __forceinline__ bool set_has(const uint32_t *tab, uint32_t mask, uint32_t id) {
  uint32_t h = hash_of(id, mask);
  for (;;) {
    const uint32_t v = tab[h];
    if (v == id) return true;
    if (v == kNoneQ) return false;
    h = (h + 1) & mask;
  }
}

// ---- SearchBuffer in registers --------------------------------------------------------------------------------
// Rank r lives in lane r / S, slot r % S (interleaved, as the flat kernel's result set: inserting at rank p is ONE DPP
// wave_shr:1 of the last slot plus per-slot selects and two v_writelane -- no carries from slot to slot; round 3: the slot-major
// layout it replaces chained a lane-63 read into the next slot's shift, 1300 shader cycles per insertion at ef = 256,
// profiles/r03_slimq_sift1m_phases_before.log).  Ranks at or beyond `size` hold key = +inf, val = 0xFFFFFFFF (reads as
// "checked"), so neither the position count nor the unchecked scan needs a size mask; entries pushed beyond the capacity are
// masked by rank < cap in the unchecked scan and can never compare below a candidate that passed is_full().
static constexpr uint32_t kNoRank = 0xFFFFFFFFu;
template <int S>
struct PoolState {
  uint32_t size, cap, cur;   // cur = rank of the first unchecked entry (SearchBuffer::cur_), kNoRank when none
  float last;                // key at rank cap-1 once size == cap, else +inf: is_full(d) == (d > last)
  unsigned long long inr[S]; // lanes whose slot-s rank is below the capacity
};
template <int S>
__device__ __forceinline__ uint32_t pool_at(const uint32_t (&v)[S], uint32_t r) {   // register value at rank r (uniform)
  const uint32_t l = r / S, sl = r % S;
  uint32_t x = __builtin_amdgcn_readlane(v[0], l);
#pragma unroll
  for (int s = 1; s < S; s++) {
    const uint32_t t = __builtin_amdgcn_readlane(v[s], l);
    x = sl == (uint32_t)s ? t : x;
  }
  return x;
}
// insert(): lower-bound position, shift the tail up by one (:112-119)
template <int S>
__device__ __forceinline__ void pool_insert(float (&key)[S], uint32_t (&val)[S], PoolState<S> &p, float d, uint32_t id, int lane) {
  uint32_t pos = 0;
#pragma unroll
  for (int s = 0; s < S; s++) pos += __popcll(hs_ballot(key[s] < d));
  const uint32_t laneS = (uint32_t)lane * S;
  const float upk = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(key[S - 1]), 0x138, 0xf, 0xf, false));   // wave_shr:1
  const uint32_t upv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)val[S - 1], 0x138, 0xf, 0xf, false);
#pragma unroll
  for (int s = S - 1; s >= 1; s--) {
    const bool gt = laneS + (uint32_t)s > pos;
    key[s] = gt ? key[s - 1] : key[s];
    val[s] = gt ? val[s - 1] : val[s];
  }
  {
    const bool gt = laneS > pos;
    key[0] = gt ? upk : key[0];
    val[0] = gt ? upv : val[0];
  }
  {   // the new entry: one v_writelane per register, in the slot the rank falls in
    const uint32_t pl = uni(pos / S), psl = pos % S, du = uni(__float_as_uint(d)), iu = uni(id);
#pragma unroll
    for (int s = 0; s < S; s++)
      if (psl == (uint32_t)s) {
        key[s] = __uint_as_float(write_lane(__float_as_uint(key[s]), du, pl));
        val[s] = write_lane(val[s], iu, pl);
      }
  }
  p.size = min(p.size + 1, p.cap);
  p.cur = min(p.cur, pos);
  if (p.size == p.cap) {
    uint32_t kb[S];
#pragma unroll
    for (int s = 0; s < S; s++) kb[s] = __float_as_uint(key[s]);
    p.last = __uint_as_float(pool_at<S>(kb, p.cap - 1));
  }
}
// pop(): the entry at cur_, then advance cur_ to the next unchecked rank (:126-134).  Every rank below cur_ is checked, so the
// next one is the lowest unchecked rank below the capacity: lowest lane with any such slot, lowest such slot in it.
template <int S>
__device__ __forceinline__ uint32_t pool_pop(uint32_t (&val)[S], PoolState<S> &p, int lane) {
  const uint32_t lc = p.cur / S, sc = p.cur % S;
  const uint32_t id = pool_at<S>(val, p.cur);
  unsigned long long um[S], any = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    if ((uint32_t)s == sc && (uint32_t)lane == lc) val[s] |= kChecked;
    um[s] = hs_ballot((int)val[s] >= 0) & p.inr[s];
    any |= um[s];
  }
  uint32_t next = kNoRank;
  if (any) {
    const uint32_t pl = (uint32_t)__ffsll((long long)any) - 1;
    uint32_t sl = S - 1;
#pragma unroll
    for (int s = S - 2; s >= 0; s--) sl = ((um[s] >> pl) & 1ull) ? (uint32_t)s : sl;
    next = pl * S + sl;
  }
  p.cur = next;
  return id;
}

// ---- k-bounded result heap in registers: element j lives in lane j (k < 64) -------------------------------------
// Every lane runs the same (wave-uniform) sift code of heap_emul.hpp; element reads are v_readlane, writes a
// lane-masked move -- a few cycles per step instead of an LDS round trip on one lane.
struct RegHeap {
  float &key;
  uint32_t &id;
  int lane;
  struct Ref {
    const RegHeap &h;
    long j;
    __device__ __forceinline__ operator Pair() const {
      return Pair{__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h.key), (int)j)), (uint32_t)__builtin_amdgcn_readlane(h.id, (int)j)};
    }
    __device__ __forceinline__ const Ref &operator=(const Pair &p) const {
      if (h.lane == (int)j) { h.key = p.d; h.id = p.id; }
      return *this;
    }
    __device__ __forceinline__ const Ref &operator=(const Ref &o) const { return *this = (Pair)o; }
  };
  __device__ __forceinline__ Ref operator[](long j) const { return Ref{*this, j}; }
};

// ---- estimator of one record (any lane, its own record) ------------------------------------------------------
// r -> {f_add, f_rescale, cluster id, (f_error | neighbour id)} + sign code; h = the header, already loaded
template <int NBLK>
__device__ __forceinline__ float est_rec(const DevSlimQ &sq, const uint64_t *planes, const float *gadd, float delta, float vl,
                                         float k1, const uint32_t *r, const uint4 &h) {
  const uint64_t *code = reinterpret_cast<const uint64_t *>(r + 4);
  float ipq;
  if (NBLK > 0) {
    uint64_t x[NBLK > 0 ? NBLK : 2];
#pragma unroll
    for (int b = 0; b < NBLK; b += 2) {
      const uint4 w = *reinterpret_cast<const uint4 *>(code + b);
      x[b] = (uint64_t)w.x | ((uint64_t)w.y << 32);
      x[b + 1] = (uint64_t)w.z | ((uint64_t)w.w << 32);
    }
    ipq = rq_ip_x0_qr(x, planes, NBLK, delta, vl);
  } else {
    // runtime code length: four 8-byte words in flight per round (integer sums, so the grouping changes nothing)
    const uint32_t nblk = sq.padded >> 6;
    uint32_t ip = 0, pc = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 4) {
      const uint32_t nb = min(4u, nblk - b0);
      uint64_t x[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++)
        if (i < nb) x[i] = code[b0 + i];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++)
        if (i < nb) {
          pc += rq_popc64(x[i]);
#pragma unroll
          for (int j = 0; j < 4; j++) ip += (uint32_t)rq_popc64(x[i] & planes[(b0 + i) * 4 + j]) << j;
        }
    }
    const float a1 = delta * (float)ip, c1 = vl * (float)pc;   // rq_ip_x0_qr's fp32 expression
    ipq = a1 + c1;
  }
  return rq_est_dist(__uint_as_float(h.x), gadd[h.z], __uint_as_float(h.y), ipq, k1);
}
template <int NBLK>
__device__ __forceinline__ float est_one(const DevSlimQ &sq, const uint64_t *planes, const float *gadd, float delta, float vl,
                                         float k1, uint32_t id) {
  const uint32_t *r = sq.rec + (size_t)id * sq.rec_words;
  const uint4 h = *reinterpret_cast<const uint4 *>(r);
  return est_rec<NBLK>(sq, planes, gadd, delta, vl, k1, r, h);
}

// ---- query preparation ---------------------------------------------------------------------------------------
__device__ __forceinline__ void flip_signs(float *y, const uint8_t *f, uint32_t n, int lane) {
#pragma unroll 1
  for (uint32_t i = lane; i < n; i += 64)
    if ((f[i >> 3] >> (i & 7)) & 1) y[i] = -y[i];
}
// in-place Walsh-Hadamard over n (power of two) floats in LDS, butterfly stages in ascending stride, then * scale
__device__ __forceinline__ void fht_lds(float *y, uint32_t n, float scale, int lane) {
  for (uint32_t h = 1; h < n; h <<= 1) {
    wave_sync();
#pragma unroll 1
    for (uint32_t t = lane; t < n / 2; t += 64) {
      const uint32_t j = ((t / h) * 2 * h) + (t % h);
      const float a = y[j], b = y[j + h];
      y[j] = a + b;
      y[j + h] = a - b;
    }
  }
  wave_sync();
#pragma unroll 1
  for (uint32_t i = lane; i < n; i += 64) y[i] *= scale;
}
__device__ __forceinline__ void rotate_lds(const DevSlimQ &sq, float *y, int lane) {
  const uint32_t P = sq.padded, T = sq.trunc, nb = P / 8;
  if (T == P) {
    for (int r = 0; r < 4; r++) {
      wave_sync();
      flip_signs(y, sq.flips + r * nb, P, lane);
      fht_lds(y, T, sq.fht_scale, lane);
    }
    wave_sync();
    return;
  }
  for (int r = 0; r < 4; r++) {
    wave_sync();
    flip_signs(y, sq.flips + r * nb, P, lane);
    fht_lds((r & 1) ? y + (P - T) : y, T, sq.fht_scale, lane);
    wave_sync();
#pragma unroll 1
    for (uint32_t i = lane; i < P / 2; i += 64) {  // kacs_walk
      const float a = y[i], b = y[i + P / 2];
      y[i] = a + b;
      y[i + P / 2] = a - b;
    }
  }
  wave_sync();
#pragma unroll 1
  for (uint32_t i = lane; i < P; i += 64) y[i] *= 0.25f;
  wave_sync();
}

// Per-query preparation record in global memory (prep_words(ncl, padded) u32 words):
//   [0] delta [1] vl [2] k1xsumq [3] -   [4 .. 4+ncl) g_add per cluster   [align 2] 4 bit planes per 64-dim block (u64)
__host__ __device__ inline uint32_t prep_planes_off(uint32_t ncl) { return (4 + ncl + 1) & ~1u; }
__host__ __device__ inline uint32_t prep_words_of(uint32_t ncl, uint32_t padded) { return (prep_planes_off(ncl) + padded / 64 * 8 + 3) & ~3u; }
uint32_t slimq_prep_words(uint32_t ncl, uint32_t padded) { return prep_words_of(ncl, padded); }

// One wavefront per query: rotation (rotator.hpp:370-423), SplitSingleQuery (query.hpp:112-156), centroid table
// (hnswalg_slimq.h:1822-1848).  Kept out of the search kernel so that its registers and LDS are sized by the
// traversal alone.
template <int METRIC>
__global__ void __launch_bounds__(64) slimq_prep_kernel(DevSlimQ sq, uint32_t dim, const float *queries, uint32_t nq,
                                                        uint32_t *prep, float *dbg_y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const uint32_t P = sq.padded;
  float *y = reinterpret_cast<float *>(smem);
  float *u = y + P;
  uint64_t *planes_lds = reinterpret_cast<uint64_t *>(u + P);
  float *red = reinterpret_cast<float *>(planes_lds + P / 64 * 4);
  float *gadd = red + 4;
  const uint32_t pw = prep_words_of(sq.ncl, P);
  for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    wave_sync();
#pragma unroll 1
    for (uint32_t i = lane; i < P; i += 64) {
      y[i] = i < dim ? queries[(size_t)qi * dim + i] : 0.f;
    }
    rotate_lds(sq, y, lane);

    // Reductions, one per lane and all in one loop: sum(q') (= <q', 1>), |q'|^2 (= |q' - 0|^2) and per centroid
    // |q' - c|^2 (L2) or <q', c> (IP).  Each is a strict left-to-right fp32 sum (the definition rabitq_host.hpp
    // documents); x * 1 and x - 0 are exact, so the two query-only sums share the centroid loops' arithmetic.
    // sqrt and division below must be the correctly rounded ones (plain sqrtf / operator/ under hipcc's default
    // -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn maps to the NATIVE (1 ulp) square root in this ROCm.
    for (uint32_t t0 = 0; t0 < 2 + sq.ncl; t0 += 64) {
      const uint32_t t = t0 + lane;
      const bool live = t < 2 + sq.ncl, is_c = live && t >= 2;
      const bool l2form = t == 1 || (is_c && METRIC == METRIC_L2);
      const float4 *ce = reinterpret_cast<const float4 *>(sq.cent + (size_t)(is_c ? t - 2 : 0) * P);
      const float cfill = t == 0 ? 1.f : 0.f;
      float s = 0.f;
      for (uint32_t i = 0; i < P; i += 16) {
        float c[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float4 v = make_float4(cfill, cfill, cfill, cfill);
          if (is_c) v = ce[(i >> 2) + j];
          c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const float yv = y[i + j];
          const float x = l2form ? yv - c[j] : yv;
          const float m = x * (l2form ? x : c[j]);
          s += m;
        }
      }
      if (t == 0) red[0] = s;
      else if (t == 1) red[1] = s;
      else if (is_c) {
        if (METRIC == METRIC_L2) {
          const float nrm = __builtin_sqrtf(s);
          gadd[t - 2] = nrm * nrm;        // get_bin_est: g_add = norm * norm (hnswalg_slimq.h:436)
        } else {
          gadd[t - 2] = -s;               // g_add = -<q', c>  (:423)
        }
      }
    }
    wave_sync();
    const float nrm = __builtin_sqrtf(red[1]);
    const float k1 = red[0] * (-0.5f);
    // 4-bit scalar code of q' (1 sign bit + 3 magnitude bits), its bit planes, and u = code - 7.5
    for (uint32_t b = 0; b < P / 64; b++) {
      const float v = y[b * 64 + lane];
      const float oa = fabsf(((v) / (nrm)));
      int c = (int)(sq.t_const * (double)oa + 1e-5);
      c = c >= 8 ? 7 : c;
      if (v < 0.f) c = (~c) & 7;
      c += v > 0.f ? 8 : 0;
      u[b * 64 + lane] = (float)c + (-7.5f);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned long long m = hs_ballot((c >> j) & 1);
        if (lane == 0) planes_lds[b * 4 + j] = __brevll(m);   // dimension i -> bit 63 - i%64
      }
    }
    wave_sync();
    {  // <q', u> on lane 0 and |u|^2 on lane 1, same loop
      float s = 0.f;
      for (uint32_t i = 0; i < P; i += 4) {
        const float4 uv = *reinterpret_cast<const float4 *>(u + i), yv = *reinterpret_cast<const float4 *>(y + i);
        s += (lane == 0 ? yv.x : uv.x) * uv.x;
        s += (lane == 0 ? yv.y : uv.y) * uv.y;
        s += (lane == 0 ? yv.z : uv.z) * uv.z;
        s += (lane == 0 ? yv.w : uv.w) * uv.w;
      }
      if (lane < 2) red[2 + lane] = s;
    }
    wave_sync();
    const float nq = __builtin_sqrtf(red[3]);
    const float cos_sim = ((red[2]) / (nrm * nq));
    const float delta = ((nrm) / (nq)) * cos_sim;
    const float vl = delta * (-7.5f);
    wave_sync();
    uint32_t *o = prep + (size_t)qi * pw;
    if (lane == 0) { o[0] = __float_as_uint(delta); o[1] = __float_as_uint(vl); o[2] = __float_as_uint(k1); o[3] = 0u; }
    for (uint32_t i = lane; i < sq.ncl; i += 64) o[4 + i] = __float_as_uint(gadd[i]);
    const uint32_t *plw = reinterpret_cast<const uint32_t *>(planes_lds);
    for (uint32_t i = lane; i < P / 64 * 8; i += 64) o[prep_planes_off(sq.ncl) + i] = plw[i];
    if (dbg_y)
      for (uint32_t i = lane; i < P; i += 64) dbg_y[(size_t)qi * P + i] = y[i];
  }
}
hipError_t launch_slimq_prep(const DevSlimQ &sq, uint32_t dim, int metric, const float *queries, uint32_t nq, uint32_t *prep,
                             float *dbg_y, hipStream_t stream) {
  const size_t lds = (size_t)sq.padded * 8 + sq.padded / 64 * 32 + (sq.ncl + 4) * 4 + 16;
  if (metric == METRIC_L2) hipLaunchKernelGGL(slimq_prep_kernel<METRIC_L2>, dim3(nq), dim3(64), lds, stream, sq, dim, queries, nq, prep, dbg_y);
  else hipLaunchKernelGGL(slimq_prep_kernel<METRIC_IP>, dim3(nq), dim3(64), lds, stream, sq, dim, queries, nq, prep, dbg_y);
  return hipGetLastError();
}

template <int METRIC, int S, int NBLK, bool DBG = false>
__device__ int slimq_one(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;
  const SlimQLds L = slimq_layout(ix.dim, sq.padded, sq.ncl, a.k, a.fb_tab ? 0u : a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  uint64_t *planes_lds = reinterpret_cast<uint64_t *>(smem + L.off_planes);
  float *gadd = reinterpret_cast<float *>(smem + L.off_red);
  uint32_t *tab = a.fb_tab ? a.fb_tab + (size_t)blockIdx.x * a.hash_slots : reinterpret_cast<uint32_t *>(smem + L.off_hash);
  Pair *heap = reinterpret_cast<Pair *>(smem + L.off_heap);
  uint32_t *pend = reinterpret_cast<uint32_t *>(smem + L.off_pend);
  float *pd = reinterpret_cast<float *>(smem + L.off_pd);
  const uint32_t P = sq.padded, mask = a.hash_slots - 1;
  const uint32_t *pr = a.prep + (size_t)qi * prep_words_of(sq.ncl, P);

  wave_sync();
#pragma unroll 1
  for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
#pragma unroll 1
  for (uint32_t i = lane; i < a.hash_slots; i += 64) tab[i] = kNoneQ;
#pragma unroll 1
  for (uint32_t i = lane; i < sq.ncl; i += 64) gadd[i] = __uint_as_float(pr[4 + i]);
  const float delta = __uint_as_float(uni(pr[0])), vl = __uint_as_float(uni(pr[1])), k1 = __uint_as_float(uni(pr[2]));
  const uint64_t *planes_g = reinterpret_cast<const uint64_t *>(pr + prep_planes_off(sq.ncl));
  constexpr bool kScalarPlanes = NBLK > 0 && NBLK <= 2;
  if (!kScalarPlanes) {
#pragma unroll 1
    for (uint32_t i = lane; i < P / 64 * 4; i += 64) planes_lds[i] = planes_g[i];
  }
  wave_sync();
  // short codes: the query's bit planes are wave-uniform -> keep them in SGPRs (no LDS reads in the estimator);
  // longer compile-time codes (d = 768: 12 words) keep the planes in LDS but load the whole sign code up front
  uint64_t upl[kScalarPlanes ? NBLK * 4 : 1];
  if (kScalarPlanes) {
#pragma unroll
    for (int i = 0; i < NBLK * 4; i++) {
      const uint64_t v = planes_g[i];
      upl[i] = (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32);
    }
  }
  const uint64_t *planes = kScalarPlanes ? upl : planes_lds;

  uint32_t n_hops = 0, n_est = 1, n_ins = 0, n_rev = 0, n_tr = 0;
  // entry point and greedy descent on estimated distances (:1850-1901)
  uint32_t cur = ix.enterpoint, cur_b = sq.ep_base;   // cur_b = up_base[cur], carried along with cur
  float curd = 0.f;
  if (a.phase == 2) {   // the descent ran in an earlier launch
    const uint4 e = a.entry[qi];
    cur = uni(e.x);
    curd = unif(__uint_as_float(e.y));
    n_est = uni(e.z);
  } else {
  curd = unif(est_one<NBLK>(sq, planes, gadd, delta, vl, k1, cur));
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      if (sq.uptile) {
        // fused upper-level tile of (cur, lvl): the neighbours' records inline, each followed by the neighbour's
        // own up_base -- one dependent HBM access per descent step instead of up_base -> up_ptr -> ids -> records
        if (cur_b == kNoneQ) continue;
        const uint32_t urw = sq.rec_words + 4;
        const uint32_t *row = sq.uptile + (size_t)(cur_b + lvl - 1) * ((size_t)sq.up_stride * urw);
        for (uint32_t base = 0; base < sq.up_stride; base += 64) {
          const uint32_t slot = base + lane;
          const uint32_t *r = row + (size_t)(slot < sq.up_stride ? slot : 0) * urw;
          const uint4 h = *reinterpret_cast<const uint4 *>(r);
          const uint32_t nb_b = r[sq.rec_words];
          const bool act = slot < sq.up_stride && h.w != kNoneQ;
          const unsigned long long am = hs_ballot(act);
          if (!am) break;
          const float mine = act ? est_rec<NBLK>(sq, planes, gadd, delta, vl, k1, r, h) : FLT_MAX;
          n_est += __popcll(am);
          const float d = wave_min_f32(mine);
          const unsigned long long eq = hs_ballot(act && mine == d);
          if (eq && d < curd) {
            const int l = __ffsll((long long)eq) - 1;
            curd = d;
            cur = __builtin_amdgcn_readlane(h.w, l);
            cur_b = __builtin_amdgcn_readlane(nb_b, l);
            changed = true;
          }
        }
        continue;
      }
      const uint32_t b = uni(ix.up_base[cur]);
      if (b == kNoneQ) continue;
      const uint32_t s = uni(ix.up_ptr[b + lvl - 1]), e = uni(ix.up_ptr[b + lvl]);
      for (uint32_t base = s; base < e; base += 64) {
        const uint32_t m = min(64u, e - base);
        const bool act = (uint32_t)lane < m;
        const uint32_t c = act ? ix.cols[base + lane] : 0u;
        const float mine = act ? est_one<NBLK>(sq, planes, gadd, delta, vl, k1, c) : FLT_MAX;
        n_est += m;
        const float d = wave_min_f32(mine);
        const unsigned long long eq = hs_ballot(act && mine == d);
        if (eq && d < curd) {  // the sequential `if (d < curdist)` scan ends on the first index attaining the minimum
          curd = d;
          cur = __builtin_amdgcn_readlane(c, __ffsll((long long)eq) - 1);
          changed = true;
        }
      }
    }
  }
  if (a.phase == 1) {
    if (lane == 0) a.entry[qi] = make_uint4(cur, __float_as_uint(curd), n_est, 0u);
    return ST_TODO;
  }
  }

  float pkey[S];
  uint32_t pval[S];
#pragma unroll
  for (int s = 0; s < S; s++) { pkey[s] = INFINITY; pval[s] = 0xFFFFFFFFu; }
  PoolState<S> ps;
  ps.size = 0u; ps.cap = a.pool_cap; ps.cur = kNoRank; ps.last = INFINITY;
#pragma unroll
  for (int s = 0; s < S; s++) ps.inr[s] = hs_ballot((uint32_t)(lane * S + s) < a.pool_cap);
  pool_insert<S>(pkey, pval, ps, curd, cur, lane);
  uint32_t heap_n = 0, n_pend = 0, n_set = 0;
  int rc = ST_DONE;

  float hkey = 0.f;      // register heap (k < 64): element j in lane j
  uint32_t hid = 0u;
  float heap_root = 0.f;
  bool heap_dup = false;
  auto flush = [&]() {  // exact distances of the pending expansions, then the k-bounded heap, in expansion order
    wave_sync();
    const int sub = lane & 3, grp = lane >> 2;
    if (ix.dim & 15u) {  // the reference's SIMD4 / residual recipes, one lane per row
      if ((uint32_t)lane < n_pend) {
        const float *row = sq.raw + (size_t)pend[lane] * ix.dim;
        pd[lane] = METRIC == METRIC_L2 ? l2_general(qv, row, ix.dim) : ip_general(qv, row, ix.dim);
      }
    } else {
    const bool act = (uint32_t)grp < n_pend;
    const uint32_t id = pend[act ? grp : 0];
    const float4 *row = reinterpret_cast<const float4 *>(sq.raw + (size_t)id * ix.dim) + sub;
    const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t steps = ix.dim >> 4;
    // rounds of kRR 16-byte loads in flight per lane (one load per iteration would serialise the HBM latency)
    constexpr int kRR = RerankDepth<NBLK>::v;
    for (uint32_t r0 = 0; r0 < steps; r0 += kRR) {
      const uint32_t nb = min((uint32_t)kRR, steps - r0);
      float4 buf[kRR];
#pragma unroll
      for (uint32_t i = 0; i < kRR; i++)
        if (i < nb) buf[i] = row[(r0 + i) * 4];
#pragma unroll
      for (uint32_t i = 0; i < kRR; i++)
        if (i < nb) step4<METRIC>(acc, qq[(r0 + i) * 4], buf[i]);
    }
    bool owner;
    const float r = lane4_reduce<METRIC>(acc, sub, owner);
    if (act && owner) pd[grp] = r;
    }
    wave_sync();
    if (a.k < 64) {
      const RegHeap rh{hkey, hid, lane};
      uint32_t hn = heap_n;
      for (uint32_t j = 0; j < n_pend; j++) {
        const float dj = unif(pd[j]);
        // Full heap, no two equal keys in it, new key strictly above the root: push_heap lifts the new element to the
        // root along its leaf-to-root path and pop_heap's sift-down retraces exactly that path (every path element is
        // strictly larger than its sibling), leaving the array as it was -- skip the emulation.
        if (hn == a.k && !heap_dup && dj > heap_root) continue;
        heap_dup = heap_dup || hs_ballot((uint32_t)lane < hn && hkey == dj) != 0ull;
        rh[hn++] = Pair{dj, uni(pend[j])};
        push_heap(rh, (long)hn, LessD());
        if (hn > a.k) { pop_heap(rh, (long)hn, LessD()); hn--; }
        heap_root = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hkey), 0));
      }
    } else if (lane == 0) {
      uint32_t hn = heap_n;
      for (uint32_t j = 0; j < n_pend; j++) {
        heap[hn++] = Pair{pd[j], pend[j]};
        push_heap(heap, (long)hn, LessD());
        if (hn > a.k) { pop_heap(heap, (long)hn, LessD()); hn--; }
      }
    }
    heap_n = min(heap_n + n_pend, a.k);
    n_pend = 0;
    wave_sync();
  };

#ifdef HS_SLIMQ_DIAG   // diagnostic build (make slimqdiag): shader cycles per phase behind the stats block (tools/slimq_config.py DIAG_EF)
  uint32_t diag_ph[6] = {0, 0, 0, 0, 0, 0};   // pop + expanded-set | tile wait | estimates | pre-test (set lookups) | insertions | re-rank flush
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long diag_tp = __builtin_amdgcn_s_memtime();
#define HS_SQ_LAP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); diag_ph[i] += (uint32_t)(t_ - diag_tp); diag_tp = t_; }
#else
#define HS_SQ_LAP(i)
#endif
  while (ps.cur != kNoRank) {
    const uint32_t node = pool_pop<S>(pval, ps, lane);
    // expanded-node set, test and (below) add in ONE LDS round: the 64 lanes read the 64 slots of the probe sequence from the
    // node's home slot on; the id is present iff it shows before the first empty slot, and that slot is where linear probing
    // puts it (so the per-lane lookups of the pre-test below, which walk the same sequence, find it)
    bool seen = false;
    uint32_t free_slot = 0;
    for (uint32_t h0 = hash_of(node, mask);; h0 = (h0 + 64u) & mask) {
      const uint32_t sl = (h0 + (uint32_t)lane) & mask;
      const uint32_t v = tab[sl];
      const unsigned long long m_eq = hs_ballot(v == node), m_emp = hs_ballot(v == kNoneQ);
      if (m_emp) {
        const uint32_t e = (uint32_t)__ffsll((long long)m_emp) - 1;
        seen = (m_eq & ((1ull << e) - 1ull)) != 0ull;
        free_slot = (h0 + e) & mask;
        break;
      }
      if (m_eq) { seen = true; break; }
    }
    if (DBG && a.trace && lane == 0 && n_tr + 1 < a.trace_cap) {
      a.trace[(size_t)qi * a.trace_cap + n_tr] = node | (seen ? kChecked : 0u);
      a.trace[(size_t)qi * a.trace_cap + n_tr + 1] = ps.size;
    }
    n_tr += 2;
    if (seen) { n_rev++; continue; }                                 // :700-702
    if (__builtin_expect((n_set + 1) * 4 > a.hash_slots * 3, 0)) { rc = ST_OVERFLOW; break; }
    if (lane == 0) tab[free_slot] = node;                            // :704
    n_set++;
    wave_sync();
    HS_SQ_LAP(0);
    bool any = false;
    // the scan of one tile of <= 64 neighbours: estimates d (lanes with act), then the buffer updates in adjacency order
    auto scan = [&](bool act, uint32_t c, float d) {
      n_est += __popcll(hs_ballot(act));
      // is_full() can only turn true as the scan proceeds (the last key never grows once the buffer is full), so
      // the pre-test with the state at the start of the tile rejects nothing the sequential scan would accept
#ifdef HS_SLIMQ_DIAG
      asm volatile("s_nop 0" : : "v"(d) : "memory");
      HS_SQ_LAP(2);
#endif
      unsigned long long pendm = hs_ballot(act && !(d > ps.last) && !set_has(tab, mask, c));
      HS_SQ_LAP(3);
      while (pendm) {
        const int l = __ffsll((long long)pendm) - 1;
        pendm &= pendm - 1;
        const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
        const uint32_t cj = __builtin_amdgcn_readlane(c, l);
        if (dj > ps.last) continue;                       // is_full(), :741
        pool_insert<S>(pkey, pval, ps, dj, cj, lane);      // :745
        n_ins++;
        if (DBG && a.trace && lane == 0 && n_tr + 1 < a.trace_cap) {
          a.trace[(size_t)qi * a.trace_cap + n_tr] = cj | 0x40000000u;
          a.trace[(size_t)qi * a.trace_cap + n_tr + 1] = __float_as_uint(dj);
        }
        n_tr += 2;
      }
    };
    if (__builtin_expect(sq.ftile != nullptr, 1)) {
      // fused level-0 tile: slot j of node's row IS neighbour j's record, its header's 4th word the neighbour id
      // (0xFFFFFFFF = empty slot) -- one dependent HBM access per expansion instead of ids-then-records
      const uint32_t *row = sq.ftile + (size_t)node * ((size_t)ix.tile_stride * sq.rec_words);
      for (uint32_t base = 0; base < ix.tile_stride; base += 64) {
        const uint32_t slot = base + lane;
        const uint32_t *r = row + (size_t)(slot < ix.tile_stride ? slot : 0) * sq.rec_words;
        const uint4 h = *reinterpret_cast<const uint4 *>(r);
#ifdef HS_SLIMQ_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        HS_SQ_LAP(1);
#endif
        const bool act = slot < ix.tile_stride && h.w != kNoneQ;
        if (!hs_ballot(act)) break;
        any = true;
        const float d = act ? est_rec<NBLK>(sq, planes, gadd, delta, vl, k1, r, h) : FLT_MAX;
        scan(act, h.w, d);
      }
    } else {
      const uint32_t beg = uni(ix.row_ptr0[node]), end = uni(ix.row_ptr0[node + 1]);
      for (uint32_t base = beg; base < end; base += 64) {
        const bool act = base + lane < end;
        const uint32_t c = act ? ix.cols[base + lane] : kNoneQ;
        any = true;
        const float d = act ? est_one<NBLK>(sq, planes, gadd, delta, vl, k1, c) : FLT_MAX;
        scan(act, c, d);
      }
    }
    HS_SQ_LAP(4);
    if (!any) continue;   // neighbors == nullptr / size == 0: not reranked either (:708-715)
    n_hops++;
    if (lane == 0) pend[n_pend] = node;
    n_pend++;
    if (n_pend == 16) { flush(); HS_SQ_LAP(5); }
  }
  if (rc == ST_DONE) {
    if (n_pend) flush();
    for (uint32_t j = lane; j < a.k; j += 64) {
      const bool have = j < heap_n;
      const uint32_t hid_j = a.k < 64 ? hid : (have ? heap[j].id : 0u);
      const float hd_j = a.k < 64 ? hkey : (have ? heap[j].d : 0.f);
      a.out_labels[(size_t)qi * a.k + j] = have ? ix.labels[hid_j] : ~0ull;
      a.out_dists[(size_t)qi * a.k + j] = have ? hd_j : INFINITY;
    }
    if (lane == 0) {
      a.out_counts[qi] = heap_n;
      if (a.stats) {
        uint32_t *st = a.stats + (size_t)qi * 4; st[0] = n_hops; st[1] = n_est; st[2] = n_ins; st[3] = n_rev;
#ifdef HS_SLIMQ_DIAG
        uint32_t *dg = a.stats + (size_t)a.nq * 4 + (size_t)qi * 16;
        dg[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - diag_t0);   // 100 MHz ticks
        for (int i = 0; i < 6; i++) dg[8 + i] = diag_ph[i];
#endif
      }
    }
  } else if (lane == 0) {
    atomicAdd(&a.counters[0], 1u);
  }
  if (lane == 0) a.status[qi] = (uint32_t)rc;
  return rc;
}

template <int METRIC, int S, int NBLK, bool DBG = false>
__global__ void __launch_bounds__(64) slimq_kernel(DevIndex ix, DevSlimQ sq, SlimQArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.select_mask != (1u << ST_TODO)) {
    // second pass: 64 statuses per read, then the (normally zero) flagged queries of the block one after the other
    for (uint32_t base = blockIdx.x * 64; base < a.nq; base += gridDim.x * 64) {
      const uint32_t q = base + threadIdx.x;
      unsigned long long m = hs_ballot(q < a.nq && ((1u << a.status[q]) & a.select_mask));
      while (m) {
        const uint32_t qi = base + (uint32_t)__ffsll((long long)m) - 1;
        m &= m - 1;
        slimq_one<METRIC, S, NBLK, DBG>(ix, sq, a, qi, smem);
      }
    }
    return;
  }
  for (uint32_t it = blockIdx.x; it < a.nq; it += gridDim.x)   // first pass: every query
    slimq_one<METRIC, S, NBLK, DBG>(ix, sq, a, (a.phase == 2 && a.order) ? a.order[it] : it, smem);
}

template <typename K>
static hipError_t launch_k(K kern, const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, size_t lds, hipStream_t stream) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(64), lds, stream, ix, sq, a);
  return hipGetLastError();
}
template <int METRIC, int NBLK>
static hipError_t launch_ms(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, size_t lds, hipStream_t stream) {
  const uint32_t S = (a.pool_cap + 63) / 64;
  if (S <= 1) return launch_k(slimq_kernel<METRIC, 1, NBLK>, ix, sq, a, lds, stream);
  if (S <= 2) return launch_k(slimq_kernel<METRIC, 2, NBLK>, ix, sq, a, lds, stream);
  if (S <= 4) return launch_k(slimq_kernel<METRIC, 4, NBLK>, ix, sq, a, lds, stream);
  if (S <= 8) return launch_k(slimq_kernel<METRIC, 8, NBLK>, ix, sq, a, lds, stream);
  return launch_k(slimq_kernel<METRIC, 16, NBLK>, ix, sq, a, lds, stream);
}
hipError_t launch_slimq(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, hipStream_t stream) {
  const size_t lds = slimq_lds_bytes(ix.dim, sq.padded, sq.ncl, a.k, a.fb_tab ? 0u : a.hash_slots);
  if (a.trace)   // the parity/debug entry: one generic configuration that records the SearchBuffer events
    return ix.metric == METRIC_L2 ? launch_k(slimq_kernel<METRIC_L2, 16, 0, true>, ix, sq, a, lds, stream)
                                  : launch_k(slimq_kernel<METRIC_IP, 16, 0, true>, ix, sq, a, lds, stream);
  if (ix.metric == METRIC_L2) {
    if (sq.padded == 128) return launch_ms<METRIC_L2, 2>(ix, sq, a, lds, stream);
    if (sq.padded == 768) return launch_ms<METRIC_L2, 12>(ix, sq, a, lds, stream);
    return launch_ms<METRIC_L2, 0>(ix, sq, a, lds, stream);
  }
  if (sq.padded == 768) return launch_ms<METRIC_IP, 12>(ix, sq, a, lds, stream);   // COHERE / text embeddings
  return launch_ms<METRIC_IP, 0>(ix, sq, a, lds, stream);
}

}  // namespace hs
