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
// Mapping to the wavefront:
//   * Query preparation runs in LDS: the Hadamard butterflies are element-wise (bit-exact in any lane order);
//     the float reductions are plain left-to-right sums (the definition rabitq_host.hpp documents), one
//     reduction per lane, all of them in flight together.
//   * The SearchBuffer (sorted array, capacity ef) lives in registers: rank r in lane r%64, slot r/64.  Lower-bound
//     position = popcount of a ballot, the memmove = one DPP wave_shr per slot, pop = ffs of the unchecked ballot.
//   * One lane per neighbour: a RaBitQ record is 16 B of factors + padded/8 B of sign code (32 B at d=128), the
//     estimator is AND + popcount against the query's 4 bit planes (LDS broadcast reads).  Candidates that survive
//     the (monotone) is_full pre-test are inserted in adjacency order, as the reference's scan does.
//   * Exact reranks are batched 16 at a time (4 lanes per row, the distance recipe of dist_recipe.hpp) and pushed
//     into the k-heap in expansion order, so the heap array ends up as the reference's.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "heap_emul.hpp"
#include "rabitq_est.hpp"
#include "slimq_engine.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr uint32_t kNoneQ = 0xFFFFFFFFu;
static constexpr uint32_t kChecked = 0x80000000u;

__host__ __device__ inline uint32_t al16(uint32_t x) { return (x + 15u) & ~15u; }

struct SlimQLds { uint32_t off_q, off_y, off_u, off_planes, off_red, off_hash, off_heap, off_pend, off_pd, total; };
__host__ __device__ inline SlimQLds slimq_layout(uint32_t dim, uint32_t padded, uint32_t ncl, uint32_t k, uint32_t hash_slots) {
  SlimQLds l;
  uint32_t o = 0;
  l.off_q = o; o += al16(dim * 4);
  l.off_y = o; o += al16(padded * 4);
  l.off_u = o; o += al16(padded * 4);
  l.off_planes = o; o += al16(padded / 64 * 4 * 8);
  l.off_red = o; o += al16((ncl + 4) * 4);   // [0] sumq [1] |q|^2 [2] <q,u> [3] |u|^2 [4+c] g_add of cluster c
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
bool slimq_supported(uint32_t pool_cap) { return pool_cap >= 1 && pool_cap <= 512; }

// ---- expanded-node set: open addressing in LDS, inserts by one lane, lookups by all ---------------------------
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
__device__ __forceinline__ void set_add(uint32_t *tab, uint32_t mask, uint32_t id) {
  uint32_t h = hash_of(id, mask);
  while (tab[h] != kNoneQ) h = (h + 1) & mask;
  tab[h] = id;
}

// ---- SearchBuffer in registers --------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ float pool_key_at(const float (&key)[S], uint32_t r) {
  float v = 0.f;
#pragma unroll
  for (int s = 0; s < S; s++)
    if ((r >> 6) == (uint32_t)s) v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(key[s]), r & 63));
  return v;
}
// insert(): lower-bound position, shift the tail up by one, drop what falls beyond the capacity (:112-119)
template <int S>
__device__ __forceinline__ void pool_insert(float (&key)[S], uint32_t (&val)[S], uint32_t &size, uint32_t cap, float d,
                                            uint32_t id, int lane) {
  uint32_t pos = 0;
#pragma unroll
  for (int s = 0; s < S; s++) pos += __popcll(__ballot((uint32_t)(lane + 64 * s) < size && key[s] < d));
  uint32_t carry_k = 0, carry_v = 0;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const uint32_t kb = __float_as_uint(key[s]);
    const uint32_t last_k = __builtin_amdgcn_readlane(kb, 63), last_v = __builtin_amdgcn_readlane(val[s], 63);
    const uint32_t up_k = wave_shr1(carry_k, kb), up_v = wave_shr1(carry_v, val[s]);
    const uint32_t r = lane + 64 * s;
    key[s] = r > pos ? __uint_as_float(up_k) : (r == pos ? d : key[s]);
    val[s] = r > pos ? up_v : (r == pos ? id : val[s]);
    carry_k = last_k;
    carry_v = last_v;
  }
  size = min(size + 1, cap);
}
// pop(): closest unchecked entry (cur_ is always the first unchecked rank, :126-134); false when none is left
template <int S>
__device__ __forceinline__ bool pool_pop(uint32_t (&val)[S], uint32_t size, uint32_t &id, int lane) {
  bool found = false;
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned long long m = found ? 0ull : __ballot((uint32_t)(lane + 64 * s) < size && !(val[s] & kChecked));
    if (m) {
      const int l = __ffsll((long long)m) - 1;
      id = __builtin_amdgcn_readlane(val[s], l);
      if (lane == l) val[s] |= kChecked;
      found = true;
    }
  }
  return found;
}

// ---- estimator of one record (any lane, its own id) ----------------------------------------------------------
template <int NBLK>
__device__ __forceinline__ float est_one(const DevSlimQ &sq, const uint64_t *planes, const float *gadd, float delta, float vl,
                                         float k1, uint32_t id) {
  const uint32_t *r = sq.rec + (size_t)id * sq.rec_words;
  const uint4 h = *reinterpret_cast<const uint4 *>(r);
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
    ipq = rq_ip_x0_qr(code, planes, sq.padded >> 6, delta, vl);
  }
  return rq_est_dist(__uint_as_float(h.x), gadd[h.z], __uint_as_float(h.y), ipq, k1);
}

// ---- query preparation ---------------------------------------------------------------------------------------
__device__ __forceinline__ void flip_signs(float *y, const uint8_t *f, uint32_t n, int lane) {
  for (uint32_t i = lane; i < n; i += 64)
    if ((f[i >> 3] >> (i & 7)) & 1) y[i] = -y[i];
}
// in-place Walsh-Hadamard over n (power of two) floats in LDS, butterfly stages in ascending stride, then * scale
__device__ __forceinline__ void fht_lds(float *y, uint32_t n, float scale, int lane) {
  for (uint32_t h = 1; h < n; h <<= 1) {
    wave_sync();
    for (uint32_t t = lane; t < n / 2; t += 64) {
      const uint32_t j = ((t / h) * 2 * h) + (t % h);
      const float a = y[j], b = y[j + h];
      y[j] = a + b;
      y[j + h] = a - b;
    }
  }
  wave_sync();
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
    for (uint32_t i = lane; i < P / 2; i += 64) {  // kacs_walk
      const float a = y[i], b = y[i + P / 2];
      y[i] = a + b;
      y[i + P / 2] = a - b;
    }
  }
  wave_sync();
  for (uint32_t i = lane; i < P; i += 64) y[i] *= 0.25f;
  wave_sync();
}

template <int METRIC, int S, int NBLK>
__device__ int slimq_one(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, const uint32_t qi, unsigned char *smem) {
  const int lane = threadIdx.x;
  const SlimQLds L = slimq_layout(ix.dim, sq.padded, sq.ncl, a.k, a.hash_slots);
  float *qv = reinterpret_cast<float *>(smem + L.off_q);
  float *y = reinterpret_cast<float *>(smem + L.off_y);
  float *u = reinterpret_cast<float *>(smem + L.off_u);
  uint64_t *planes = reinterpret_cast<uint64_t *>(smem + L.off_planes);
  float *red = reinterpret_cast<float *>(smem + L.off_red);
  float *gadd = red + 4;
  uint32_t *tab = reinterpret_cast<uint32_t *>(smem + L.off_hash);
  Pair *heap = reinterpret_cast<Pair *>(smem + L.off_heap);
  uint32_t *pend = reinterpret_cast<uint32_t *>(smem + L.off_pend);
  float *pd = reinterpret_cast<float *>(smem + L.off_pd);
  const uint32_t P = sq.padded, mask = a.hash_slots - 1;

  wave_sync();
  for (uint32_t i = lane; i < ix.dim; i += 64) qv[i] = a.queries[(size_t)qi * ix.dim + i];
  for (uint32_t i = lane; i < P; i += 64) y[i] = i < ix.dim ? a.queries[(size_t)qi * ix.dim + i] : 0.f;
  for (uint32_t i = lane; i < a.hash_slots; i += 64) tab[i] = kNoneQ;
  rotate_lds(sq, y, lane);

  // reductions, one per lane: sum(q'), |q'|^2, and per centroid |q'-c|^2 (L2) or <q',c> (IP)
  for (uint32_t t = lane; t < 2 + sq.ncl; t += 64) {
    float s = 0.f;
    if (t == 0) {
      for (uint32_t i = 0; i < P; i++) s += y[i];
      red[0] = s;
    } else if (t == 1) {
      for (uint32_t i = 0; i < P; i++) s += y[i] * y[i];
      red[1] = s;
    } else {
      const float *ce = sq.cent + (size_t)(t - 2) * P;
      if (METRIC == METRIC_L2) {
        for (uint32_t i = 0; i < P; i++) { const float d = y[i] - ce[i]; s += d * d; }
        const float nrm = __fsqrt_rn(s);
        gadd[t - 2] = nrm * nrm;        // get_bin_est: g_add = norm * norm (hnswalg_slimq.h:436)
      } else {
        for (uint32_t i = 0; i < P; i++) s += y[i] * ce[i];
        gadd[t - 2] = -s;               // g_add = -<q', c>  (:423)
      }
    }
  }
  wave_sync();
  const float nrm = __fsqrt_rn(red[1]);
  const float k1 = red[0] * (-0.5f);
  // 4-bit scalar code of q' (1 sign bit + 3 magnitude bits), its bit planes, and u = code - 7.5
  for (uint32_t b = 0; b < P / 64; b++) {
    const float v = y[b * 64 + lane];
    const float oa = fabsf(__fdiv_rn(v, nrm));
    int c = (int)(sq.t_const * (double)oa + 1e-5);
    c = c >= 8 ? 7 : c;
    if (v < 0.f) c = (~c) & 7;
    c += v > 0.f ? 8 : 0;
    u[b * 64 + lane] = (float)c + (-7.5f);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned long long m = __ballot((c >> j) & 1);
      if (lane == 0) planes[b * 4 + j] = __brevll(m);   // dimension i -> bit 63 - i%64
    }
  }
  wave_sync();
  if (lane == 0) {
    float s = 0.f;
    for (uint32_t i = 0; i < P; i++) s += y[i] * u[i];
    red[2] = s;
  } else if (lane == 1) {
    float s = 0.f;
    for (uint32_t i = 0; i < P; i++) s += u[i] * u[i];
    red[3] = s;
  }
  wave_sync();
  const float nq = __fsqrt_rn(red[3]);
  const float cos_sim = __fdiv_rn(red[2], nrm * nq);
  const float delta = __fdiv_rn(nrm, nq) * cos_sim;
  const float vl = delta * (-7.5f);

  uint32_t n_hops = 0, n_est = 1, n_ins = 0, n_rev = 0;
  // entry point and greedy descent on estimated distances (:1850-1901)
  uint32_t cur = ix.enterpoint;
  float curd = unif(est_one<NBLK>(sq, planes, gadd, delta, vl, k1, cur));
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
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
        const unsigned long long eq = __ballot(act && mine == d);
        if (eq && d < curd) {  // the sequential `if (d < curdist)` scan ends on the first index attaining the minimum
          curd = d;
          cur = __builtin_amdgcn_readlane(c, __ffsll((long long)eq) - 1);
          changed = true;
        }
      }
    }
  }

  float pkey[S];
  uint32_t pval[S];
#pragma unroll
  for (int s = 0; s < S; s++) { pkey[s] = 0.f; pval[s] = 0u; }
  uint32_t psize = 0;
  pool_insert<S>(pkey, pval, psize, a.pool_cap, curd, cur, lane);
  uint32_t heap_n = 0, n_pend = 0, n_set = 0;
  int rc = ST_DONE;

  auto flush = [&]() {  // exact distances of the pending expansions, then the k-bounded heap, in expansion order
    wave_sync();
    const int sub = lane & 3, grp = lane >> 2;
    const bool act = (uint32_t)grp < n_pend;
    const uint32_t id = pend[act ? grp : 0];
    const float4 *row = reinterpret_cast<const float4 *>(sq.raw + (size_t)id * ix.dim) + sub;
    const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t steps = ix.dim >> 4;
#pragma unroll 4
    for (uint32_t s2 = 0; s2 < steps; s2++) step4<METRIC>(acc, qq[s2 * 4], row[s2 * 4]);
    bool owner;
    const float r = lane4_reduce<METRIC>(acc, sub, owner);
    if (act && owner) pd[grp] = r;
    wave_sync();
    if (lane == 0) {
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

  uint32_t node;
  while (pool_pop<S>(pval, psize, node, lane)) {
    node &= ~kChecked;
    if (set_has(tab, mask, node)) { n_rev++; continue; }            // :700-702
    if ((n_set + 1) * 4 > a.hash_slots * 3) { rc = ST_OVERFLOW; break; }
    wave_sync();
    if (lane == 0) set_add(tab, mask, node);                         // :704
    n_set++;
    wave_sync();
    uint32_t beg, end;
    if (ix.tile0) { beg = 0; end = ix.tile_stride; }
    else { beg = uni(ix.row_ptr0[node]); end = uni(ix.row_ptr0[node + 1]); }
    const uint32_t *adj = ix.tile0 ? ix.tile0 + (size_t)node * ix.tile_stride : ix.cols;
    bool any = false;
    for (uint32_t base = beg; base < end; base += 64) {
      const uint32_t c = base + lane < end ? adj[base + lane] : kNoneQ;
      const bool act = c != kNoneQ;
      const unsigned long long am = __ballot(act);
      if (!am) break;
      any = true;
      const float d = act ? est_one<NBLK>(sq, planes, gadd, delta, vl, k1, c) : FLT_MAX;
      n_est += __popcll(am);
      const bool full = psize == a.pool_cap;
      const float last = full ? pool_key_at<S>(pkey, a.pool_cap - 1) : FLT_MAX;
      // is_full() can only turn true as the scan proceeds (the last key never grows once the buffer is full), so
      // the pre-test with the state at the start of the tile rejects nothing the sequential scan would accept
      unsigned long long pendm = __ballot(act && !(full && d > last) && !set_has(tab, mask, c));
      while (pendm) {
        const int l = __ffsll((long long)pendm) - 1;
        pendm &= pendm - 1;
        const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
        const uint32_t cj = __builtin_amdgcn_readlane(c, l);
        if (psize == a.pool_cap && dj > pool_key_at<S>(pkey, a.pool_cap - 1)) continue;   // :741
        pool_insert<S>(pkey, pval, psize, a.pool_cap, dj, cj, lane);                       // :745
        n_ins++;
      }
    }
    if (!any) continue;   // neighbors == nullptr / size == 0: not reranked either (:708-715)
    n_hops++;
    if (lane == 0) pend[n_pend] = node;
    n_pend++;
    if (n_pend == 16) flush();
  }
  if (rc == ST_DONE) {
    if (n_pend) flush();
    for (uint32_t j = lane; j < a.k; j += 64) {
      const bool have = j < heap_n;
      a.out_labels[(size_t)qi * a.k + j] = have ? ix.labels[heap[j].id] : ~0ull;
      a.out_dists[(size_t)qi * a.k + j] = have ? heap[j].d : INFINITY;
    }
    if (lane == 0) {
      a.out_counts[qi] = heap_n;
      if (a.stats) { uint32_t *st = a.stats + (size_t)qi * 4; st[0] = n_hops; st[1] = n_est; st[2] = n_ins; st[3] = n_rev; }
    }
  } else if (lane == 0) {
    atomicAdd(&a.counters[0], 1u);
  }
  if (lane == 0) a.status[qi] = (uint32_t)rc;
  return rc;
}

template <int METRIC, int S, int NBLK>
__global__ void __launch_bounds__(64) slimq_kernel(DevIndex ix, DevSlimQ sq, SlimQArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    if (!((1u << a.status[qi]) & a.select_mask)) continue;
    slimq_one<METRIC, S, NBLK>(ix, sq, a, qi, smem);
  }
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
  return launch_k(slimq_kernel<METRIC, 8, NBLK>, ix, sq, a, lds, stream);
}
hipError_t launch_slimq(const DevIndex &ix, const DevSlimQ &sq, const SlimQArgs &a, hipStream_t stream) {
  const size_t lds = slimq_lds_bytes(ix.dim, sq.padded, sq.ncl, a.k, a.hash_slots);
  if (ix.metric == METRIC_L2) {
    if (sq.padded == 128) return launch_ms<METRIC_L2, 2>(ix, sq, a, lds, stream);
    return launch_ms<METRIC_L2, 0>(ix, sq, a, lds, stream);
  }
  return launch_ms<METRIC_IP, 0>(ix, sq, a, lds, stream);
}

}  // namespace hs
