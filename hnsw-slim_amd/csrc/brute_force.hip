// brute_force.hip -- exact k-NN by exhaustive scan on gfx950: hnswlib::BruteforceSearch::searchKnn
// (/root/reference/third_party/hnswlib/bruteforce.h:106-135) for a batch of queries.
//
// The reference scans the rows in order with a priority_queue of (dist, label) pairs under the default pair
// ordering and the test `dist <= lastdist`; the net effect is order independent: the k lexicographically smallest
// (dist, label) pairs.  Distances are the same fp32 recipes as everywhere else (dist_recipe.hpp: 16 lane
// accumulators, 4 lanes per row), so the result is the reference's bit for bit, ties included.
//
// Mapping: a workgroup of 4 waves owns a tile of QT queries (staged in LDS) and a chunk of rows.  A wave takes 16
// rows per pass (4 lanes per row); each 16-byte piece of a row is loaded once and used against all QT queries, so
// the scan is vector-ALU bound (3 operations per element and query: subtract, multiply, add -- the recipe forbids
// FMA for L2), not bandwidth bound.  Every wave keeps a sorted k-list per query in LDS; a candidate is looked at
// only if its distance does not exceed the list's last key, which after the first few hundred rows is rare.  A
// second kernel merges the per-(chunk, wave) lists of a query (sorted runs -> k rounds of a k-way merge).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "bf_engine.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr int kQT = 8;       // queries per workgroup tile
static constexpr int kWaves = 4;    // waves per workgroup

struct BfEntry { float d; uint32_t row; uint64_t label; };   // 16 bytes

__device__ __forceinline__ bool bf_less(float d, uint64_t l, const BfEntry &e) { return d < e.d || (d == e.d && l < e.label); }

// Candidates of one pass (mask m, value d in the owning lane, row = rb + (lane >> SHIFT)) against one query's sorted k-list.
template <int SHIFT>
__device__ __forceinline__ void bf_offer(unsigned long long m, float d, uint32_t rb, const uint64_t *labels, BfEntry *L, uint32_t *sz,
                                         uint32_t k, int lane) {
  while (m) {
    const int l = __ffsll((long long)m) - 1;
    m &= m - 1;
    const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
    const uint32_t rj = rb + (uint32_t)(l >> SHIFT);
    if (lane == 0) {
      const uint64_t lab = labels ? labels[rj] : (uint64_t)rj;
      uint32_t cur = *sz;
      if (cur < k || bf_less(dj, lab, L[k - 1])) {
        uint32_t pos = cur < k ? cur : k - 1;
        while (pos > 0 && bf_less(dj, lab, L[pos - 1])) { L[pos] = L[pos - 1]; pos--; }
        L[pos] = BfEntry{dj, rj, lab};
        if (cur < k) *sz = cur + 1;
      }
    }
    wave_sync();
  }
}

// Sorted runs of this (chunk, wave), padded with +inf, for the merge kernel.
__device__ __forceinline__ void bf_write_runs(const BfEntry *mine, const uint32_t *msz, BfEntry *partial, uint32_t q0, uint32_t nq, uint32_t k,
                                              int wave, int lane) {
  const uint32_t run = blockIdx.x * kWaves + wave, nruns = gridDim.x * kWaves;
  for (int t = 0; t < kQT; t++) {
    if (q0 + t >= nq) break;
    BfEntry *dst = partial + ((size_t)(q0 + t) * nruns + run) * k;
    for (uint32_t i = lane; i < k; i += 64) dst[i] = i < msz[t] ? mine[(size_t)t * k + i] : BfEntry{FLT_MAX, 0xFFFFFFFFu, ~0ull};
  }
}

template <int METRIC>
__global__ void __launch_bounds__(64 * kWaves) bf_scan_kernel(const float *base, const uint64_t *labels, uint32_t n, uint32_t dim,
                                                            const float *queries, uint32_t nq, uint32_t k, uint32_t rows_per_block,
                                                            BfEntry *partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *q = reinterpret_cast<float *>(smem);                                   // kQT x dim
  BfEntry *lists = reinterpret_cast<BfEntry *>(smem + (size_t)kQT * dim * 4);   // kWaves x kQT x k
  uint32_t *sizes = reinterpret_cast<uint32_t *>(lists + (size_t)kWaves * kQT * k);   // kWaves x kQT
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 3, grp = lane >> 2;
  const uint32_t q0 = blockIdx.y * kQT;
  for (uint32_t i = tid; i < kQT * dim; i += 64 * kWaves) {
    const uint32_t t = i / dim;
    q[i] = q0 + t < nq ? queries[(size_t)(q0 + t) * dim + (i - t * dim)] : 0.f;
  }
  if (tid < kWaves * kQT) sizes[tid] = 0;
  __syncthreads();
  BfEntry *mine = lists + (size_t)wave * kQT * k;
  uint32_t *msz = sizes + wave * kQT;
  const uint32_t r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  const uint32_t steps = dim >> 4;
  // (two rows per lane group and pass would halve the LDS reads of the queries, but costs 60 VGPRs and a resident
  // wave: measured 156 ms against 128 ms at 1M x 10k x 128)
  for (uint32_t rb = r0 + wave * 16; rb < r1; rb += 16 * kWaves) {
    const uint32_t row = rb + grp;
    const bool act = row < r1;
    const float4 *x = reinterpret_cast<const float4 *>(base + (size_t)(act ? row : r0) * dim) + sub;
    float acc[kQT][4];
#pragma unroll
    for (int t = 0; t < kQT; t++) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f;
    for (uint32_t s = 0; s < steps; s++) {
      const float4 xv = x[s * 4];
#pragma unroll
      for (int t = 0; t < kQT; t++) {
        const float4 qv = *reinterpret_cast<const float4 *>(q + (size_t)t * dim + s * 16 + sub * 4);
        step4<METRIC>(acc[t], qv, xv);
      }
    }
#pragma unroll
    for (int t = 0; t < kQT; t++) {
      bool owner;
      const float d = lane4_reduce<METRIC>(acc[t], sub, owner);
      if (q0 + t >= nq) continue;
      const uint32_t sz = msz[t];
      const float thr = sz < k ? FLT_MAX : mine[(size_t)t * k + k - 1].d;
      bf_offer<2>(hs_ballot(act && owner && d <= thr), d, rb, labels, mine + (size_t)t * k, msz + t, k, lane);   // bruteforce.h:120 `dist <= lastdist`
    }
  }
  wave_sync();
  bf_write_runs(mine, msz, partial, q0, nq, k, wave, lane);
}

// dim % 16 != 0: the reference's SIMD4 / residual / scalar recipes (dist_recipe.hpp l2_general / ip_general), one lane per
// row and 64 rows per wave and pass -- the shapes ground truth is occasionally needed for (GloVe d=100, 25, 50, ...), far
// from the tuned kernel above but the same values.
template <int METRIC>
__global__ void __launch_bounds__(64 * kWaves) bf_scan_general_kernel(const float *base, const uint64_t *labels, uint32_t n, uint32_t dim,
                                                                    const float *queries, uint32_t nq, uint32_t k,
                                                                    uint32_t rows_per_block, BfEntry *partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *q = reinterpret_cast<float *>(smem);                                                      // kQT x dim
  BfEntry *lists = reinterpret_cast<BfEntry *>(smem + (((size_t)kQT * dim * 4 + 15) & ~(size_t)15));   // kWaves x kQT x k
  uint32_t *sizes = reinterpret_cast<uint32_t *>(lists + (size_t)kWaves * kQT * k);               // kWaves x kQT
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t q0 = blockIdx.y * kQT;
  for (uint32_t i = tid; i < kQT * dim; i += 64 * kWaves) {
    const uint32_t t = i / dim;
    q[i] = q0 + t < nq ? queries[(size_t)(q0 + t) * dim + (i - t * dim)] : 0.f;
  }
  if (tid < kWaves * kQT) sizes[tid] = 0;
  __syncthreads();
  BfEntry *mine = lists + (size_t)wave * kQT * k;
  uint32_t *msz = sizes + wave * kQT;
  const uint32_t r0 = blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  for (uint32_t rb = r0 + wave * 64; rb < r1; rb += 64 * kWaves) {
    const uint32_t row = rb + lane;
    const bool act = row < r1;
    const float *x = base + (size_t)(act ? row : r0) * dim;
#pragma unroll 1
    for (int t = 0; t < kQT; t++) {
      if (q0 + t >= nq) break;
      const float d = METRIC == METRIC_L2 ? l2_general(q + (size_t)t * dim, x, dim) : ip_general(q + (size_t)t * dim, x, dim);
      const uint32_t sz = msz[t];
      const float thr = sz < k ? FLT_MAX : mine[(size_t)t * k + k - 1].d;
      bf_offer<0>(hs_ballot(act && d <= thr), d, rb, labels, mine + (size_t)t * k, msz + t, k, lane);
    }
  }
  wave_sync();
  bf_write_runs(mine, msz, partial, q0, nq, k, wave, lane);
}

// one wave per query: k rounds of a k-way merge over the sorted runs
__global__ void __launch_bounds__(64) bf_merge_kernel(const BfEntry *partial, uint32_t nq, uint32_t k, uint32_t nruns, uint64_t *out_labels,
                                                      float *out_dists, uint32_t *out_counts) {
  const uint32_t qi = blockIdx.x;
  const int lane = threadIdx.x;
  const BfEntry *runs = partial + (size_t)qi * nruns * k;
  // lane owns runs lane, lane+64, ...; head[j] = next unread entry of its j-th run (at most 16 runs per lane here)
  uint32_t head[16];
#pragma unroll
  for (int j = 0; j < 16; j++) head[j] = 0;
  uint32_t found = 0;
  for (uint32_t round = 0; round < k; round++) {
    float bd = FLT_MAX;
    uint64_t bl = ~0ull;
    int bj = -1;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t r = lane + 64 * j;
      if (r < nruns && head[j] < k) {
        const BfEntry e = runs[(size_t)r * k + head[j]];
        if (e.row != 0xFFFFFFFFu && (bj < 0 || e.d < bd || (e.d == bd && e.label < bl))) { bd = e.d; bl = e.label; bj = j; }
      }
    }
    const float md = wave_min_f32(bj >= 0 ? bd : FLT_MAX);
    unsigned long long m = hs_ballot(bj >= 0 && bd == md);
    if (!m) break;
    int win = __ffsll((long long)m) - 1;
    uint64_t wl = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(bl >> 32), win) << 32) | __builtin_amdgcn_readlane((uint32_t)bl, win);
    for (unsigned long long mm = m & (m - 1); mm; mm &= mm - 1) {
      const int l = __ffsll((long long)mm) - 1;
      const uint64_t ll = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(bl >> 32), l) << 32) | __builtin_amdgcn_readlane((uint32_t)bl, l);
      if (ll < wl) { wl = ll; win = l; }
    }
    if (lane == win) {
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (j == bj) head[j]++;
      out_labels[(size_t)qi * k + round] = bl;
      out_dists[(size_t)qi * k + round] = bd;
    }
    found++;
  }
  for (uint32_t i = found + lane; i < k; i += 64) { out_labels[(size_t)qi * k + i] = ~0ull; out_dists[(size_t)qi * k + i] = INFINITY; }
  if (lane == 0 && out_counts) out_counts[qi] = found;
}

size_t bf_partial_bytes(uint32_t n, uint32_t nq, uint32_t k, uint32_t *grid_x, uint32_t *rows_per_block) {
  // enough row chunks to fill the chip (256 CUs) for small query counts, at most 256 runs per query (16 per merge lane / 4 waves)
  const uint32_t tiles = (nq + kQT - 1) / kQT;
  uint32_t gx = (1024 + tiles - 1) / tiles;
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  uint32_t rpb = (n + gx - 1) / gx;
  rpb = (rpb + 63) / 64 * 64;
  if (rpb == 0) rpb = 64;
  gx = (n + rpb - 1) / rpb;
  if (gx == 0) gx = 1;
  *grid_x = gx;
  *rows_per_block = rpb;
  return (size_t)nq * gx * kWaves * k * sizeof(BfEntry);
}

hipError_t launch_brute_force(const float *base, const uint64_t *labels, uint32_t n, uint32_t dim, int metric, const float *queries,
                              uint32_t nq, uint32_t k, void *partial, uint32_t grid_x, uint32_t rows_per_block, uint64_t *out_labels,
                              float *out_dists, uint32_t *out_counts, hipStream_t stream) {
  const size_t lds = (((size_t)kQT * dim * 4 + 15) & ~(size_t)15) + (size_t)kWaves * kQT * k * sizeof(BfEntry) + kWaves * kQT * 4;
  if (lds > 160 * 1024) return hipErrorInvalidValue;   // dim above ~4500: the query tile no longer fits the CU's LDS
  const dim3 grid(grid_x, (nq + kQT - 1) / kQT);
  auto kern = (dim & 15u) ? (metric == METRIC_L2 ? bf_scan_general_kernel<METRIC_L2> : bf_scan_general_kernel<METRIC_IP>)
                          : (metric == METRIC_L2 ? bf_scan_kernel<METRIC_L2> : bf_scan_kernel<METRIC_IP>);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * kWaves), lds, stream, base, labels, n, dim, queries, nq, k, rows_per_block, (BfEntry *)partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bf_merge_kernel, dim3(nq), dim3(64), 0, stream, (const BfEntry *)partial, nq, k, grid_x * kWaves, out_labels, out_dists, out_counts);
  return hipGetLastError();
}

}  // namespace hs
