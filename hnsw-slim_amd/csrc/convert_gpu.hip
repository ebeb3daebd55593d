// convert_gpu.hip -- HierarchicalNSWSlim::convertFromHNSW on gfx950 (hnswalg_slim.h:867-1108, PruneByHeuristic :836-865):
// the step immediately before the search path.  The list-level work -- distances, the by-distance
// std::sort, the pruning heuristic, the reverse-edge union, the re-prune of over-full lists -- runs on the device, one
// wavefront per (node, level) list; the degree histograms / hub thresholds (:904-945) and the final assembly of the element
// array and blobs (hierarchical filter :1063-1084, offsets, labels, vectors) stay on the host (host_graph.hpp), they are a
// few linear passes.  The output is the same bytes as the CPU harness SlimGraph::convert:
//   * distances by the same fp32 recipes as the search kernels (dist_recipe.hpp; 4 lanes per row for dim % 16 == 0, the
//     reference's SIMD4 / residual / scalar recipes one lane per row otherwise),
//   * std::sort by distance with libstdc++'s own tie behaviour: lists of <= 16 entries are an insertion sort (stable), done
//     as a rank sort by the whole wave; longer lists with equal keys run heap_emul.hpp's std_sort emulation on one lane,
//   * the reverse-edge lists are filled with atomics in any order and then sorted by id (as :1003-1011 does).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "convert_engine.hpp"
#include "dist_recipe.hpp"
#include "heap_emul.hpp"
#include "wave_util.hpp"

namespace hs {

static constexpr uint32_t kCvMaxList = 64;     // a source list (level-0 list of the vanilla graph) holds at most this many ids
static constexpr uint32_t kCvMaxKeep = 32;     // pruned lists hold at most this many ids (top_degree_M0, maxM0 <= 32 .. see host check)
static constexpr uint32_t kCvUnionCap = 2048;  // own list + reverse edges of one (node, level), in LDS

// distances query (LDS, dim floats) -> rows nid[0..cnt) (LDS) into nd[0..cnt) (LDS)
template <int METRIC>
__device__ __forceinline__ void cv_dists(const float *vec, uint32_t dim, const float *qv, const uint32_t *nid, float *nd, uint32_t cnt, int lane) {
  if ((dim & 15u) == 0) {
    const int sub = lane & 3, grp = lane >> 2;
    const uint32_t steps = dim >> 4;
    const float4 *qq = reinterpret_cast<const float4 *>(qv) + sub;
    for (uint32_t base = 0; base < cnt; base += 16) {
      const uint32_t j = base + grp;
      const bool act = j < cnt;
      const uint32_t id = nid[act ? j : base];
      const float4 *row = reinterpret_cast<const float4 *>(vec + (size_t)id * dim) + sub;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (uint32_t r0 = 0; r0 < steps; r0 += 8) {
        const uint32_t nb = min(8u, steps - r0);
        float4 buf[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
          if (i < nb) buf[i] = row[(r0 + i) * 4];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++)
          if (i < nb) step4<METRIC>(acc, qq[(r0 + i) * 4], buf[i]);
      }
      bool owner;
      const float r = lane4_reduce<METRIC>(acc, sub, owner);
      if (act && owner) nd[j] = r;
    }
  } else {
    for (uint32_t base = 0; base < cnt; base += 64) {
      const uint32_t j = base + lane;
      if (j < cnt) {
        const float *row = vec + (size_t)nid[j] * dim;
        nd[j] = METRIC == METRIC_L2 ? l2_general(qv, row, dim) : ip_general(qv, row, dim);
      }
    }
  }
}

// PruneByHeuristic (hnswalg_slim.h:836-865) over arr[0..sz) sorted ascending by distance: a candidate is kept unless a kept
// neighbour is closer to it than the node itself.  Sequential in the candidates, parallel over the kept set.
template <int METRIC>
__device__ __forceinline__ uint32_t cv_prune(const float *vec, uint32_t dim, const Pair *arr, uint32_t sz, uint32_t mlim, float *qc /*LDS dim*/,
                                             uint32_t *kept, float *kd, int lane) {
  uint32_t kc = 0;
  for (uint32_t t = 0; t < sz && kc < mlim; t++) {
    const float cd = unif(arr[t].d);
    const uint32_t cid = uni(arr[t].id);
    bool good = true;
    if (kc > 0) {
      for (uint32_t i = lane; i < dim; i += 64) qc[i] = vec[(size_t)cid * dim + i];
      wave_sync();
      cv_dists<METRIC>(vec, dim, qc, kept, kd, kc, lane);
      wave_sync();
      bool bad = false;
      for (uint32_t i = lane; i < kc; i += 64) bad = bad || kd[i] < cd;
      good = hs_ballot(bad) == 0;
    }
    if (good) {
      if (lane == 0) kept[kc] = cid;
      kc++;
    }
    wave_sync();
  }
  return kc;
}

// by-distance std::sort of (nd[j], nid[j]), j < sz, into arr[]: rank sort when libstdc++'s result is the stable order (sz <= 16:
// pure insertion sort) or no two keys are equal; otherwise the step-for-step emulation on one lane
__device__ __forceinline__ void cv_sort_by_dist(const uint32_t *nid, const float *nd, uint32_t sz, Pair *arr, int lane) {
  if (sz <= 64) {
    const bool act = (uint32_t)lane < sz;
    const float my_d = act ? nd[lane] : 0.f;
    const uint32_t my_id = act ? nid[lane] : 0u;
    uint32_t rank = 0;
    bool tie = false;
    for (uint32_t i = 0; i < sz; i++) {
      const float di = nd[i];
      rank += (di < my_d || (di == my_d && i < (uint32_t)lane)) ? 1u : 0u;
      tie = tie || (di == my_d && i != (uint32_t)lane);
    }
    const bool anytie = hs_ballot(act && tie) != 0;
    if (sz <= 16 || !anytie) {
      if (act) arr[rank] = Pair{my_d, my_id};
      wave_sync();
      return;
    }
  }
  for (uint32_t i = lane; i < sz; i += 64) arr[i] = Pair{nd[i], nid[i]};
  wave_sync();
  if (lane == 0) (void)std_sort(arr, (long)sz, LessD());
  wave_sync();
}

// ---- phase 1 (hnswalg_slim.h:951-986): every source list sorted by distance to its node and pruned to its degree budget ----
template <int METRIC>
__global__ void __launch_bounds__(64) cv_prune_kernel(const float *vec, uint32_t dim, const uint32_t *t_node, const uint32_t *t_off,
                                                      const uint32_t *t_size, const uint32_t *t_mlim, const uint32_t *lists, uint32_t ntasks,
                                                      uint32_t *out_nn, uint32_t *out_cnt) {
  extern __shared__ __align__(16) unsigned char smem[];
  float *qv = reinterpret_cast<float *>(smem);
  float *qc = qv + dim;
  uint32_t *nid = reinterpret_cast<uint32_t *>(qc + dim);
  float *nd = reinterpret_cast<float *>(nid + kCvMaxList);
  Pair *arr = reinterpret_cast<Pair *>(nd + kCvMaxList);
  uint32_t *kept = reinterpret_cast<uint32_t *>(arr + kCvMaxList);
  float *kd = reinterpret_cast<float *>(kept + kCvMaxKeep);
  const int lane = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < ntasks; t += gridDim.x) {
    const uint32_t v = t_node[t], sz = t_size[t], mlim = t_mlim[t];
    wave_sync();
    for (uint32_t i = lane; i < dim; i += 64) qv[i] = vec[(size_t)v * dim + i];
    if ((uint32_t)lane < sz) nid[lane] = lists[t_off[t] + lane];
    wave_sync();
    cv_dists<METRIC>(vec, dim, qv, nid, nd, sz, lane);
    wave_sync();
    cv_sort_by_dist(nid, nd, sz, arr, lane);
    const uint32_t kc = cv_prune<METRIC>(vec, dim, arr, sz, mlim, qc, kept, kd, lane);
    if ((uint32_t)lane < kc) out_nn[(size_t)t * kCvMaxKeep + lane] = kept[lane];
    if (lane == 0) out_cnt[t] = kc;
  }
}

// (node u, level l) -> task index: level 0 = u, level l >= 1 = n + upb[u] + l - 1
__device__ __forceinline__ uint32_t cv_task_of(uint32_t u, uint32_t l, uint32_t n, const uint32_t *upb) { return l == 0 ? u : n + upb[u] + l - 1; }

// ---- phase 2 (hnswalg_slim.h:988-998): reverse edges, counted then filled ---------------------------------------------------
__global__ void cv_rev_count_kernel(const uint32_t *nn, const uint32_t *cnt, const uint32_t *t_level, const uint32_t *upb, uint32_t n,
                                    uint32_t ntasks, uint32_t *rcnt) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = g / kCvMaxKeep, i = g % kCvMaxKeep;
  if (t >= ntasks || i >= cnt[t]) return;
  atomicAdd(&rcnt[cv_task_of(nn[(size_t)t * kCvMaxKeep + i], t_level[t], n, upb)], 1u);
}
__global__ void cv_rev_fill_kernel(const uint32_t *nn, const uint32_t *cnt, const uint32_t *t_node, const uint32_t *t_level, const uint32_t *upb,
                                   uint32_t n, uint32_t ntasks, const uint32_t *roff, uint32_t *rcur, uint32_t *rev) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t t = g / kCvMaxKeep, i = g % kCvMaxKeep;
  if (t >= ntasks || i >= cnt[t]) return;
  const uint32_t tu = cv_task_of(nn[(size_t)t * kCvMaxKeep + i], t_level[t], n, upb);
  rev[roff[tu] + atomicAdd(&rcur[tu], 1u)] = t_node[t];
}

// ---- phase 3 (hnswalg_slim.h:999-1012, 1038-1062): own list + reverse edges, sorted by id, duplicates removed; a list over
//      its level's capacity is sorted by distance and pruned again ---------------------------------------------------------------
template <int METRIC>
__global__ void __launch_bounds__(64) cv_union_kernel(const float *vec, uint32_t dim, const uint32_t *t_node, const uint32_t *t_limit,
                                                      const uint32_t *nn, const uint32_t *cnt, const uint32_t *roff, const uint32_t *rcnt,
                                                      const uint32_t *rev, uint32_t ntasks, uint32_t *fin, uint32_t *fin_cnt, uint32_t *flags) {
  extern __shared__ __align__(16) unsigned char smem[];
  float *qv = reinterpret_cast<float *>(smem);
  float *qc = qv + dim;
  uint32_t *ids = reinterpret_cast<uint32_t *>(qc + dim);
  float *nd = reinterpret_cast<float *>(ids + kCvUnionCap);
  Pair *arr = reinterpret_cast<Pair *>(nd + kCvUnionCap);
  uint32_t *kept = reinterpret_cast<uint32_t *>(arr + kCvUnionCap);
  float *kd = reinterpret_cast<float *>(kept + kCvMaxKeep);
  const int lane = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < ntasks; t += gridDim.x) {
    const uint32_t m1 = cnt[t], r = rcnt[t], total = m1 + r, limit = t_limit[t];
    wave_sync();
    if (total > kCvUnionCap) {   // does not fit the on-chip buffers: the host redoes the conversion on the CPU
      if (lane == 0) { atomicAdd(flags, 1u); fin_cnt[t] = 0; }
      continue;
    }
    uint32_t N = 64;
    while (N < total) N <<= 1;
    for (uint32_t i = lane; i < N; i += 64)
      ids[i] = i < m1 ? nn[(size_t)t * kCvMaxKeep + i] : (i < total ? rev[roff[t] + (i - m1)] : 0xFFFFFFFFu);
    wave_sync();
    // bitonic sort ascending (std::sort of plain ids: equal keys are equal values, any correct sort gives the same array)
    for (uint32_t k = 2; k <= N; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = lane; i < N; i += 64) {
          const uint32_t p = i ^ j;
          if (p > i) {
            const uint32_t a = ids[i], b = ids[p];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { ids[i] = b; ids[p] = a; }
          }
        }
        wave_sync();
      }
    // std::unique
    uint32_t m = 0;
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t i = base + lane;
      const uint32_t x = i < total ? ids[i] : 0u;
      const bool first = i < total && (i == 0 || ids[i - 1] != x);
      const unsigned long long fm = hs_ballot(first);
      wave_sync();
      if (first) ids[m + __popcll(fm & ((1ull << lane) - 1ull))] = x;   // m + prefix <= i: never overwrites an unread entry of a later chunk
      m += __popcll(fm);
      wave_sync();
    }
    if (m <= limit) {
      if (m <= kCvMaxKeep) {
        if ((uint32_t)lane < m) fin[(size_t)t * kCvMaxKeep + lane] = ids[lane];
        if (lane == 0) fin_cnt[t] = m;
      } else if (lane == 0) { atomicAdd(flags, 1u); fin_cnt[t] = 0; }   // capacity above 32 ids: host path
      continue;
    }
    const uint32_t v = t_node[t];
    for (uint32_t i = lane; i < dim; i += 64) qv[i] = vec[(size_t)v * dim + i];
    wave_sync();
    cv_dists<METRIC>(vec, dim, qv, ids, nd, m, lane);
    wave_sync();
    cv_sort_by_dist(ids, nd, m, arr, lane);
    const uint32_t kc = cv_prune<METRIC>(vec, dim, arr, m, min(limit, kCvMaxKeep), qc, kept, kd, lane);
    if ((uint32_t)lane < kc) fin[(size_t)t * kCvMaxKeep + lane] = kept[lane];
    if (lane == 0) fin_cnt[t] = kc;
  }
}

// ---- host driver ----------------------------------------------------------------------------------------------------------
#define CV_TRY(expr)                        \
  do {                                      \
    hipError_t _e = (expr);                 \
    if (_e != hipSuccess) { err = _e; goto done; } \
  } while (0)

template <typename T>
static hipError_t cv_upload(T **dp, const std::vector<T> &v) {
  hipError_t e = hipMalloc((void **)dp, std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess) return e;
  return v.empty() ? hipSuccess : hipMemcpy(*dp, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

hipError_t gpu_convert_lists(const ConvertInput &in, int device, std::vector<uint32_t> &fin, std::vector<uint32_t> &fin_cnt, bool &needs_host,
                             double *kernel_ms) {
  hipError_t err = hipSuccess;
  needs_host = false;
  const uint32_t nt = (uint32_t)in.t_node.size();
  float *d_vec = nullptr;
  uint32_t *d_node = nullptr, *d_off = nullptr, *d_size = nullptr, *d_mlim = nullptr, *d_limit = nullptr, *d_level = nullptr, *d_lists = nullptr,
           *d_upb = nullptr, *d_nn = nullptr, *d_cnt = nullptr, *d_rcnt = nullptr, *d_roff = nullptr, *d_rcur = nullptr, *d_rev = nullptr,
           *d_fin = nullptr, *d_fcnt = nullptr, *d_flags = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<uint32_t> rcnt(nt), roff(nt + 1, 0);
  uint32_t flags = 0;
  const size_t lds1 = (size_t)in.dim * 8 + kCvMaxList * 16 + kCvMaxKeep * 8;
  const size_t lds3 = (size_t)in.dim * 8 + (size_t)kCvUnionCap * 16 + kCvMaxKeep * 8;
  const uint32_t grid = 256 * 16;
  const uint32_t eg = (uint32_t)(((size_t)nt * kCvMaxKeep + 255) / 256);
  if (nt == 0) return hipSuccess;
  CV_TRY(hipSetDevice(device));
  CV_TRY(hipMalloc((void **)&d_vec, (size_t)in.n * in.dim * 4));
  CV_TRY(hipMemcpy(d_vec, in.vec, (size_t)in.n * in.dim * 4, hipMemcpyHostToDevice));
  CV_TRY(cv_upload(&d_node, in.t_node)); CV_TRY(cv_upload(&d_off, in.t_off)); CV_TRY(cv_upload(&d_size, in.t_size));
  CV_TRY(cv_upload(&d_mlim, in.t_mlim)); CV_TRY(cv_upload(&d_limit, in.t_limit)); CV_TRY(cv_upload(&d_level, in.t_level));
  CV_TRY(cv_upload(&d_lists, in.lists)); CV_TRY(cv_upload(&d_upb, in.upb));
  CV_TRY(hipMalloc((void **)&d_nn, (size_t)nt * kCvMaxKeep * 4)); CV_TRY(hipMalloc((void **)&d_cnt, (size_t)nt * 4));
  CV_TRY(hipMalloc((void **)&d_rcnt, (size_t)nt * 4)); CV_TRY(hipMalloc((void **)&d_roff, (size_t)(nt + 1) * 4)); CV_TRY(hipMalloc((void **)&d_rcur, (size_t)nt * 4));
  CV_TRY(hipMalloc((void **)&d_fin, (size_t)nt * kCvMaxKeep * 4)); CV_TRY(hipMalloc((void **)&d_fcnt, (size_t)nt * 4)); CV_TRY(hipMalloc((void **)&d_flags, 4));
  CV_TRY(hipMemset(d_rcnt, 0, (size_t)nt * 4)); CV_TRY(hipMemset(d_rcur, 0, (size_t)nt * 4)); CV_TRY(hipMemset(d_flags, 0, 4));
  CV_TRY(hipEventCreate(&e0)); CV_TRY(hipEventCreate(&e1));
  if (lds3 > 64 * 1024) {
    CV_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(in.metric == METRIC_L2 ? cv_union_kernel<METRIC_L2> : cv_union_kernel<METRIC_IP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
  }
  CV_TRY(hipEventRecord(e0, nullptr));
  if (in.metric == METRIC_L2) hipLaunchKernelGGL(cv_prune_kernel<METRIC_L2>, dim3(std::min(grid, nt)), dim3(64), lds1, nullptr, d_vec, in.dim, d_node, d_off, d_size, d_mlim, d_lists, nt, d_nn, d_cnt);
  else hipLaunchKernelGGL(cv_prune_kernel<METRIC_IP>, dim3(std::min(grid, nt)), dim3(64), lds1, nullptr, d_vec, in.dim, d_node, d_off, d_size, d_mlim, d_lists, nt, d_nn, d_cnt);
  CV_TRY(hipGetLastError());
  hipLaunchKernelGGL(cv_rev_count_kernel, dim3(eg), dim3(256), 0, nullptr, d_nn, d_cnt, d_level, d_upb, in.n, nt, d_rcnt);
  CV_TRY(hipGetLastError());
  CV_TRY(hipMemcpy(rcnt.data(), d_rcnt, (size_t)nt * 4, hipMemcpyDeviceToHost));
  for (uint32_t t = 0; t < nt; t++) roff[t + 1] = roff[t] + rcnt[t];
  CV_TRY(hipMemcpy(d_roff, roff.data(), (size_t)(nt + 1) * 4, hipMemcpyHostToDevice));
  CV_TRY(hipMalloc((void **)&d_rev, std::max<size_t>(roff[nt], 1) * 4));
  hipLaunchKernelGGL(cv_rev_fill_kernel, dim3(eg), dim3(256), 0, nullptr, d_nn, d_cnt, d_node, d_level, d_upb, in.n, nt, d_roff, d_rcur, d_rev);
  CV_TRY(hipGetLastError());
  if (in.metric == METRIC_L2) hipLaunchKernelGGL(cv_union_kernel<METRIC_L2>, dim3(std::min(grid, nt)), dim3(64), lds3, nullptr, d_vec, in.dim, d_node, d_limit, d_nn, d_cnt, d_roff, d_rcnt, d_rev, nt, d_fin, d_fcnt, d_flags);
  else hipLaunchKernelGGL(cv_union_kernel<METRIC_IP>, dim3(std::min(grid, nt)), dim3(64), lds3, nullptr, d_vec, in.dim, d_node, d_limit, d_nn, d_cnt, d_roff, d_rcnt, d_rev, nt, d_fin, d_fcnt, d_flags);
  CV_TRY(hipGetLastError());
  CV_TRY(hipEventRecord(e1, nullptr));
  CV_TRY(hipEventSynchronize(e1));
  if (kernel_ms) { float ms = 0.f; CV_TRY(hipEventElapsedTime(&ms, e0, e1)); *kernel_ms = ms; }
  CV_TRY(hipMemcpy(&flags, d_flags, 4, hipMemcpyDeviceToHost));
  needs_host = flags != 0;
  fin.resize((size_t)nt * kCvMaxKeep);
  fin_cnt.resize(nt);
  CV_TRY(hipMemcpy(fin.data(), d_fin, fin.size() * 4, hipMemcpyDeviceToHost));
  CV_TRY(hipMemcpy(fin_cnt.data(), d_fcnt, (size_t)nt * 4, hipMemcpyDeviceToHost));
done:
  for (void *p : {(void *)d_vec, (void *)d_node, (void *)d_off, (void *)d_size, (void *)d_mlim, (void *)d_limit, (void *)d_level, (void *)d_lists, (void *)d_upb,
                  (void *)d_nn, (void *)d_cnt, (void *)d_rcnt, (void *)d_roff, (void *)d_rcur, (void *)d_rev, (void *)d_fin, (void *)d_fcnt, (void *)d_flags})
    if (p) (void)hipFree(p);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return err;
}

}  // namespace hs
