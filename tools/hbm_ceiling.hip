// hbm_ceiling.hip -- what the MI355X memory system delivers to kernels that do nothing but move bytes, next to the 8 TB/s
// figure the roofline is priced against (SURVEY.md 8d asks for the confirmation on the box):
//   copy    : grid-stride float4 copy of 4 GiB (read + write)
//   read    : grid-stride float4 read of 4 GiB (sum)
//   gather4 : random 512-byte rows of an 8 GiB table, each read once, 4 lanes per row / eight 16-byte loads per lane (the
//             search kernels' wave_dists mapping), 16 rows in flight per wave
//   gather32: the same rows, 32 lanes per row (one instruction = one whole row), 16 rows in flight per wave
//   gatherN : row bytes 384 (d=96) and 3840 (d=960) with the 4- / 8-lane mappings the kernels use for them
// hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip -o /tmp/hbm_ceiling && /tmp/hbm_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

__global__ void copy_k(const float4 *a, float4 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_k(const float4 *a, size_t n, float *out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
// LPR lanes per row, ROWF4 float4 per row; each lane issues ROWF4 / LPR loads at a stride of LPR float4
template <int LPR, int ROWF4>
__global__ void gather_k(const float4 *tab, const uint32_t *ids, uint32_t nrows, float *out) {
  constexpr int RPP = 64 / LPR;   // rows per wave pass
  constexpr int NL = ROWF4 / LPR; // loads per lane
  const int lane = threadIdx.x & 63, sub = lane % LPR, grp = lane / LPR;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  for (uint32_t base = wave * RPP; base + RPP <= nrows; base += nwaves * RPP) {
    const float4 *row = tab + (size_t)ids[base + grp] * ROWF4 + sub;
    float4 b[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) b[i] = row[i * LPR];
#pragma unroll
    for (int i = 0; i < NL; i++) acc += b[i].x + b[i].y + b[i].z + b[i].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <typename F>
static void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int rep = 0; rep < 12; rep++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float t; hipEventElapsedTime(&t, e0, e1);
    if (rep >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-28s %8.1f GB/s median  %8.1f GB/s best   (%.3f ms median, %.2f GiB moved)\n", name, bytes / ms[ms.size() / 2] / 1e6, bytes / ms[0] / 1e6,
         ms[ms.size() / 2], bytes / (1 << 30));
  fflush(stdout);
}

int main() {
  const size_t TAB = 8ull << 30, MOVE = 4ull << 30;
  float4 *tab, *dst;
  float *out;
  uint32_t *ids;
  if (hipMalloc(&tab, TAB) != hipSuccess || hipMalloc(&dst, MOVE) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4);
  hipMemset(tab, 1, TAB);
  const int blocks = 256 * 16, threads = 256;
  timeit("copy 4 GiB (read+write)", 2.0 * MOVE, [&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(threads), 0, 0, tab, dst, MOVE / 16); });
  timeit("read 4 GiB", 1.0 * MOVE, [&] { hipLaunchKernelGGL(read_k, dim3(blocks), dim3(threads), 0, 0, tab, MOVE / 16, out); });
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpy(dst, tab, MOVE, hipMemcpyDeviceToDevice);
    hipEventRecord(e0); for (int i = 0; i < 5; i++) hipMemcpyAsync(dst, tab, MOVE, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipDeviceSynchronize();
    float t; hipEventElapsedTime(&t, e0, e1);
    printf("%-28s %8.1f GB/s (read+write)\n", "hipMemcpy D2D 4 GiB", 2.0 * MOVE * 5 / t / 1e6);
  }
  auto run_gather = [&](const char *name, int rowbytes, auto kern, int waves_per_cu_x) {
    const size_t nrows_tab = TAB / rowbytes;
    const uint32_t nread = (uint32_t)std::min<size_t>(MOVE / rowbytes, nrows_tab) / 64 * 64;
    std::vector<uint32_t> perm(nrows_tab);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    hipMalloc(&ids, (size_t)nread * 4);
    hipMemcpy(ids, perm.data(), (size_t)nread * 4, hipMemcpyHostToDevice);
    timeit(name, (double)nread * rowbytes, [&] { hipLaunchKernelGGL(kern, dim3(256 * waves_per_cu_x / 4), dim3(256), 0, 0, tab, ids, nread, out); });
    hipFree(ids);
  };
  run_gather("gather 512 B rows, 4 lanes/row", 512, gather_k<4, 32>, 32);
  run_gather("gather 512 B rows, 32 lanes/row", 512, gather_k<32, 32>, 32);
  run_gather("gather 512 B, 4 l/r, 16 waves/CU", 512, gather_k<4, 32>, 16);
  run_gather("gather 384 B rows, 4 lanes/row", 384, gather_k<4, 24>, 32);
  run_gather("gather 3840 B rows, 8 lanes/row", 3840, gather_k<8, 240>, 12);
  return 0;
}
