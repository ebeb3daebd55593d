// gather_calib.hip -- (1) calibrates rocprofv3's FETCH_SIZE for this kernel family's access pattern and
// (2) measures what the memory system delivers for random 512-byte row gathers under different lane mappings.
//   A: 4 lanes per row, 16 rows per wave pass, eight 16-byte loads per lane at a 64-byte stride (wave_dists)
//   B: 8 lanes per row (each instruction covers one 128-byte line of 8 rows), four loads per lane
//   C: 32 lanes per row (one instruction = one whole row, 2 rows per instruction), 8 rows in flight per wave
// Random rows of a 2 GiB table (>> 256 MiB Infinity Cache), each row read exactly once: known bytes = rows*512.
//   rocprofv3 --pmc FETCH_SIZE -- ./gather_calib A
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int MODE>
__global__ void gather(const float *tab, const uint32_t *ids, uint32_t nrows, float *out) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  for (uint32_t base = wave * 16; base < nrows; base += nwaves * 16) {
    float4 b[8];
    if (MODE == 0) {
      const int sub = lane & 3, grp = lane >> 2;
      const float4 *row = reinterpret_cast<const float4 *>(tab + (size_t)ids[base + grp] * 128) + sub;
#pragma unroll
      for (int i = 0; i < 8; i++) b[i] = row[i * 4];
    } else if (MODE == 1) {
      const int sub = lane & 7, grp = lane >> 3;  // 8 rows per half-pass, two half-passes
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const float4 *row = reinterpret_cast<const float4 *>(tab + (size_t)ids[base + h * 8 + grp] * 128) + sub;
#pragma unroll
        for (int i = 0; i < 4; i++) b[h * 4 + i] = row[i * 8];
      }
    } else {
      const int sub = lane & 31, grp = lane >> 5;  // 2 rows per instruction, 8 instructions
#pragma unroll
      for (int i = 0; i < 8; i++) b[i] = *(reinterpret_cast<const float4 *>(tab + (size_t)ids[base + i * 2 + grp] * 128) + sub);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc += b[i].x + b[i].y + b[i].z + b[i].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
  const char mode = argc > 1 ? argv[1][0] : 'A';
  const int blocks = argc > 2 ? atoi(argv[2]) : 4096, threads = argc > 3 ? atoi(argv[3]) : 256;
  const size_t N = 4u << 20;       // 4 Mi rows x 512 B = 2 GiB table
  const uint32_t rows = 2u << 20;  // read 2 Mi distinct random rows = 1 GiB
  float *tab, *out;
  uint32_t *ids;
  hipMalloc(&tab, N * 512);
  hipMalloc(&out, 4);
  hipMalloc(&ids, rows * 4);
  hipMemset(tab, 1, N * 512);
  std::vector<uint32_t> perm(N);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  hipMemcpy(ids, perm.data(), rows * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    if (mode == 'A') hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(threads), 0, 0, tab, ids, rows, out);
    else if (mode == 'B') hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(threads), 0, 0, tab, ids, rows, out);
    else hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(threads), 0, 0, tab, ids, rows, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gather_calib mode=%c grid=%dx%d rows=%u known_bytes=%zu  time=%.3f ms  %.1f GB/s\n", mode, blocks, threads, rows, (size_t)rows * 512, ms, rows * 512.0 / ms / 1e6);
  }
  return 0;
}
