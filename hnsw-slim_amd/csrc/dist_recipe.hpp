// dist_recipe.hpp -- the reference's fp32 distance arithmetic, operation for operation.
//
// The reference dispatches to AVX-512 kernels on an AVX-512 host: L2SqrSIMD16ExtAVX512
// (space_l2.h:25-54) keeps 16 lane accumulators, acc_j += (q_i-x_i)*(q_i-x_i) for i = j, j+16, ...
// as a rounded multiply then a rounded add (:45), and sums the lanes left to right (:49-51);
// InnerProductSIMD16ExtAVX512 (space_ip.h:146-199) uses FMA accumulators and the pairwise-halves
// tree of _mm512_reduce_add_ps.  Reproducing that order makes distances -- and therefore every heap
// decision -- bit-identical (verified against the compiled reference: tests/golden/dist_ref.npz).
//
// On the GPU four lanes cooperate on one row (lane `sub` owns accumulators 4*sub..4*sub+3 and loads
// the 16-byte chunk at 16*s + 4*sub of every 64-byte step s); the host versions below are the same
// recipe written per row and are used by the host index builder.  Compile with -ffp-contract=off.
#pragma once
#include "hd.hpp"

namespace hs {

enum Metric : int { METRIC_L2 = 0, METRIC_IP = 1 };

// One 4-lane partial step: acc[j] (+)= contribution of elements (x[j], q[j]), j<4.
HS_HD void l2_step4(float acc[4], const float q[4], const float x[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float t = q[j] - x[j];
    float p = t * t;
    acc[j] = acc[j] + p;
  }
}
HS_HD void ip_step4(float acc[4], const float q[4], const float x[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++) acc[j] = __builtin_fmaf(q[j], x[j], acc[j]);
}

// Host-side full-row recipes (dim % 16 == 0).
inline float l2_row16(const float *q, const float *x, size_t d) {
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  for (size_t s = 0; s < d; s += 16)
    for (int j = 0; j < 16; j++) {
      float t = q[s + j] - x[s + j];
      float p = t * t;
      acc[j] = acc[j] + p;
    }
  float r = acc[0];
  for (int j = 1; j < 16; j++) r = r + acc[j];
  return r;
}
inline float ip_row16(const float *q, const float *x, size_t d) {
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  for (size_t s = 0; s < d; s += 16)
    for (int j = 0; j < 16; j++) acc[j] = __builtin_fmaf(q[s + j], x[s + j], acc[j]);
  float h[8];
  for (int j = 0; j < 8; j++) h[j] = acc[j] + acc[j + 8];
  for (int j = 0; j < 4; j++) h[j] = h[j] + h[j + 4];
  for (int j = 0; j < 2; j++) h[j] = h[j] + h[j + 2];
  float ip = h[0] + h[1];
  return 1.0f - ip;
}
// L2Space's dispatch for every dim (space_l2.h:214-234): dim%16==0 -> SIMD16 (above); dim%4==0 -> SIMD4
// (:166-190: 4 lane accumulators, TmpRes[0..3] summed left to right); dim>16 -> SIMD16 on the first dim/16*16
// elements + scalar L2Sqr on the rest (:149-160); dim>4 -> SIMD4 + scalar rest (:192-205); else scalar (:6-20).
// One thread does a whole row: usable per lane on the device (runtime-dim kernels) and by the host builder.
HS_HD float l2_part16(const float *q, const float *x, uint32_t d16) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Device (rare, runtime-dim path): one lane accumulator at a time -- the same additions in the same order (acc_j over
  // the steps, then acc_0 + acc_1 + ... left to right) with two live registers instead of sixteen, so that this
  // fallback does not set the register budget (= resident waves) of the kernels it is compiled into.
  float r = 0.f;
  for (uint32_t j = 0; j < 16; j++) {
    float a = 0.f;
    for (uint32_t s = 0; s < d16; s += 16) {
      const float t = q[s + j] - x[s + j];
      const float p = t * t;
      a = a + p;
    }
    r = j == 0 ? a : r + a;
  }
  return r;
#else
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  for (uint32_t s = 0; s < d16; s += 16) {
    for (int j = 0; j < 16; j++) {
      const float t = q[s + j] - x[s + j];
      const float p = t * t;
      acc[j] = acc[j] + p;
    }
  }
  float r = acc[0];
  for (int j = 1; j < 16; j++) r = r + acc[j];
  return r;
#endif
}
HS_HD float l2_part4(const float *q, const float *x, uint32_t d4) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t s = 0; s < d4; s += 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float t = q[s + j] - x[s + j];
      const float p = t * t;
      acc[j] = acc[j] + p;
    }
  }
  return ((acc[0] + acc[1]) + acc[2]) + acc[3];
}
HS_HD float l2_scalar(const float *q, const float *x, uint32_t d) {
  float r = 0.f;
  for (uint32_t i = 0; i < d; i++) {
    const float t = q[i] - x[i];
    const float p = t * t;
    r = r + p;
  }
  return r;
}
HS_HD float l2_general(const float *q, const float *x, uint32_t d) {
  if ((d & 15u) == 0) return l2_part16(q, x, d);
  if ((d & 3u) == 0) return l2_part4(q, x, d);
  if (d > 16) {
    const uint32_t d16 = d & ~15u;
    const float r = l2_part16(q, x, d16);
    const float t = l2_scalar(q + d16, x + d16, d - d16);
    return r + t;
  }
  if (d > 4) {
    const uint32_t d4 = d & ~3u;
    const float r = l2_part4(q, x, d4);
    const float t = l2_scalar(q + d4, x + d4, d - d4);
    return r + t;
  }
  return l2_scalar(q, x, d);
}
inline float host_dist(Metric m, const float *q, const float *x, size_t d) {
  if (m == METRIC_L2) return (d & 15) ? l2_general(q, x, (uint32_t)d) : l2_row16(q, x, d);
  return ip_row16(q, x, d);   // callers guarantee dim % 16 == 0 for the inner product
}

}  // namespace hs
