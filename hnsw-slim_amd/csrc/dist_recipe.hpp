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
inline float host_dist(Metric m, const float *q, const float *x, size_t d) {
  return m == METRIC_L2 ? l2_row16(q, x, d) : ip_row16(q, x, d);
}

}  // namespace hs
