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
// InnerProductSpace's dispatch for every dim (space_ip.h:374-382) on an AVX-512 host: dim%16==0 -> SIMD16ExtAVX512
// (ip_row16 above: FMA accumulators + halves tree); dim%4==0 -> SIMD4ExtAVX (:24-69: eight lane accumulators over the
// 16-element steps as multiply-then-add, folded lo+hi to four, 4-element steps added to those, TmpRes[0..3] summed left to
// right); dim>16 -> SIMD16 on the first dim/16*16 elements + scalar InnerProduct on the rest, 1-(res+tail) (:311-322);
// dim>4 -> SIMD4 + scalar rest (:324-337); else scalar (:6-19).  One accumulator chain at a time (the chains are
// independent, so the values are those of the vector code) -- small register footprint on the device.
HS_HD float ip_quad16(const float *q, const float *x, uint32_t d16, uint32_t j) {  // (a_j+a_{j+8}) + (a_{j+4}+a_{j+12})
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // lane accumulators j, j+4, j+8, j+12 of the SIMD16 loop (FMA)
#pragma unroll 1
  for (uint32_t s = 0; s < d16; s += 16) {
    v0 = __builtin_fmaf(q[s + j], x[s + j], v0);
    v1 = __builtin_fmaf(q[s + j + 4], x[s + j + 4], v1);
    v2 = __builtin_fmaf(q[s + j + 8], x[s + j + 8], v2);
    v3 = __builtin_fmaf(q[s + j + 12], x[s + j + 12], v3);
  }
  const float lo = v0 + v2;
  const float hi = v1 + v3;
  return lo + hi;
}
HS_HD float ip_part16(const float *q, const float *x, uint32_t d16) {  // _mm512_reduce_add_ps order
  float e = 0.f, o = 0.f;
#pragma unroll 1
  for (uint32_t j = 0; j < 2; j++) {
    const float a = ip_quad16(q, x, d16, j);
    const float b = ip_quad16(q, x, d16, j + 2);
    const float h = a + b;   // h4[j] + h4[j+2]
    if (j == 0) e = h; else o = h;
  }
  return e + o;
}
HS_HD float ip_part4(const float *q, const float *x, uint32_t d4) {  // InnerProductSIMD4ExtAVX, d4 % 4 == 0
  const uint32_t d16 = d4 & ~15u;
  float sum = 0.f;
#pragma unroll 1
  for (uint32_t j = 0; j < 4; j++) {
    float lo = 0.f, hi = 0.f;  // ymm lanes j and j+4
#pragma unroll 1
    for (uint32_t s = 0; s < d16; s += 16) {
      float p = q[s + j] * x[s + j];
      lo = lo + p;
      p = q[s + 8 + j] * x[s + 8 + j];
      lo = lo + p;
      p = q[s + 4 + j] * x[s + 4 + j];
      hi = hi + p;
      p = q[s + 12 + j] * x[s + 12 + j];
      hi = hi + p;
    }
    float a = lo + hi;
#pragma unroll 1
    for (uint32_t s = d16; s < d4; s += 4) {
      const float p = q[s + j] * x[s + j];
      a = a + p;
    }
    sum = j == 0 ? a : sum + a;   // ((sp0 + sp1) + sp2) + sp3
  }
  return sum;
}
HS_HD float ip_scalar(const float *q, const float *x, uint32_t d) {
  float r = 0.f;
  for (uint32_t i = 0; i < d; i++) {
    const float p = q[i] * x[i];
    r = r + p;
  }
  return r;
}
HS_HD float ip_general(const float *q, const float *x, uint32_t d) {  // the distance: 1 - <q,x>
  if ((d & 15u) == 0) return 1.0f - ip_part16(q, x, d);
  if ((d & 3u) == 0) return 1.0f - ip_part4(q, x, d);
  if (d > 16) {
    const uint32_t d16 = d & ~15u;
    const float r = ip_part16(q, x, d16);
    const float t = ip_scalar(q + d16, x + d16, d - d16);
    return 1.0f - (r + t);
  }
  if (d > 4) {
    const uint32_t d4 = d & ~3u;
    const float r = ip_part4(q, x, d4);
    const float t = ip_scalar(q + d4, x + d4, d - d4);
    return 1.0f - (r + t);
  }
  return 1.0f - ip_scalar(q, x, d);
}
inline float host_dist(Metric m, const float *q, const float *x, size_t d) {
  if (m == METRIC_L2) return (d & 15) ? l2_general(q, x, (uint32_t)d) : l2_row16(q, x, d);
  return (d & 15) ? ip_general(q, x, (uint32_t)d) : ip_row16(q, x, d);
}

}  // namespace hs
