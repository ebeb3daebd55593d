// wave_util.hpp -- wavefront-level device helpers shared by beam_search.hip and slimq_search.hip (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "dist_recipe.hpp"

namespace hs {

// ---- wave helpers -------------------------------------------------------------------------------
// One wavefront per workgroup: LDS operations of a wave execute in issue order, so lanes see each other's
// LDS writes without s_barrier; all that is needed is that the COMPILER keeps the order.  (__syncthreads()
// would also drain vmcnt, i.e. stall on every global load still in flight -- the adjacency / row reads this
// kernel deliberately keeps outstanding while it works on the LDS heap.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
// whole-wave shift right by one lane (DPP wave_shr:1, a single VALU op on GFX9); lane 0 receives `carry`
__device__ __forceinline__ uint32_t wave_shr1(uint32_t carry, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}

// DPP move within a row of 16 lanes (row_shr / row_shl): one VALU op, no LDS round trip.  Lanes whose
// source falls outside the row keep their own value; callers only consume in-row results.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// Lane mask of a predicate.  HIP's __ballot(int) compares an int with zero, so a bool is first materialised in a VGPR
// (v_cndmask + v_cmp_ne per call); the builtin takes the comparison's SGPR pair as it stands.
__device__ __forceinline__ unsigned long long hs_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// v with lane `lane` (uniform) replaced by the uniform value x: one v_writelane instead of compare + move + select.
__device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t x, uint32_t lane) {
  asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(lane) : "m0");
  return v;
}

// Compare-and-swap on a word of LDS, addressed through a generic pointer (its low half is the LDS address).  Written as the DS
// instruction itself: an atomic on a generic pointer is expanded with is-shared / is-private tests of the address, and where this
// compiler only half-proves that the pointer is LDS it folds such a test into an illegal V_CMP against src_shared_base.
__device__ __forceinline__ uint32_t lds_cas(uint32_t *p, uint32_t expected, uint32_t desired) {
  const uint32_t addr = (uint32_t)(uintptr_t)p;
  uint32_t old;
  asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(expected), "v"(desired) : "memory");
  return old;
}

// Minimum of v over the wave without touching LDS: inclusive min-scan inside each row of 16 lanes with DPP
// row_shr, then the four row results (lanes 15/31/47/63) are combined through SGPRs.
__device__ __forceinline__ float wave_min_f32(float v) {
  const uint32_t inf = __float_as_uint(FLT_MAX);
#define HS_SHR_MIN(ctrl) v = fminf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)inf, (int)__float_as_uint(v), ctrl, 0xf, 0xf, false)))
  HS_SHR_MIN(0x111);  // row_shr:1
  HS_SHR_MIN(0x112);  // row_shr:2
  HS_SHR_MIN(0x114);  // row_shr:4
  HS_SHR_MIN(0x118);  // row_shr:8
#undef HS_SHR_MIN
  const uint32_t b = __float_as_uint(v);
  const float r0 = __uint_as_float(__builtin_amdgcn_readlane(b, 15)), r1 = __uint_as_float(__builtin_amdgcn_readlane(b, 31));
  const float r2 = __uint_as_float(__builtin_amdgcn_readlane(b, 47)), r3 = __uint_as_float(__builtin_amdgcn_readlane(b, 63));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}

template <int METRIC>
__device__ __forceinline__ float lane4_reduce(const float (&acc)[4], int sub, bool &owner) {
  if (METRIC == METRIC_L2) {
    // TmpRes[0] + TmpRes[1] + ... + TmpRes[15], left to right (space_l2.h:49-51)
    float r = ((acc[0] + acc[1]) + acc[2]) + acc[3];
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const float p = dpp_f<0x111>(r);  // row_shr:1 -- the running sum of the lane to the left (same 4-lane group)
      if (sub == k) r = (((p + acc[0]) + acc[1]) + acc[2]) + acc[3];
    }
    owner = sub == 3;
    return r;
  } else {
    // _mm512_reduce_add_ps: halves 16 -> 8 -> 4 -> 2 -> 1 (space_ip.h:197), then 1 - ip (:201-204)
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; i++) h[i] = acc[i] + dpp_f<0x102>(acc[i]);  // row_shl:2: lane+2 of the group
#pragma unroll
    for (int i = 0; i < 4; i++) h[i] = h[i] + dpp_f<0x101>(h[i]);      // row_shl:1: lane+1
    const float a0 = h[0] + h[2], a1 = h[1] + h[3];
    const float ip = a0 + a1;
    owner = sub == 0;
    return 1.0f - ip;
  }
}
// dim % 4 == 0, dim % 16 != 0 (GloVe-100, 200, 300 ...): the reference's 4-lane recipes with one lane per accumulator --
// lane `sub` of a 4-lane group owns TmpRes[sub] of L2SqrSIMD4Ext (space_l2.h:166-190: acc += (q-x)*(q-x) per 4-element
// step, rounded multiply then rounded add), or ymm lanes sub / sub+4 of InnerProductSIMD4ExtAVX (space_ip.h:24-69: even
// 4-element steps of the 16-blocks go to the low half, odd ones to the high half, lo+hi, then the remaining steps), and
// the group sums ((a0 + a1) + a2) + a3.  Dword loads, but the four lanes read 16 contiguous bytes and the sixteen groups
// walk sixteen rows in step, so the rows stream through the cache as in the float4 kernels.  q, x point at element `sub`.
// Every lane of the group returns the distance.  `hook()` runs after the first round of loads has been issued.
template <int METRIC, class Hook>
__device__ __forceinline__ float quad_dist4(const float *q, const float *x, uint32_t dim, Hook &&hook) {
  const uint32_t steps = dim >> 2;
  const uint32_t lim = METRIC == METRIC_L2 ? steps : (dim >> 4) << 2;   // IP: steps inside the 16-blocks
  float lo = 0.f, hi = 0.f;
  for (uint32_t s0 = 0; s0 < lim; s0 += 8) {
    const uint32_t nb = min(8u, lim - s0);
    float buf[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++)
      if (i < nb) buf[i] = x[(s0 + i) * 4];
    if (s0 == 0) hook();
#pragma unroll
    for (uint32_t i = 0; i < 8; i++)
      if (i < nb) {
        const float qe = q[(s0 + i) * 4];
        if (METRIC == METRIC_L2) {
          const float t = qe - buf[i];
          const float p = t * t;
          lo = lo + p;
        } else {
          const float p = qe * buf[i];
          if (i & 1) hi = hi + p; else lo = lo + p;   // s0 is a multiple of 8: parity of the step = parity of i
        }
      }
  }
  float a = lo;
  if (METRIC == METRIC_IP) {
    a = lo + hi;
    float buf[3];   // dim % 16 != 0 and dim % 4 == 0: one to three steps remain
#pragma unroll
    for (uint32_t i = 0; i < 3; i++)
      if (lim + i < steps) buf[i] = x[(lim + i) * 4];
    if (lim == 0) hook();
#pragma unroll
    for (uint32_t i = 0; i < 3; i++)
      if (lim + i < steps) {
        const float p = q[(lim + i) * 4] * buf[i];
        a = a + p;
      }
  }
  const float a0 = dpp_f<0x00>(a), a1 = dpp_f<0x55>(a), a2 = dpp_f<0xAA>(a), a3 = dpp_f<0xFF>(a);   // quad_perm broadcasts
  const float r = ((a0 + a1) + a2) + a3;
  return METRIC == METRIC_L2 ? r : 1.0f - r;
}

// One 16-byte chunk of the row against the query chunk: acc[j] (+)= contribution of elements j < 4, as two packed-f32 halves.
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 do two lanes of the recipe per issue slot with the same IEEE rounding per element
// (a rounded subtract, a rounded multiply, a rounded add for L2 -- never fused; one fused multiply-add for IP), so the values are
// those of dist_recipe.hpp's l2_step4 / ip_step4 bit for bit and the distance arithmetic costs half the vector-issue slots.
typedef float hs_f2 __attribute__((ext_vector_type(2)));
template <int METRIC>
__device__ __forceinline__ void step4(float (&acc)[4], const float4 &q4, const float4 &x4) {
  hs_f2 a0 = {acc[0], acc[1]}, a1 = {acc[2], acc[3]};
  const hs_f2 q0 = {q4.x, q4.y}, q1 = {q4.z, q4.w}, x0 = {x4.x, x4.y}, x1 = {x4.z, x4.w};
  if (METRIC == METRIC_L2) {
    const hs_f2 t0 = q0 - x0, t1 = q1 - x1;
    const hs_f2 p0 = t0 * t0, p1 = t1 * t1;
    a0 = a0 + p0;
    a1 = a1 + p1;
  } else {
    a0 = __builtin_elementwise_fma(q0, x0, a0);
    a1 = __builtin_elementwise_fma(q1, x1, a1);
  }
  acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
}


}  // namespace hs
