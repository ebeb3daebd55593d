// rabitq_est.hpp -- the RaBitQ 1-bit distance estimator, shared by host and device code.
// Restates (paths relative to /root/reference/third_party/rabitqlib/):
//   warmup_ip_x0_q<4>        utils/warmup_space.hpp:8-102   (integer AND/popcount/shift/add: exact)
//   split_single_estdist     index/estimator.hpp:164-188    (fp32 expression, evaluated in the written order)
#pragma once
#include "hd.hpp"

namespace hs {

HS_HD int rq_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}

// ip_x0_qr = delta * sum_blk sum_j (popcnt(x & q_j) << j) + vl * sum_blk popcnt(x)
HS_HD float rq_ip_x0_qr(const uint64_t *code, const uint64_t *bins, uint32_t nblk, float delta, float vl) {
  uint32_t ip = 0, ppc = 0;
  for (uint32_t b = 0; b < nblk; b++) {
    const uint64_t x = code[b];
    ppc += rq_popc64(x);
#pragma unroll
    for (int j = 0; j < 4; j++) ip += (uint32_t)rq_popc64(x & bins[b * 4 + j]) << j;
  }
  const float a = delta * (float)ip;
  const float c = vl * (float)ppc;
  return a + c;
}
// est_dist = f_add + g_add + f_rescale * (ip_x0_qr + k1xsumq)
HS_HD float rq_est_dist(float f_add, float g_add, float f_rescale, float ip_x0_qr, float k1xsumq) {
  const float s = f_add + g_add;
  const float t = ip_x0_qr + k1xsumq;
  const float u = f_rescale * t;
  return s + u;
}

}  // namespace hs
