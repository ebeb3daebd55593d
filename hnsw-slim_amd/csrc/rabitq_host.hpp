// rabitq_host.hpp -- host side of the HNSW-SlimQ (RaBitQ) path: rotation, query preparation, 1-bit data
// quantisation, the SlimQ index file, and the CPU harness that builds such a file.
//
// Restates (paths relative to /root/reference/third_party/):
//   FhtKacRotator::rotate                 rabitqlib/utils/rotator.hpp:370-423  (flip_sign :100-205, kacs_walk :299-368)
//   SplitSingleQuery ctor                 rabitqlib/index/query.hpp:112-156
//     quantize_scalar / rabitq_scalar_impl  rabitqlib/quantization/rabitq.hpp:322-337, rabitq_impl.hpp:534-581
//     ex_bits_code / faster_quantize_ex     rabitqlib/quantization/rabitq_impl.hpp:405-432, 379-403
//     new_transpose_bin                     rabitqlib/utils/space.hpp:1405-1516
//   one_bit_code_with_factor              rabitqlib/quantization/rabitq_impl.hpp:75-138 (+ pack_binary space.hpp:272-286)
//   HierarchicalNSWSlimQ::saveIndex/loadIndex   hnswlib/hnswalg_slimq.h:1161-1313 (element layout :1498-1505)
//
// The rotation is reproduced bit for bit (verified against the compiled rabitqlib: the fast Hadamard transform
// runs its butterfly stages in ascending stride).  The reference's float reductions go through Eigen, whose
// summation order depends on SIMD width and pointer alignment; here every reduction is a plain left-to-right
// fp32 sum, which is this implementation's DEFINITION (pinned to the compiled library within 1e-5 relative;
// integer codes are compared exactly).  Compile with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host_graph.hpp"

namespace hs {

inline size_t rq_padded(size_t dim) { return (dim + 63) / 64 * 64; }

struct Rotator {
  size_t dim = 0, padded = 0, trunc = 0;
  float fac = 0;
  std::vector<uint8_t> flip;  // 4 * padded / 8 bytes
  void init(size_t d) {
    dim = d;
    padded = rq_padded(d);
    size_t lg = 0;
    while ((size_t(2) << lg) <= d) lg++;
    trunc = size_t(1) << lg;  // 2^floor(log2(dim))   (rotator.hpp:231-233: log of the UNPADDED dim)
    fac = 1.0f / std::sqrt((float)trunc);
    flip.assign(4 * padded / 8, 0);
  }
  void random_flips(uint64_t seed) {
    std::mt19937 gen((uint32_t)seed);
    std::uniform_int_distribution<int> d(0, 255);
    for (auto &b : flip) b = (uint8_t)d(gen);
  }
  static void flip_sign(const uint8_t *f, float *x, size_t n) {
    for (size_t i = 0; i < n; i++)
      if ((f[i >> 3] >> (i & 7)) & 1) x[i] = -x[i];
  }
  static void fht(float *x, size_t n) {  // in-place Walsh-Hadamard, stride 1, 2, 4, ... (as FFHT's helper_float_N)
    for (size_t h = 1; h < n; h <<= 1)
      for (size_t i = 0; i < n; i += 2 * h)
        for (size_t j = i; j < i + h; j++) {
          const float a = x[j], b = x[j + h];
          x[j] = a + b;
          x[j + h] = a - b;
        }
  }
  static void kac(float *x, size_t n) {
    const size_t h = n / 2;
    for (size_t i = 0; i < h; i++) {
      const float a = x[i], b = x[i + h];
      x[i] = a + b;
      x[i + h] = a - b;
    }
  }
  void rotate(const float *in, float *out) const {
    memcpy(out, in, 4 * dim);
    std::fill(out + dim, out + padded, 0.f);
    const size_t nb = padded / 8;
    if (trunc == padded) {
      for (int r = 0; r < 4; r++) {
        flip_sign(flip.data() + r * nb, out, padded);
        fht(out, trunc);
        for (size_t i = 0; i < trunc; i++) out[i] *= fac;
      }
      return;
    }
    const size_t start = padded - trunc;
    for (int r = 0; r < 4; r++) {
      flip_sign(flip.data() + r * nb, out, padded);
      float *p = (r & 1) ? out + start : out;
      fht(p, trunc);
      for (size_t i = 0; i < trunc; i++) p[i] *= fac;
      kac(out, padded);
    }
    for (size_t i = 0; i < padded; i++) out[i] *= 0.25f;
  }
};

// plain left-to-right fp32 reductions (this implementation's definition, see header)
inline float rq_dot(const float *a, const float *b, size_t n) { float s = 0; for (size_t i = 0; i < n; i++) s += a[i] * b[i]; return s; }
inline float rq_l2sqr(const float *a, const float *b, size_t n) { float s = 0; for (size_t i = 0; i < n; i++) { float t = a[i] - b[i]; s += t * t; } return s; }

// 1-bit code + estimator factors of one (rotated) vector against its (rotated) centroid.
// code: dimension i -> bit 63-(i%64) of word i/64 (pack_binary).  fac = {f_add, f_rescale, f_error}.
inline void rq_quantize_data(const float *x, const float *cen, size_t padded, int metric, uint64_t *code, float *fac) {
  constexpr float kConstEpsilon = 1.9f;  // rabitqlib/defines.hpp
  float l2_sqr = 0, ip_resi = 0, ip_cent = 0, l2_xucb = 0, ip_resi_cent = 0;
  for (size_t w = 0; w < padded / 64; w++) code[w] = 0;
  for (size_t i = 0; i < padded; i++) {
    const float r = x[i] - cen[i];
    const int bit = r > 0 ? 1 : 0;
    const float xu = (float)bit + (-0.5f);
    if (bit) code[i >> 6] |= 1ull << (63 - (i & 63));
    l2_sqr += r * r;
    ip_resi += r * xu;
    ip_cent += cen[i] * xu;
    l2_xucb += xu * xu;
    ip_resi_cent += r * cen[i];
  }
  const float l2_norm = std::sqrt(l2_sqr);
  if (ip_resi == 0) ip_resi = std::numeric_limits<float>::infinity();
  const float tmp_error = l2_norm * kConstEpsilon * std::sqrt((((l2_sqr * l2_xucb) / (ip_resi * ip_resi)) - 1) / (float)(padded - 1));
  if (metric == METRIC_L2) {
    fac[0] = l2_sqr + 2 * l2_sqr * ip_cent / ip_resi;
    fac[1] = -2 * l2_sqr / ip_resi;
    fac[2] = 2 * tmp_error;
  } else {
    fac[0] = 1 - ip_resi_cent + l2_sqr * ip_cent / ip_resi;
    fac[1] = -l2_sqr / ip_resi;
    fac[2] = 1 * tmp_error;
  }
}

// Prepared query: everything the estimator needs besides the per-vector record.
struct RqQuery {
  float delta = 0, vl = 0, k1xsumq = 0;
  std::vector<uint64_t> bins;  // padded/64 blocks x 4 planes: bins[blk*4 + b] = plane of code bit b
};
inline void rq_prepare_query(const float *rq, size_t padded, double t_const, RqQuery &out) {
  float sumq = 0;
  for (size_t i = 0; i < padded; i++) sumq += rq[i];  // std::accumulate (query.hpp:131-132)
  out.k1xsumq = sumq * (-0.5f);
  // ex_bits_code: |normalised residual| -> 3-bit code, flipped for negative coordinates (rabitq_impl.hpp:405-432)
  float nrm2 = 0;
  for (size_t i = 0; i < padded; i++) nrm2 += rq[i] * rq[i];
  const float nrm = std::sqrt(nrm2);
  std::vector<uint16_t> code(padded);
  float dot_ru = 0, l2_u = 0;
  for (size_t i = 0; i < padded; i++) {
    const float o_abs = std::fabs(rq[i] / nrm);
    int c = (int)((t_const * (double)o_abs) + 1e-5);  // faster_quantize_ex (:387-390)
    if (c >= 8) c = 7;
    if (rq[i] < 0) c = (~c) & 7;                       // :423-429
    const int bit = rq[i] > 0 ? 1 : 0;                 // one_bit_code against the zero centroid
    code[i] = (uint16_t)(c + (bit << 3));
    const float u = (float)code[i] + (-7.5f);
    dot_ru += rq[i] * u;
    l2_u += u * u;
  }
  const float norm_data = std::sqrt(nrm2), norm_quan = std::sqrt(l2_u);
  const float cos_sim = dot_ru / (norm_data * norm_quan);
  out.delta = norm_data / norm_quan * cos_sim;  // RECONSTRUCTION (rabitq_impl.hpp:570-571)
  out.vl = out.delta * (-7.5f);
  out.bins.assign(padded / 64 * 4, 0);
  for (size_t i = 0; i < padded; i++)
    for (int b = 0; b < 4; b++)
      if ((code[i] >> b) & 1) out.bins[(i >> 6) * 4 + b] |= 1ull << (63 - (i & 63));
}

// faster_config's t_const (rabitqlib/quantization/rabitq.hpp:27-33 -> rabitq_impl.hpp:276-377): the mean optimal
// rescale factor of 100 random unit vectors.  The reference draws them from std::random_device, i.e. every process
// gets a slightly different constant (about +-1 % at dim 128); here the generator is seeded.
inline double rq_best_rescale(const double *o_abs, size_t dim, size_t ex_bits) {
  static const double tight[9] = {0, 0.15, 0.20, 0.52, 0.59, 0.71, 0.75, 0.77, 0.81};
  const double mx = *std::max_element(o_abs, o_abs + dim);
  const double t_end = (double)(((1 << ex_bits) - 1) + 10) / mx;
  const double t_start = t_end * (double)(float)tight[ex_bits];
  std::vector<int> cur(dim);
  double den = (double)dim * 0.25, num = 0;
  for (size_t i = 0; i < dim; i++) {
    cur[i] = (int)(t_start * o_abs[i] + 1e-5);
    den += cur[i] * cur[i] + cur[i];
    num += (cur[i] + 0.5) * o_abs[i];
  }
  using E = std::pair<double, size_t>;
  std::vector<E> heap;
  for (size_t i = 0; i < dim; i++) heap.emplace_back((double)(cur[i] + 1) / o_abs[i], i);
  std::make_heap(heap.begin(), heap.end(), std::greater<E>());
  double best = 0, t = 0;
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), std::greater<E>());
    const E e = heap.back();
    heap.pop_back();
    const int ob = ++cur[e.second];
    den += 2 * ob;
    num += o_abs[e.second];
    const double ip = num / std::sqrt(den);
    if (ip > best) { best = ip; t = e.first; }
    if (ob < (1 << ex_bits) - 1) {
      const double tn = (double)(ob + 1) / o_abs[e.second];
      if (tn < t_end) { heap.emplace_back(tn, e.second); std::push_heap(heap.begin(), heap.end(), std::greater<E>()); }
    }
  }
  return t;
}
inline double rq_default_tconst(size_t padded, uint64_t seed, size_t ex_bits = 3) {
  std::mt19937 gen((uint32_t)seed);
  std::normal_distribution<double> nd(0, 1);
  std::vector<double> v(padded);
  double sum = 0;
  for (int j = 0; j < 100; j++) {
    double n2 = 0;
    for (auto &x : v) { x = nd(gen); n2 += x * x; }
    const double inv = 1.0 / std::sqrt(n2);
    for (auto &x : v) x = std::fabs(x * inv);
    sum += rq_best_rescale(v.data(), padded, ex_bits);
  }
  return sum / 100;
}

// ------------------------------------------------------------------------------------------------
// HierarchicalNSWSlimQ file image.
struct SlimQGraph {
  size_t count = 0, dim = 0, padded = 0, num_cluster = 0, ex_bits = 3;
  size_t maxM = 0, maxM0 = 0, M = 0, efC = 0;
  int maxlevel = 0, threshold_level = 0;
  uint32_t enterpoint = 0;
  int metric = METRIC_L2;
  std::vector<float> centroids;  // num_cluster x padded (rotated)
  Rotator rot;
  std::vector<int32_t> level;
  std::vector<uint64_t> label;
  std::vector<uint32_t> cluster;
  std::vector<uint64_t> code;    // count x padded/64
  std::vector<float> factors;    // count x 3
  std::vector<std::vector<char>> blobs;  // CHAL blobs as in SlimGraph

  size_t size_bin() const { return padded / 8 + 12; }
  size_t size_ex() const { return padded * ex_bits / 8 + 8; }
  size_t size_per_el() const { return 28 + size_bin() + size_ex(); }
  uint32_t total(size_t i) const { return blobs[i].empty() ? 0u : (uint32_t)((blobs[i].size() - 2 * (size_t)level[i]) / 4); }

  void save(const std::string &path) const {  // hnswalg_slimq.h:1161-1216
    std::ofstream o(path, std::ios::binary);
    if (!o.is_open()) throw std::runtime_error("Cannot open file");
    put<uint64_t>(o, count); put<uint64_t>(o, size_per_el());
    put<uint64_t>(o, 8); put<uint64_t>(o, 4); put<uint64_t>(o, 28); put<uint64_t>(o, 16);
    put<int32_t>(o, maxlevel); put<int32_t>(o, threshold_level); put<uint32_t>(o, enterpoint);
    put<uint64_t>(o, maxM); put<uint64_t>(o, maxM0); put<uint64_t>(o, M); put<uint64_t>(o, efC);
    put<uint8_t>(o, 0);
    put<uint64_t>(o, num_cluster); put<uint64_t>(o, dim); put<uint64_t>(o, padded);
    put<uint64_t>(o, 24); put<uint64_t>(o, 28); put<uint64_t>(o, 28 + size_bin());
    put<uint64_t>(o, size_bin()); put<uint64_t>(o, size_ex()); put<uint64_t>(o, ex_bits);
    put<uint8_t>(o, (uint8_t)metric);
    o.write((const char *)centroids.data(), num_cluster * padded * 4);
    o.write((const char *)rot.flip.data(), rot.flip.size());
    std::vector<char> el(size_per_el(), 0);
    for (size_t i = 0; i < count; i++) {
      std::fill(el.begin(), el.end(), 0);
      const uint32_t tot = total(i);
      memcpy(el.data(), &level[i], 4); memcpy(el.data() + 4, &tot, 4); memcpy(el.data() + 8, &label[i], 8);
      memcpy(el.data() + 24, &cluster[i], 4);
      memcpy(el.data() + 28, &code[i * padded / 64], padded / 8);
      memcpy(el.data() + 28 + padded / 8, &factors[i * 3], 12);
      o.write(el.data(), el.size());  // ex-bits area stays zero: no function on the search path reads it
    }
    for (size_t i = 0; i < count; i++) {
      const uint32_t sz = 2 * (uint32_t)level[i] + 4 * total(i);
      put<uint32_t>(o, sz);
      if (sz && total(i) != 0) o.write(blobs[i].data(), sz);
    }
  }

  void load(const BinSource &src, int metric_expected, size_t d) {  // hnswalg_slimq.h:1218-1313
    BinReader r(src);
    count = r.pod<uint64_t>();
    const uint64_t spe = r.pod<uint64_t>();
    const uint64_t label_off = r.pod<uint64_t>(), off_total = r.pod<uint64_t>(), off_data = r.pod<uint64_t>(), off_nb = r.pod<uint64_t>();
    maxlevel = r.pod<int32_t>(); threshold_level = r.pod<int32_t>(); enterpoint = r.pod<uint32_t>();
    maxM = r.pod<uint64_t>(); maxM0 = r.pod<uint64_t>(); M = r.pod<uint64_t>(); efC = r.pod<uint64_t>();
    (void)r.pod<uint8_t>();
    num_cluster = r.pod<uint64_t>(); dim = r.pod<uint64_t>(); padded = r.pod<uint64_t>();
    const uint64_t off_cid = r.pod<uint64_t>(), off_bin = r.pod<uint64_t>(), off_ex = r.pod<uint64_t>();
    const uint64_t sbin = r.pod<uint64_t>(), sex = r.pod<uint64_t>();
    ex_bits = r.pod<uint64_t>();
    metric = r.pod<uint8_t>();
    if (label_off != 8 || off_total != 4 || off_nb != 16 || off_cid != 24 || off_bin != 28 || off_data != 28 || dim != d ||
        padded != rq_padded(d) || sbin != size_bin() || off_ex != 28 + sbin || spe != 28 + sbin + sex || metric != metric_expected)
      throw std::runtime_error("Index seems to be corrupted or unsupported");
    centroids.resize(num_cluster * padded);
    r.bytes(centroids.data(), centroids.size() * 4);
    rot.init(dim);
    r.bytes(rot.flip.data(), rot.flip.size());
    level.resize(count); label.resize(count); cluster.resize(count);
    code.resize(count * padded / 64); factors.resize(count * 3);
    std::vector<uint32_t> totals(count);
    std::vector<char> el(spe);
    for (size_t i = 0; i < count; i++) {
      r.bytes(el.data(), spe);
      memcpy(&level[i], el.data(), 4); memcpy(&totals[i], el.data() + 4, 4); memcpy(&label[i], el.data() + 8, 8);
      memcpy(&cluster[i], el.data() + 24, 4);
      memcpy(&code[i * padded / 64], el.data() + 28, padded / 8);
      memcpy(&factors[i * 3], el.data() + 28 + padded / 8, 12);
      if (cluster[i] >= num_cluster) throw std::runtime_error("Index seems to be corrupted or unsupported");
    }
    blobs.assign(count, {});
    for (size_t i = 0; i < count; i++) {
      const uint32_t sz = r.pod<uint32_t>();
      if (sz == 0 || totals[i] == 0) continue;
      if (sz != 2 * (uint32_t)level[i] + 4 * totals[i]) throw std::runtime_error("Index seems to be corrupted or unsupported");
      blobs[i].resize(sz);
      r.bytes(blobs[i].data(), sz);
    }
  }

  // CPU harness: take the graph of a HierarchicalNSWSlim file as it is and quantise its vectors.  (The reference
  // derives its SlimQ graph from rabitqlib's own HNSW, built with ESTIMATED distances -- hnsw_slimq_strategy.h:101-133,
  // hnswalg_slimq.h:1471-1790; the search path only needs A valid SlimQ file, and a graph built with exact distances
  // is the better graph.)  Every vector is rotated and 1-bit quantised against the rotated centroid of its cluster
  // (nearest raw centroid when cluster_ids is null).
  void from_slim(const SlimGraph &s, int metric_, const float *cent_raw, size_t ncl, const uint32_t *cluster_ids,
                 uint64_t flip_seed, int threads) {
    const size_t n = s.count, d = s.dim;
    count = n; dim = d; padded = rq_padded(d); metric = metric_; num_cluster = ncl;
    maxM = s.maxM; maxM0 = s.maxM0; M = s.M; efC = s.efC; maxlevel = s.maxlevel; threshold_level = s.threshold_level;
    enterpoint = s.enterpoint;
    rot.init(d);
    rot.random_flips(flip_seed);
    centroids.resize(ncl * padded);
    for (size_t c = 0; c < ncl; c++) rot.rotate(cent_raw + c * d, &centroids[c * padded]);
    level.resize(n); label.resize(n); cluster.resize(n); code.resize(n * padded / 64); factors.resize(n * 3);
    blobs = s.blobs;
    int T = std::max(1, threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++)
      pool.emplace_back([&, t]() {
        std::vector<float> rx(padded);
        for (size_t i = t; i < n; i += T) {
          level[i] = s.level(i);
          label[i] = s.label(i);
          uint32_t cid;
          if (cluster_ids) cid = cluster_ids[i];
          else {
            float best = std::numeric_limits<float>::max();
            cid = 0;
            for (size_t c = 0; c < ncl; c++) {
              float dd = rq_l2sqr(s.vec(i), cent_raw + c * d, d);
              if (dd < best) { best = dd; cid = (uint32_t)c; }
            }
          }
          cluster[i] = cid;
          rot.rotate(s.vec(i), rx.data());
          rq_quantize_data(rx.data(), &centroids[cid * padded], padded, metric, &code[i * padded / 64], &factors[i * 3]);
        }
      });
    for (auto &th : pool) th.join();
  }
};

}  // namespace hs
