// oracle/hs_oracle_slimq.hpp -- TEST INFRASTRUCTURE ONLY (see hs_oracle.hpp).
//
// CPU restatement of the HNSW-SlimQ search path:
//   HierarchicalNSWSlimQ::searchKnn(q, k, result)     hnswlib/hnswalg_slimq.h:1810-1924
//     -> searchBaseLayerST<true>(.., k, qw, q2c)      hnswlib/hnswalg_slimq.h:688-759
//     -> get_bin_est                                  hnswlib/hnswalg_slimq.h:408-440
//     -> SearchBuffer                                 hnswlib/hnswalg_slimq.h:80-151
//   and the rabitqlib pieces under it (rotator, SplitSingleQuery, split_single_estdist), paths relative to
//   /root/reference/third_party/.
//
// Pinning status: the rabitqlib pieces are pinned against the compiled library (oracle/_ref/ref_rabitq ->
// tests/golden/rabitq_ref.npz): rotation, sign codes, query bit planes and the estimator bit for bit; float
// reductions (Eigen: order depends on alignment / SIMD width, i.e. not even stable across builds of the reference)
// are DEFINED here as left-to-right fp32 sums and agree with the library to 1e-4 of the value range.  The
// search loop itself (hnswalg_slimq.h) needs folly to compile: PARITY UNPINNED by a compiled reference, restated
// from source reading only.
#pragma once
#include "hs_oracle.hpp"

namespace hso {

struct SlimQIndex {
  size_t count = 0, dim = 0, padded = 0, ncl = 0, ef = 10;
  int maxlevel = 0, threshold_level = 0, metric = 0;
  uint32_t enterpoint = 0;
  std::vector<float> cent;       // ncl x padded
  std::vector<uint8_t> flips;    // 4 x padded/8
  std::vector<int32_t> lvl;
  std::vector<uint32_t> total, cid;
  std::vector<uint64_t> label, code;
  std::vector<float> fac;
  std::vector<std::vector<char>> nb;  // [level x u16 offsets][ids]
  double t_const = 0;
  const float *raw = nullptr;    // setDataset (hnswalg_slimq.h:303): rows indexed by INTERNAL id (:748)

  void load(const std::string &path) {  // hnswalg_slimq.h:1218-1313
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) throw std::runtime_error("Cannot open file");
    auto rd = [&](auto &v) { in.read((char *)&v, sizeof(v)); if (!in) throw std::runtime_error("truncated index file"); };
    uint64_t n, spe, loff, toff, doff, noff, maxM, maxM0, M, efc, offc, offb, offe, sbin, sex, exb, ncl64, d64, p64;
    uint8_t hasdel, mt;
    int32_t ml, tl; uint32_t ep;
    rd(n); rd(spe); rd(loff); rd(toff); rd(doff); rd(noff); rd(ml); rd(tl); rd(ep); rd(maxM); rd(maxM0); rd(M); rd(efc);
    rd(hasdel); rd(ncl64); rd(d64); rd(p64); rd(offc); rd(offb); rd(offe); rd(sbin); rd(sex); rd(exb); rd(mt);
    count = n; maxlevel = ml; threshold_level = tl; enterpoint = ep; ncl = ncl64; dim = d64; padded = p64; metric = mt;
    if (padded % 64 || padded < dim || sbin != padded / 8 + 12) throw std::runtime_error("unsupported SlimQ layout");
    cent.resize(ncl * padded);
    in.read((char *)cent.data(), cent.size() * 4);
    flips.resize(4 * padded / 8);
    in.read((char *)flips.data(), flips.size());
    lvl.resize(n); total.resize(n); cid.resize(n); label.resize(n); code.resize(n * padded / 64); fac.resize(n * 3);
    std::vector<char> el(spe);
    for (size_t i = 0; i < n; i++) {
      in.read(el.data(), spe);
      if (!in) throw std::runtime_error("truncated index file");
      memcpy(&lvl[i], &el[0], 4); memcpy(&total[i], &el[toff], 4); memcpy(&label[i], &el[loff], 8);
      memcpy(&cid[i], &el[offc], 4); memcpy(&code[i * padded / 64], &el[offb], padded / 8);
      memcpy(&fac[i * 3], &el[offb + padded / 8], 12);
    }
    nb.assign(n, {});
    for (size_t i = 0; i < n; i++) {
      uint32_t sz; rd(sz);
      if (sz == 0 || total[i] == 0) continue;
      nb[i].resize(sz);
      in.read(nb[i].data(), sz);
      if (!in) throw std::runtime_error("truncated index file");
    }
  }

  // slice of `level` in node i's blob (:1867-1884, :711-717)
  std::pair<const uint32_t *, size_t> slice(uint32_t i, int level) const {
    if (nb[i].empty()) return {nullptr, 0};
    const uint16_t *off = (const uint16_t *)nb[i].data();
    const int el = lvl[i];
    const uint32_t *ids = (const uint32_t *)(nb[i].data() + 2 * el);
    size_t b = level == 0 ? 0 : off[level - 1];
    size_t e = level == el ? total[i] : off[level];
    return {ids + b, e - b};
  }

  // FhtKacRotator::rotate (rabitqlib/utils/rotator.hpp:370-423)
  void rotate(const float *x, float *y) const {
    size_t lg = 0;
    while ((size_t(1) << (lg + 1)) <= dim) lg++;
    const size_t t = size_t(1) << lg;
    const float s = 1.0f / std::sqrt((float)t);
    for (size_t i = 0; i < padded; i++) y[i] = i < dim ? x[i] : 0.f;
    auto flip = [&](int round) { const uint8_t *f = &flips[round * padded / 8]; for (size_t i = 0; i < padded; i++) if (f[i / 8] & (1u << (i % 8))) y[i] = -y[i]; };
    auto hadamard = [&](float *p) {
      for (size_t len = 2; len <= t; len *= 2)
        for (size_t base = 0; base < t; base += len)
          for (size_t k = 0; k < len / 2; k++) { float u = p[base + k], v = p[base + k + len / 2]; p[base + k] = u + v; p[base + k + len / 2] = u - v; }
      for (size_t i = 0; i < t; i++) p[i] *= s;
    };
    if (t == padded) { for (int r = 0; r < 4; r++) { flip(r); hadamard(y); } return; }
    for (int r = 0; r < 4; r++) {
      flip(r);
      hadamard(r % 2 ? y + (padded - t) : y);
      for (size_t i = 0; i < padded / 2; i++) { float u = y[i], v = y[i + padded / 2]; y[i] = u + v; y[i + padded / 2] = u - v; }
    }
    for (size_t i = 0; i < padded; i++) y[i] *= 0.25f;
  }

  struct Query { float delta, vl, k1xsumq; std::vector<uint64_t> planes; std::vector<float> q2c; };
  // Sensitivity probe (tools/slimq_sensitivity.py): move {delta, vl, k1xsumq, every q_to_centroids entry} by this many units in
  // the last place after they have been computed -- how far may a different summation order (Eigen's, in the reference) move the ids?
  int perturb_ulps[4] = {0, 0, 0, 0};
  static float nudge(float v, int ulps) {
    for (int i = 0; i < (ulps < 0 ? -ulps : ulps); i++) v = std::nextafter(v, ulps > 0 ? INFINITY : -INFINITY);
    return v;
  }

  // SplitSingleQuery ctor (rabitqlib/index/query.hpp:112-156) + centroid table (hnswalg_slimq.h:1822-1848)
  void prepare(const float *rq, Query &Q) const {
    float sum = 0, n2 = 0;
    for (size_t i = 0; i < padded; i++) sum += rq[i];
    for (size_t i = 0; i < padded; i++) n2 += rq[i] * rq[i];
    const float nrm = std::sqrt(n2);
    Q.k1xsumq = -0.5f * sum;
    Q.planes.assign(padded / 64 * 4, 0);
    float ru = 0, uu = 0;
    for (size_t i = 0; i < padded; i++) {
      int c = (int)(t_const * (double)std::fabs(rq[i] / nrm) + 1e-5);
      c = std::min(c, 7);
      if (rq[i] < 0) c = 7 - c;
      if (rq[i] > 0) c += 8;
      const float u = (float)c - 7.5f;
      ru += rq[i] * u;
      uu += u * u;
      for (int b = 0; b < 4; b++) if (c & (1 << b)) Q.planes[i / 64 * 4 + b] |= uint64_t(1) << (63 - i % 64);
    }
    const float nq = std::sqrt(uu);
    Q.delta = nrm / nq * (ru / (nrm * nq));
    Q.vl = Q.delta * -7.5f;
    Q.q2c.assign(metric == METRIC_IP ? 2 * ncl : ncl, 0.f);
    for (size_t c = 0; c < ncl; c++) {
      const float *ce = &cent[c * padded];
      float l2 = 0, ip = 0;
      for (size_t i = 0; i < padded; i++) { float t = rq[i] - ce[i]; l2 += t * t; }
      for (size_t i = 0; i < padded; i++) ip += rq[i] * ce[i];
      if (metric == METRIC_IP) { Q.q2c[c] = ip; Q.q2c[c + ncl] = std::sqrt(l2); }
      else Q.q2c[c] = std::sqrt(l2);
    }
    if (perturb_ulps[0] | perturb_ulps[1] | perturb_ulps[2] | perturb_ulps[3]) {
      Q.delta = nudge(Q.delta, perturb_ulps[0]);
      Q.vl = nudge(Q.vl, perturb_ulps[1]);
      Q.k1xsumq = nudge(Q.k1xsumq, perturb_ulps[2]);
      for (auto &v : Q.q2c) v = nudge(v, perturb_ulps[3]);
    }
  }

  // get_bin_est (hnswalg_slimq.h:408-440) -> split_single_estdist (rabitqlib/index/estimator.hpp:164-188)
  float est(const Query &Q, uint32_t id) const {
    const float norm = Q.q2c[cid[id]];
    return est_g(Q, id, metric == METRIC_IP ? -norm : norm * norm);
  }
  float est_g(const Query &Q, uint32_t id, float g_add) const {
    const uint64_t *x = &code[id * padded / 64];
    uint32_t ip = 0, pc = 0;
    for (size_t b = 0; b < padded / 64; b++) {
      pc += __builtin_popcountll(x[b]);
      for (int j = 0; j < 4; j++) ip += (uint32_t)__builtin_popcountll(x[b] & Q.planes[b * 4 + j]) << j;
    }
    const float ipq = Q.delta * (float)ip + Q.vl * (float)pc;
    return fac[id * 3] + g_add + fac[id * 3 + 1] * (ipq + Q.k1xsumq);
  }
};

struct SlimQCounters { uint64_t n_hops = 0, n_est = 0, n_insert = 0, n_revisit = 0; };

// SearchBuffer (hnswalg_slimq.h:80-151), literal semantics
struct Pool {
  std::vector<std::pair<float, uint32_t>> d;
  size_t size = 0, cur = 0, cap;
  explicit Pool(size_t c) : d(c + 1), cap(c) {}
  size_t lower(float dist) const {
    size_t lo = 0, len = size;
    while (len > 1) { size_t half = len >> 1; len -= half; lo += (d[lo + half - 1].first < dist) * half; }
    return (lo < size && d[lo].first < dist) ? lo + 1 : lo;
  }
  void insert(uint32_t id, float dist) {
    size_t lo = lower(dist);
    for (size_t i = size; i > lo; i--) d[i] = d[i - 1];  // memmove(&data_[lo + 1], &data_[lo], ..)
    d[lo] = {dist, id};
    size += size < cap;
    cur = lo < cur ? lo : cur;
  }
  bool is_full(float dist) const { return size == cap && dist > d[size - 1].first; }
  uint32_t pop() {
    uint32_t id = d[cur].second;
    d[cur].second |= 1u << 31;
    ++cur;
    while (cur < size && (d[cur].second >> 31)) ++cur;
    return id;
  }
  bool has_next() const { return cur < size; }
};

// searchKnn(q, k, result) (hnswalg_slimq.h:1810-1924).  Output = the k-bounded max-heap ARRAY of (exact distance,
// internal id) in libstdc++ heap order (:1921-1923 reads top_candidates[i] for i < k); `found` = its size.
inline size_t slimq_search(const SlimQIndex &ix, const float *q, size_t k, std::vector<std::pair<float, uint32_t>> &heap,
                           SlimQCounters *ctr = nullptr, std::vector<uint32_t> *trace = nullptr) {
  heap.clear();
  if (ix.count == 0) return 0;
  std::vector<float> rq(ix.padded);
  ix.rotate(q, rq.data());
  SlimQIndex::Query Q;
  ix.prepare(rq.data(), Q);
  uint32_t cur = ix.enterpoint;
  float curd = ix.est(Q, cur);
  if (ctr) ctr->n_est++;
  for (int level = ix.maxlevel; level > ix.threshold_level; level--) {
    bool changed = true;
    while (changed) {
      changed = false;
      auto [ids, n] = ix.slice(cur, level);
      if (!ids || n == 0) continue;
      for (size_t i = 0; i < n; i++) {
        float d = ix.est(Q, ids[i]);
        if (ctr) ctr->n_est++;
        if (d < curd) { curd = d; cur = ids[i]; changed = true; }
      }
    }
  }
  Pool pool(ix.ef);
  std::vector<uint8_t> visited(ix.count, 0);
  auto less_first = [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) { return a.first < b.first; };
  pool.insert(cur, curd);
  while (pool.has_next()) {
    uint32_t node = pool.pop();
    if (trace) { trace->push_back(node | (visited[node] ? 1u << 31 : 0u)); trace->push_back((uint32_t)pool.size); }
    if (visited[node]) { if (ctr) ctr->n_revisit++; continue; }
    visited[node] = 1;
    auto [ids, n] = ix.slice(node, 0);
    if (!ids || n == 0) continue;   // NB: such a node is never reranked (:708-715 `continue` before :747)
    if (ctr) ctr->n_hops++;
    for (size_t j = 0; j < n; j++) {
      const uint32_t c = ids[j];
      const float d = ix.est(Q, c);
      if (ctr) ctr->n_est++;
      if (pool.is_full(d) || visited[c]) continue;
      pool.insert(c, d);
      if (ctr) ctr->n_insert++;
      if (trace) { uint32_t b; memcpy(&b, &d, 4); trace->push_back(c | (1u << 30)); trace->push_back(b); }
    }
    const float od = dist(ix.metric == METRIC_IP ? METRIC_IP : METRIC_L2, q, ix.raw + (size_t)node * ix.dim, ix.dim);
    heap.emplace_back(od, node);
    std::push_heap(heap.begin(), heap.end(), less_first);
    while (heap.size() > k) { std::pop_heap(heap.begin(), heap.end(), less_first); heap.pop_back(); }
  }
  return heap.size();
}

}  // namespace hso
