// oracle/hs_oracle.hpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (own code, C++17 + libstdc++ heap algorithms) of the reference's
// searchKnn -> searchBaseLayerST -> distance path, used ONLY as the parity checker by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product (hnsw-slim_amd/) never
// includes, links or calls anything in this directory.
//
// Pinning status (see DESIGN.md "Oracle pinning"):
//   * distance recipes + vanilla HierarchicalNSW search: pinned against the compiled reference
//     (oracle/_ref/ref_hnsw, built from /root/reference/third_party/hnswlib as-is) through the
//     fixtures in tests/golden/.
//   * HierarchicalNSWSlim (hnswalg_slim.h needs folly, absent here, so the class itself cannot be compiled): the
//     search, the CHAL slice addressing and the Slim loader are CROSS-PINNED to the compiled vanilla reference's golden
//     outputs through a verbatim vanilla -> Slim re-encoding of the reference-built graphs (oracle/chal_encode.py,
//     tests/test_oracle_golden.py::test_slim_search_on_verbatim_encoding_matches_reference); convertFromHNSW's pruning
//     is restated from source reading only.
//
// All file:line citations are relative to /root/reference/third_party/hnswlib/.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace hso {

using pairfi = std::pair<float, uint32_t>;
enum Metric { METRIC_L2 = 0, METRIC_IP = 1 };

// ----------------------------------------------------------------------------------------
// Distance recipes.  The reference picks the AVX-512 kernels on an AVX-512 host
// (space_l2.h:214-228, space_ip.h:345-371).  They are restated as scalar code that performs the
// same IEEE operations in the same order, so the value is bit-identical when the reference is
// compiled with -ffp-contract=off (our pinned oracle flags).  THIS FILE MUST BE
// COMPILED WITH -ffp-contract=off AND WITHOUT -ffast-math (oracle/Makefile does).
// ----------------------------------------------------------------------------------------

// space_l2.h:25-54  L2SqrSIMD16ExtAVX512: 16 lane accumulators, acc_j += (a-b)*(a-b) as a rounded
// multiply followed by a rounded add (:45), then TmpRes[0]+...+TmpRes[15] left to right (:49-51).
inline float l2_simd16(const float *a, const float *b, size_t d) {
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  size_t n16 = d >> 4;
  for (size_t s = 0; s < n16; s++)
    for (int j = 0; j < 16; j++) {
      float t = a[s * 16 + j] - b[s * 16 + j];
      float p = t * t;
      acc[j] = acc[j] + p;
    }
  float r = acc[0];
  for (int j = 1; j < 16; j++) r = r + acc[j];
  return r;
}
// space_l2.h:166-190  L2SqrSIMD4Ext (SSE): 4 lane accumulators, final TmpRes[0..3] left to right.
inline float l2_simd4(const float *a, const float *b, size_t d) {
  float acc[4] = {0, 0, 0, 0};
  size_t n4 = d >> 2;
  for (size_t s = 0; s < n4; s++)
    for (int j = 0; j < 4; j++) {
      float t = a[s * 4 + j] - b[s * 4 + j];
      float p = t * t;
      acc[j] = acc[j] + p;
    }
  return ((acc[0] + acc[1]) + acc[2]) + acc[3];
}
// space_l2.h:6-20  scalar L2Sqr.
inline float l2_scalar(const float *a, const float *b, size_t d) {
  float r = 0;
  for (size_t i = 0; i < d; i++) {
    float t = a[i] - b[i];
    float p = t * t;
    r = r + p;
  }
  return r;
}
// space_l2.h:227-234 dispatch.
inline float l2_dist(const float *a, const float *b, size_t d) {
  if (d % 16 == 0) return l2_simd16(a, b, d);
  if (d % 4 == 0) return l2_simd4(a, b, d);
  if (d > 16) {  // :149-160 L2SqrSIMD16ExtResiduals
    size_t d16 = d >> 4 << 4;
    float r = l2_simd16(a, b, d16);
    float t = l2_scalar(a + d16, b + d16, d - d16);
    return r + t;
  }
  if (d > 4) {  // :192-205 L2SqrSIMD4ExtResiduals
    size_t d4 = d >> 2 << 2;
    float r = l2_simd4(a, b, d4);
    float t = l2_scalar(a + d4, b + d4, d - d4);
    return r + t;
  }
  return l2_scalar(a, b, d);
}

// space_ip.h:146-199 InnerProductSIMD16ExtAVX512: 16 FMA accumulators (:183-195), then
// _mm512_reduce_add_ps (:197) = pairwise halves 16->8->4->2->1; distance = 1 - ip (:201-204).
inline float ip_simd16(const float *a, const float *b, size_t d) {
  float acc[16];
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  size_t n16 = d >> 4;
  for (size_t s = 0; s < n16; s++)
    for (int j = 0; j < 16; j++) acc[j] = std::fma(a[s * 16 + j], b[s * 16 + j], acc[j]);
  float h8[8], h4[4], h2[2];
  for (int j = 0; j < 8; j++) h8[j] = acc[j] + acc[j + 8];
  for (int j = 0; j < 4; j++) h4[j] = h8[j] + h8[j + 4];
  for (int j = 0; j < 2; j++) h2[j] = h4[j] + h4[j + 2];
  return h2[0] + h2[1];
}
// space_ip.h:24-69 InnerProductSIMD4ExtAVX (the SIMD4 entry on an AVX host): one ymm accumulator takes two 8-wide
// multiply-then-add steps per 16 elements (:41-53), its halves are added (:56), 4-wide steps follow (:58-64), and the
// four lanes are summed left to right (:67).
inline float ip_simd4(const float *a, const float *b, size_t d) {
  float y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t n16 = d >> 4, n4 = d >> 2, i = 0;
  for (size_t s = 0; s < n16; s++)
    for (int h = 0; h < 2; h++, i += 8)
      for (int j = 0; j < 8; j++) {
        float p = a[i + j] * b[i + j];
        y[j] = y[j] + p;
      }
  float x4[4];
  for (int j = 0; j < 4; j++) x4[j] = y[j] + y[j + 4];
  for (; i < n4 * 4; i += 4)
    for (int j = 0; j < 4; j++) {
      float p = a[i + j] * b[i + j];
      x4[j] = x4[j] + p;
    }
  return ((x4[0] + x4[1]) + x4[2]) + x4[3];
}
// space_ip.h:6-14 scalar InnerProduct.
inline float ip_scalar(const float *a, const float *b, size_t d) {
  float r = 0;
  for (size_t i = 0; i < d; i++) {
    float p = a[i] * b[i];
    r = r + p;
  }
  return r;
}
// space_ip.h:374-382 dispatch; residual forms :311-337 return 1 - (res + res_tail).
inline float ip_dist(const float *a, const float *b, size_t d) {
  if (d % 16 == 0) return 1.0f - ip_simd16(a, b, d);
  if (d % 4 == 0) return 1.0f - ip_simd4(a, b, d);
  if (d > 16) {
    size_t d16 = d >> 4 << 4;
    float r = ip_simd16(a, b, d16);
    float t = ip_scalar(a + d16, b + d16, d - d16);
    return 1.0f - (r + t);
  }
  if (d > 4) {
    size_t d4 = d >> 2 << 2;
    float r = ip_simd4(a, b, d4);
    float t = ip_scalar(a + d4, b + d4, d - d4);
    return 1.0f - (r + t);
  }
  return 1.0f - ip_scalar(a, b, d);
}
inline float dist(Metric m, const float *a, const float *b, size_t d) {
  return m == METRIC_L2 ? l2_dist(a, b, d) : ip_dist(a, b, d);
}

// ----------------------------------------------------------------------------------------
// File readers
// ----------------------------------------------------------------------------------------
struct Reader {
  std::ifstream in;
  explicit Reader(const std::string &p) : in(p, std::ios::binary) {
    if (!in.is_open()) throw std::runtime_error("Cannot open file");  // hnswalg.h:785-786
  }
  template <typename T> T pod() {
    T v;
    in.read((char *)&v, sizeof(T));
    if (!in) throw std::runtime_error("Index seems to be corrupted or unsupported");
    return v;
  }
  void bytes(void *dst, size_t n) {
    in.read((char *)dst, n);
    if (!in) throw std::runtime_error("Index seems to be corrupted or unsupported");
  }
};

// Vanilla index as written by HierarchicalNSW::saveIndex (hnswalg.h:748-779).
struct VanillaIndex {
  uint64_t offsetLevel0, max_elements, count, size_per_el, label_offset, offsetData;
  int32_t maxlevel;
  uint32_t enterpoint;
  uint64_t maxM, maxM0, M, efC;
  double mult;
  size_t dim;
  Metric metric;
  std::vector<char> level0;                // count * size_per_el
  std::vector<std::vector<char>> links;    // per element: level * (4 + 4*maxM) bytes
  std::vector<int> levels;
  size_t ef = 10;
  size_t num_deleted = 0;

  const float *vec(uint32_t i) const { return (const float *)(level0.data() + i * size_per_el + offsetData); }
  uint64_t label(uint32_t i) const {
    uint64_t l;
    memcpy(&l, level0.data() + i * size_per_el + label_offset, 8);
    return l;
  }
  // hnswalg.h:525-529, 1012-1014: level-0 list = [u16 cnt][u8 flags][u8][u32 ids[maxM0]]
  const uint32_t *list0(uint32_t i, size_t &cnt) const {
    const char *p = level0.data() + i * size_per_el + offsetLevel0;
    uint16_t c;
    memcpy(&c, p, 2);
    cnt = c;
    return (const uint32_t *)(p + 4);
  }
  // hnswalg.h:538-541
  const uint32_t *list(uint32_t i, int level, size_t &cnt) const {
    const char *p = links[i].data() + (size_t)(level - 1) * (4 + 4 * maxM);
    uint16_t c;
    memcpy(&c, p, 2);
    cnt = c;
    return (const uint32_t *)(p + 4);
  }
  // A filter (BaseFilterFunctor) is folded in as allowed[id]: the reference tests
  // !isMarkedDeleted(id) && (*isIdAllowed)(label) together everywhere (hnswalg.h:348-349, 442-444).
  std::vector<uint8_t> allowed;  // by internal id; empty = no filter
  bool mark(uint32_t i) const {  // hnswalg.h:~1000: byte 2 of the level-0 header & 1
    return (level0[i * size_per_el + offsetLevel0 + 2] & 1) != 0;
  }
  bool deleted(uint32_t i) const { return mark(i) || (!allowed.empty() && !allowed[i]); }

  void load(const std::string &path, Metric m, size_t dim_) {  // hnswalg.h:781-893
    Reader r(path);
    metric = m;
    dim = dim_;
    offsetLevel0 = r.pod<uint64_t>();
    max_elements = r.pod<uint64_t>();
    count = r.pod<uint64_t>();
    size_per_el = r.pod<uint64_t>();
    label_offset = r.pod<uint64_t>();
    offsetData = r.pod<uint64_t>();
    maxlevel = r.pod<int32_t>();
    enterpoint = r.pod<uint32_t>();
    maxM = r.pod<uint64_t>();
    maxM0 = r.pod<uint64_t>();
    M = r.pod<uint64_t>();
    mult = r.pod<double>();
    efC = r.pod<uint64_t>();
    if (size_per_el != 4 + 4 * maxM0 + 4 * dim + 8)
      throw std::runtime_error("Index seems to be corrupted or unsupported");
    level0.resize(count * size_per_el);
    r.bytes(level0.data(), level0.size());
    links.resize(count);
    levels.resize(count);
    size_t per = 4 + 4 * maxM;
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = r.pod<uint32_t>();
      levels[i] = sz / per;
      links[i].resize(sz);
      if (sz) r.bytes(links[i].data(), sz);
    }
    if (r.in.peek() != EOF) throw std::runtime_error("Index seems to be corrupted or unsupported");
    num_deleted = 0;
    for (size_t i = 0; i < count; i++) num_deleted += mark(i);
  }
};

// Slim index as written by HierarchicalNSWSlim::saveIndex (hnswalg_slim.h:717-751).
struct SlimIndex {
  uint64_t count, size_per_el, label_offset, offsetTotal, offsetData, offsetNeighbor;
  int32_t maxlevel, threshold_level;
  uint32_t enterpoint;
  uint64_t maxM, maxM0, M, efC;
  uint8_t has_deleted;
  size_t dim;
  Metric metric;
  std::vector<char> elements;            // count * size_per_el
  std::vector<std::vector<char>> blobs;  // per element [u16 off[level]][u32 ids[total]]
  size_t ef = 10;

  const char *el(uint32_t i) const { return elements.data() + (size_t)i * size_per_el; }
  int32_t level(uint32_t i) const { int32_t v; memcpy(&v, el(i), 4); return v; }               // :636
  uint32_t total(uint32_t i) const { uint32_t v; memcpy(&v, el(i) + offsetTotal, 4); return v; }  // :644
  uint64_t label(uint32_t i) const { uint64_t v; memcpy(&v, el(i) + label_offset, 8); return v; } // :195
  const float *vec(uint32_t i) const { return (const float *)(el(i) + offsetData); }            // :208
  std::vector<uint8_t> allowed;  // filter by internal id (searchKnn(q,k,isIdAllowed), hnswalg_slim.h:1783-1905)
  bool deleted(uint32_t i) const { return (el(i)[4 + 2] & 1) != 0 || (!allowed.empty() && !allowed[i]); }  // :1776-1781, :578-580
  // slice of node i at `lvl` (hnswalg_slim.h:2050-2062 / :363-369); false when the node has no blob
  bool slice(uint32_t i, int lvl, const uint32_t *&ids, size_t &n) const {
    if (blobs[i].empty()) return false;  // neighbors == nullptr (:2047, :360)
    int32_t L = level(i);
    const uint16_t *off = (const uint16_t *)blobs[i].data();
    size_t start = lvl == 0 ? 0 : off[lvl - 1];
    size_t end = (lvl == L) ? total(i) : off[lvl];
    n = end - start;
    ids = (const uint32_t *)(blobs[i].data() + 2 * (size_t)L) + start;
    return true;
  }

  void load(const std::string &path, Metric m, size_t dim_) {  // hnswalg_slim.h:753-815
    Reader r(path);
    metric = m;
    dim = dim_;
    count = r.pod<uint64_t>();
    size_per_el = r.pod<uint64_t>();
    label_offset = r.pod<uint64_t>();
    offsetTotal = r.pod<uint64_t>();
    offsetData = r.pod<uint64_t>();
    offsetNeighbor = r.pod<uint64_t>();
    maxlevel = r.pod<int32_t>();
    threshold_level = r.pod<int32_t>();
    enterpoint = r.pod<uint32_t>();
    maxM = r.pod<uint64_t>();
    maxM0 = r.pod<uint64_t>();
    M = r.pod<uint64_t>();
    efC = r.pod<uint64_t>();
    has_deleted = r.pod<uint8_t>();
    if (size_per_el != offsetData + 4 * dim)
      throw std::runtime_error("Index seems to be corrupted or unsupported");
    elements.resize(count * size_per_el);
    r.bytes(elements.data(), elements.size());
    blobs.assign(count, {});
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = r.pod<uint32_t>();
      if (sz == 0 || total(i) == 0) continue;  // :799-801
      blobs[i].resize(sz);
      r.bytes(blobs[i].data(), sz);
    }
  }
};

// ----------------------------------------------------------------------------------------
// Search restatements
// ----------------------------------------------------------------------------------------
struct Counters {
  uint32_t n_dist = 0;      // distance evaluations (entry + upper layers + level 0)
  uint32_t n_hops = 0;      // adjacency rows looked up (upper-layer steps + level-0 expansions)
  uint32_t n_nbr = 0;       // neighbour ids scanned (visited or not)
  uint32_t n_accept = 0;    // candidates pushed into the candidate heap
  uint32_t max_cand = 0;    // peak size of the candidate heap
};

struct cmp_max { bool operator()(const pairfi &a, const pairfi &b) const { return a.first < b.first; } };  // slim.h:169-175
struct cmp_min { bool operator()(const pairfi &a, const pairfi &b) const { return a.first > b.first; } };  // slim.h:177-183

struct Scratch {
  std::vector<uint16_t> visited;
  uint16_t tag = 0;
  std::vector<pairfi> top, cand;
  void begin(size_t n) {  // visited_list_pool.h:22-28 epoch reset
    if (visited.size() != n) { visited.assign(n, 0); tag = 0; }
    tag++;
    if (tag == 0) { std::fill(visited.begin(), visited.end(), 0); tag++; }
  }
};

// Level-0 beam, restating hnswalg_slim.h:321-457 (bare_bone / non-bare_bone, no stop_condition).
// Works on top/cand arrays exactly as the reference does (raw arrays + std::push_heap/pop_heap).
template <class Index, class GetList0>
inline void beam_level0(const Index &ix, const float *q, size_t ef, bool bare_bone, Scratch &s,
                        float &lowerBound, Counters &c, GetList0 list0, bool seed_from_top = true) {
  if (seed_from_top) {
    s.cand.assign(s.top.begin(), s.top.end());                    // :327-329
    std::make_heap(s.cand.begin(), s.cand.end(), cmp_min());      // :331-332
  }
  while (!s.cand.empty()) {
    pairfi cur = s.cand.front();
    bool stop = bare_bone ? (cur.first > lowerBound)              // :340
                          : (cur.first > lowerBound && s.top.size() == ef);  // :346-347
    if (stop) break;
    std::pop_heap(s.cand.begin(), s.cand.end(), cmp_min());       // :353-354
    s.cand.pop_back();
    const uint32_t *ids;
    size_t n;
    c.n_hops++;
    if (!list0(cur.second, ids, n)) continue;                      // :360-362
    if (n == 0) continue;                                          // :366-367
    for (size_t j = 0; j < n; j++) {                               // :383
      uint32_t id = ids[j];
      c.n_nbr++;
      if (s.visited[id] == s.tag) continue;                        // :392
      s.visited[id] = s.tag;
      float d = dist(ix.metric, q, ix.vec(id), ix.dim);            // :396
      c.n_dist++;
      if (s.top.size() < ef || lowerBound > d) {                   // :403-404
        s.cand.emplace_back(d, id);                                // :408-411
        std::push_heap(s.cand.begin(), s.cand.end(), cmp_min());
        c.n_accept++;
        c.max_cand = std::max<uint32_t>(c.max_cand, s.cand.size());
        if (bare_bone || !ix.deleted(id)) {                        // :418
          s.top.emplace_back(d, id);
          std::push_heap(s.top.begin(), s.top.end(), cmp_max());
        }
        while (s.top.size() > ef) {                                // :434-448
          std::pop_heap(s.top.begin(), s.top.end(), cmp_max());
          s.top.pop_back();
        }
        if (!s.top.empty()) lowerBound = s.top.front().first;      // :450-452
      }
    }
  }
}

struct SlimResult {
  std::vector<pairfi> top;  // raw top_candidates array after the level-0 beam (heap order)
  Counters c;
};

// Upper-layer greedy of HierarchicalNSWSlim::searchKnn (hnswalg_slim.h:2040-2078).
inline void slim_upper(const SlimIndex &ix, const float *q, uint32_t &cur, float &curdist, Counters &c) {
  for (int lvl = ix.maxlevel; lvl > ix.threshold_level; lvl--) {
    bool changed = true;
    while (changed) {
      changed = false;
      const uint32_t *ids;
      size_t n;
      c.n_hops++;
      if (!ix.slice(cur, lvl, ids, n)) continue;   // :2047-2049
      if (n == 0) continue;                          // :2057-2058
      for (size_t i = 0; i < n; i++) {               // scan continues over the OLD node's list
        uint32_t cand = ids[i];
        c.n_nbr++;
        float d = dist(ix.metric, q, ix.vec(cand), ix.dim);
        c.n_dist++;
        if (d < curdist) { curdist = d; cur = cand; changed = true; }  // :2071-2075
      }
    }
  }
}

// searchBaseLayer(layer) for 0 < layer <= threshold_level (hnswalg_slim.h:222-316).
inline void slim_beam_layer(const SlimIndex &ix, const float *q, int layer, size_t ef, Scratch &s,
                            float &lowerBound, Counters &c) {
  s.cand.assign(s.top.begin(), s.top.end());
  std::make_heap(s.cand.begin(), s.cand.end(), cmp_min());
  while (!s.cand.empty()) {
    pairfi cur = s.cand.front();
    if (cur.first > lowerBound && s.top.size() == ef) break;       // :237
    std::pop_heap(s.cand.begin(), s.cand.end(), cmp_min());
    s.cand.pop_back();
    const uint32_t *ids;
    size_t n;
    c.n_hops++;
    if (!ix.slice(cur.second, layer, ids, n)) continue;
    if (n == 0) continue;
    for (size_t j = 0; j < n; j++) {
      uint32_t id = ids[j];
      c.n_nbr++;
      if (s.visited[id] == s.tag) continue;
      s.visited[id] = s.tag;
      float d = dist(ix.metric, q, ix.vec(id), ix.dim);
      c.n_dist++;
      if (s.top.size() < ef || lowerBound > d) {                   // :284
        s.cand.emplace_back(d, id);
        std::push_heap(s.cand.begin(), s.cand.end(), cmp_min());
        c.n_accept++;
        c.max_cand = std::max<uint32_t>(c.max_cand, s.cand.size());
        if (!ix.deleted(id)) {                                     // :297
          s.top.emplace_back(d, id);
          std::push_heap(s.top.begin(), s.top.end(), cmp_max());
        }
        if (s.top.size() > ef) {                                   // :304-308
          std::pop_heap(s.top.begin(), s.top.end(), cmp_max());
          s.top.pop_back();
        }
        if (!s.top.empty()) lowerBound = s.top.front().first;      // :310-312
      }
    }
  }
}

// Common body of the three HierarchicalNSWSlim::searchKnn overloads up to the end of the
// level-0 beam.  mark_ep: the (q,k) and (q,k,filter) overloads tag the enter point as visited
// before the descent (hnswalg_slim.h:1796, 1919); the (q,k,tableint*) overload does not (:2036-2038).
inline SlimResult slim_search_core(const SlimIndex &ix, const float *q, size_t k, Scratch &s, bool mark_ep) {
  SlimResult r;
  s.begin(ix.count);
  uint32_t cur = ix.enterpoint;
  float curdist = dist(ix.metric, q, ix.vec(cur), ix.dim);  // :2033-2035
  r.c.n_dist++;
  if (mark_ep) s.visited[cur] = s.tag;
  slim_upper(ix, q, cur, curdist, r.c);
  size_t ef = std::max(ix.ef, k);                            // :2080
  s.top.clear();
  s.top.emplace_back(curdist, cur);                          // :2100-2101
  s.visited[cur] = s.tag;                                    // :2102
  float lowerBound = !ix.deleted(cur) ? curdist : std::numeric_limits<float>::max();  // :2104-2106
  for (int lvl = std::min(ix.threshold_level, ix.maxlevel); lvl > 0; lvl--)            // :2108-2113
    slim_beam_layer(ix, q, lvl, ef, s, lowerBound, r.c);
  bool bare = !ix.has_deleted && ix.allowed.empty();         // :2114, :1884
  beam_level0(ix, q, ef, bare, s, lowerBound, r.c,
              [&](uint32_t id, const uint32_t *&ids, size_t &n) { return ix.slice(id, 0, ids, n); });
  r.top = s.top;
  return r;
}

// Level-0 entry of a query: the node the upper-layer greedy ends on (hnswalg_slim.h:2033-2078).
inline uint32_t slim_entry(const SlimIndex &ix, const float *q) {
  Counters c;
  uint32_t cur = ix.enterpoint;
  float curdist = dist(ix.metric, q, ix.vec(cur), ix.dim);
  slim_upper(ix, q, cur, curdist, c);
  return cur;
}

// searchKnn(q, k, tableint* result)  (hnswalg_slim.h:2030-2131): nth_element + label truncation.
inline SlimResult slim_search_ids(const SlimIndex &ix, const float *q, size_t k, Scratch &s, uint32_t *out) {
  SlimResult r = slim_search_core(ix, q, k, s, /*mark_ep=*/false);
  std::vector<pairfi> t = r.top;
  if (t.size() < k) throw std::runtime_error("oracle: top_size < k (UB in reference, slim.h:2126)");
  std::nth_element(t.begin(), t.begin() + k, t.end(), cmp_max());   // :2126-2127
  for (size_t i = 0; i < k; i++) out[i] = (uint32_t)ix.label(t[i].second);  // :2128-2130
  return r;
}

// searchKnn(q, k) -> priority_queue (hnswalg_slim.h:1907-2028).  Returned in pop order
// (farthest first) as (dist,label).
// mark_ep = false is NOT a reference overload: it is the cross-pin knob of tests/test_oracle_golden.py (the filter branch
// of the beam against the compiled vanilla reference, which does not pre-mark its enter point).
inline SlimResult slim_search_pq(const SlimIndex &ix, const float *q, size_t k, Scratch &s,
                                 std::vector<std::pair<float, uint64_t>> &out, bool mark_ep = true) {
  SlimResult r = slim_search_core(ix, q, k, s, mark_ep);
  std::vector<pairfi> t = r.top;
  while (t.size() > k) {                                             // :2019-2022
    std::pop_heap(t.begin(), t.end(), cmp_max());
    t.pop_back();
  }
  std::priority_queue<std::pair<float, uint64_t>> pq;                // :2023-2026
  for (auto &e : t) pq.emplace(e.first, ix.label(e.second));
  out.clear();
  while (!pq.empty()) { out.push_back(pq.top()); pq.pop(); }
  return r;
}

// HierarchicalNSW::searchKnn (hnswalg.h:1378-1440) + searchBaseLayerST (hnswalg.h:326-479).
inline SlimResult vanilla_search_pq(const VanillaIndex &ix, const float *q, size_t k, Scratch &s,
                                    std::vector<std::pair<float, uint64_t>> &out) {
  SlimResult r;
  out.clear();
  if (ix.count == 0) return r;
  s.begin(ix.count);
  uint32_t cur = ix.enterpoint;
  float curdist = dist(ix.metric, q, ix.vec(cur), ix.dim);  // :1386-1387
  r.c.n_dist++;
  for (int lvl = ix.maxlevel; lvl > 0; lvl--) {              // :1389-1415
    bool changed = true;
    while (changed) {
      changed = false;
      size_t n;
      const uint32_t *ids = ix.list(cur, lvl, n);
      r.c.n_hops++;
      for (size_t i = 0; i < n; i++) {
        uint32_t cand = ids[i];
        r.c.n_nbr++;
        float d = dist(ix.metric, q, ix.vec(cand), ix.dim);
        r.c.n_dist++;
        if (d < curdist) { curdist = d; cur = cand; changed = true; }
      }
    }
  }
  size_t ef = std::max(ix.ef, k);
  bool bare = ix.num_deleted == 0 && ix.allowed.empty();     // :1421
  // searchBaseLayerST prologue (hnswalg.h:346-364): recomputes the entry distance.
  s.top.clear();
  float lowerBound;
  if (bare || !ix.deleted(cur)) {
    float d = dist(ix.metric, q, ix.vec(cur), ix.dim);       // :351
    r.c.n_dist++;
    lowerBound = d;
    s.top.emplace_back(d, cur);
    s.cand.assign(1, pairfi(d, cur));                        // :358
  } else {
    // (:360-361) deleted entry point: candidate (-FLT_MAX, ep) with an empty result heap
    lowerBound = std::numeric_limits<float>::max();
    s.cand.assign(1, pairfi(lowerBound, cur));
  }
  s.visited[cur] = s.tag;
  // candidate_set holds (-dist,id) in a max-heap by .first (:358, :432) == min-heap on dist with
  // identical sift decisions; beam_level0 seeds cand from top, so handle the deleted-entry case here.
  beam_level0(ix, q, ef, bare, s, lowerBound, r.c, [&](uint32_t id, const uint32_t *&ids, size_t &n) {
    ids = ix.list0(id, n);
    return true;
  }, /*seed_from_top=*/false);
  r.top = s.top;
  std::vector<pairfi> t = r.top;
  while (t.size() > k) {                                     // :1430-1432
    std::pop_heap(t.begin(), t.end(), cmp_max());
    t.pop_back();
  }
  std::priority_queue<std::pair<float, uint64_t>> pq;        // :1433-1438
  while (!t.empty()) {
    pq.emplace(t.front().first, ix.label(t.front().second));
    std::pop_heap(t.begin(), t.end(), cmp_max());
    t.pop_back();
  }
  while (!pq.empty()) { out.push_back(pq.top()); pq.pop(); }
  return r;
}

// Exact brute-force k-NN (bruteforce.h:106-135 semantics: k smallest distances), ids ascending by
// (dist, id) -- ground truth for recall.
inline void brute_force(Metric m, const float *base, size_t n, size_t d, const float *q, size_t k,
                        uint32_t *out_ids) {
  std::vector<pairfi> all(n);
  for (size_t i = 0; i < n; i++) all[i] = {dist(m, q, base + i * d, d), (uint32_t)i};
  std::partial_sort(all.begin(), all.begin() + std::min(k, n), all.end());
  for (size_t i = 0; i < std::min(k, n); i++) out_ids[i] = all[i].second;
}

}  // namespace hso
