// host_graph.hpp -- host-side (CPU) index structures around the GPU search path:
//   * VanillaGraph : hnswlib::HierarchicalNSW memory image; serial-reproducible builder (harness,
//                    restating hnswalg.h:229-322, 481-687, 1248-1376) + saveIndex/loadIndex
//                    (hnswalg.h:748-893).  A serial build writes a file byte-identical to the
//                    reference's (tests/test_host_graph.py vs tests/golden/*.hnsw.bin).
//   * SlimGraph    : hnswlib::HierarchicalNSWSlim image: convertFromHNSW (hnswalg_slim.h:836-1108),
//                    saveIndex/loadIndex (hnswalg_slim.h:717-815).
//   * PackedIndex  : what the GPU consumes -- vectors row-major + CSR adjacency (level 0 and upper
//                    levels), labels, delete flags -- produced from either image.
// Index construction/pruning is harness for the search path (SURVEY.md 2, rows 5-6 "harness"); it is
// CPU code and not accelerated this round.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dist_recipe.hpp"

namespace hs {

using pairfi = std::pair<float, uint32_t>;
struct CmpFirst { bool operator()(const pairfi &a, const pairfi &b) const { return a.first < b.first; } };  // hnswalg.h:176-182
using MaxQ = std::priority_queue<pairfi, std::vector<pairfi>, CmpFirst>;

// Where a serialized index comes from: a file, or the same bytes already in host memory (hs_index_load_mem).
struct BinSource {
  std::string path;
  const char *mem = nullptr;
  size_t len = 0;
  BinSource(const std::string &p) : path(p) {}
  BinSource(const char *p) : path(p) {}
  BinSource(const void *m, size_t n) : mem((const char *)m), len(n) {}
};
struct MemBuf : std::streambuf {
  MemBuf(const char *b, size_t n) { setg(const_cast<char *>(b), const_cast<char *>(b), const_cast<char *>(b) + n); }
  pos_type seekoff(off_type off, std::ios_base::seekdir dir, std::ios_base::openmode) override {
    char *base = eback(), *end = egptr();
    char *to = dir == std::ios_base::beg ? base + off : dir == std::ios_base::end ? end + off : gptr() + off;
    if (to < base || to > end) return pos_type(off_type(-1));
    setg(base, to, end);
    return pos_type(to - base);
  }
  pos_type seekpos(pos_type pos, std::ios_base::openmode m) override { return seekoff(off_type(pos), std::ios_base::beg, m); }
};
struct BinReader {
  std::ifstream file;
  MemBuf mb;
  std::istream mem_in;
  std::istream &in;
  explicit BinReader(const BinSource &s) : mb(s.mem, s.mem ? s.len : 0), mem_in(&mb), in(s.mem ? mem_in : static_cast<std::istream &>(file)) {
    if (!s.mem) {
      file.open(s.path, std::ios::binary);
      if (!file.is_open()) throw std::runtime_error("Cannot open file");
    }
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
template <typename T> static void put(std::ostream &o, const T &v) { o.write((const char *)&v, sizeof(T)); }

inline double branching_mult(const std::string &bf, size_t) {  // hnswalg.h:143-158
  if (bf == "e") return 1 / log(M_E);
  if (bf == "sqrt") return 1 / log(sqrt(2.0) / (sqrt(2.0) - 1));
  try {
    return 1 / log(std::stod(bf));
  } catch (const std::invalid_argument &) {
    throw std::runtime_error("Invalid branching factor: " + bf);
  } catch (const std::out_of_range &) {
    throw std::runtime_error("Branching factor out of range: " + bf);
  }
}

// raw_dist_func_ = euclidean_sqr / dot_product_dis (rabitqlib/utils/space.hpp:226-240): `(v0 - v1).dot(v0 - v1)` and
// `1 - v0.dot(v1)` through the vendored Eigen's inner product (Eigen/src/Core/InnerProduct.h:113-159) with 16-float packets
// (the AVX-512 build the parity flags pin): four packet accumulators, the first four packets multiplied, every later one
// fused-multiply-added (pmadd = _mm512_fmadd_ps), folded 2+=3, 1+=2, 0+=1, then predux = halves 8, 4, (0+2, 1+3), (0+1)
// (arch/AVX512/PacketMath.h:1456-1461, AVX :1954, SSE :1852-1853); the < 16 tail is a scalar fused chain.  Unlike the Eigen
// reductions on the SEARCH path this one does not depend on pointer alignment: the operands are expressions.
inline float rabitq_raw_dist(Metric metric, const float *a, const float *b, size_t d) {
  const bool l2 = metric == METRIC_L2;
  auto X = [&](size_t k) { return l2 ? a[k] - b[k] : a[k]; };
  auto Y = [&](size_t k) { return l2 ? a[k] - b[k] : b[k]; };
  float res;
  if (d < 16) {
    if (d == 0) return l2 ? 0.f : 1.f;
    res = X(0) * Y(0);
    for (size_t k = 1; k < d; k++) res = __builtin_fmaf(X(k), Y(k), res);
  } else {
    const size_t packet_end = d / 16 * 16, quad_end = d / 64 * 64, np = d / 16, nrem = (packet_end - quad_end) / 16;
    float acc[4][16];
    for (size_t p = 0; p < std::min<size_t>(np, 4); p++)
      for (size_t j = 0; j < 16; j++) acc[p][j] = X(p * 16 + j) * Y(p * 16 + j);
    if (np >= 4) {
      for (size_t k = 64; k < quad_end; k += 64)
        for (size_t p = 0; p < 4; p++)
          for (size_t j = 0; j < 16; j++) acc[p][j] = __builtin_fmaf(X(k + p * 16 + j), Y(k + p * 16 + j), acc[p][j]);
      for (size_t p = 0; p < nrem; p++)
        for (size_t j = 0; j < 16; j++) acc[p][j] = __builtin_fmaf(X(quad_end + p * 16 + j), Y(quad_end + p * 16 + j), acc[p][j]);
      for (size_t j = 0; j < 16; j++) acc[2][j] = acc[2][j] + acc[3][j];
    }
    if (np >= 3)
      for (size_t j = 0; j < 16; j++) acc[1][j] = acc[1][j] + acc[2][j];
    if (np >= 2)
      for (size_t j = 0; j < 16; j++) acc[0][j] = acc[0][j] + acc[1][j];
    float t8[8], t4[4];
    for (size_t j = 0; j < 8; j++) t8[j] = acc[0][j] + acc[0][j + 8];
    for (size_t j = 0; j < 4; j++) t4[j] = t8[j] + t8[j + 4];
    const float u0 = t4[0] + t4[2], u1 = t4[1] + t4[3];
    res = u0 + u1;
    for (size_t k = packet_end; k < d; k++) res = __builtin_fmaf(X(k), Y(k), res);
  }
  return l2 ? res : 1 - res;
}

// ------------------------------------------------------------------------------------------------
struct VanillaGraph {
  size_t max_elements = 0, count = 0, dim = 0;
  size_t M = 0, maxM = 0, maxM0 = 0, efC = 0;
  size_t size_links0 = 0, size_per_el = 0, offsetData = 0, label_offset = 0, size_links_up = 0;
  double mult = 0;
  int maxlevel = -1;
  uint32_t enterpoint = (uint32_t)-1;
  Metric metric = METRIC_L2;
  std::vector<char> level0;                 // max_elements * size_per_el
  std::vector<std::vector<char>> links;     // per element: level * size_links_up (+1 as the reference mallocs)
  std::vector<int> levels;
  std::unique_ptr<std::mutex[]> locks;
  std::mutex global;

  void init(size_t n, size_t d, Metric m, size_t M_, size_t efC_, const std::string &bf) {
    max_elements = n; dim = d; metric = m;
    M = M_ > 10000 ? 10000 : M_;           // hnswalg.h:97-107
    maxM = M; maxM0 = 2 * M;
    efC = std::max(efC_, M);
    size_links0 = maxM0 * 4 + 4;            // hnswalg.h:116
    size_per_el = size_links0 + 4 * d + 8;
    offsetData = size_links0;
    label_offset = size_links0 + 4 * d;
    size_links_up = maxM * 4 + 4;           // hnswalg.h:141-142
    mult = branching_mult(bf, M);
    level0.assign(n * size_per_el, 0);
    links.assign(n, {});
    levels.assign(n, 0);
    locks.reset(new std::mutex[n]);
    count = 0; maxlevel = -1; enterpoint = (uint32_t)-1;
  }
  char *el(uint32_t i) { return level0.data() + (size_t)i * size_per_el; }
  const char *el(uint32_t i) const { return level0.data() + (size_t)i * size_per_el; }
  const float *vec(uint32_t i) const { return (const float *)(el(i) + offsetData); }
  uint64_t label(uint32_t i) const { uint64_t l; memcpy(&l, el(i) + label_offset, 8); return l; }
  uint32_t *list_at(uint32_t i, int level) {  // hnswalg.h:543-547: [u16 cnt][u16][u32 ids...]
    return level == 0 ? (uint32_t *)el(i) : (uint32_t *)(links[i].data() + (size_t)(level - 1) * size_links_up);
  }
  const uint32_t *list_at(uint32_t i, int level) const { return const_cast<VanillaGraph *>(this)->list_at(i, level); }
  static uint16_t cnt_of(const uint32_t *l) { return *(const uint16_t *)l; }       // hnswalg.h:1012-1014
  static void set_cnt(uint32_t *l, uint16_t c) { *(uint16_t *)l = c; }              // hnswalg.h:1016-1018
  bool deleted(uint32_t i) const { return (((const unsigned char *)el(i))[2] & 1) != 0; }
  float dist(const float *a, const float *b) const { return host_dist(metric, a, b, dim); }

  struct Visited {
    std::vector<uint16_t> mass; uint16_t cur = 0;
    void begin(size_t n) {
      if (mass.size() != n) { mass.assign(n, 0); cur = 0; }
      if (++cur == 0) { std::fill(mass.begin(), mass.end(), 0); cur = 1; }
    }
  };

  // searchBaseLayer (hnswalg.h:229-322): build-time beam with ef_construction.
  MaxQ search_layer(uint32_t ep, const float *q, int layer, Visited &vl) {
    vl.begin(max_elements);
    MaxQ top, cand;
    float lower;
    if (!deleted(ep)) {
      float d = dist(q, vec(ep));
      top.emplace(d, ep);
      lower = d;
      cand.emplace(-d, ep);
    } else {
      lower = std::numeric_limits<float>::max();
      cand.emplace(-lower, ep);
    }
    vl.mass[ep] = vl.cur;
    while (!cand.empty()) {
      pairfi c = cand.top();
      if ((-c.first) > lower && top.size() == efC) break;
      cand.pop();
      uint32_t node = c.second;
      std::unique_lock<std::mutex> lk(locks[node]);
      const uint32_t *l = list_at(node, layer);
      size_t n = cnt_of(l);
      for (size_t j = 0; j < n; j++) {
        uint32_t id = l[1 + j];
        if (vl.mass[id] == vl.cur) continue;
        vl.mass[id] = vl.cur;
        float d = dist(q, vec(id));
        if (top.size() < efC || lower > d) {
          cand.emplace(-d, id);
          if (!deleted(id)) top.emplace(d, id);
          if (top.size() > efC) top.pop();
          if (!top.empty()) lower = top.top().first;
        }
      }
    }
    return top;
  }

  // getNeighborsByHeuristic2 (hnswalg.h:481-523)
  void heuristic(MaxQ &top, size_t Mlim) const {
    if (top.size() < Mlim) return;
    std::priority_queue<pairfi> closest;  // default pair ordering, as the reference
    std::vector<pairfi> keep;
    while (!top.empty()) { closest.emplace(-top.top().first, top.top().second); top.pop(); }
    while (!closest.empty()) {
      if (keep.size() >= Mlim) break;
      pairfi cur = closest.top();
      float dq = -cur.first;
      closest.pop();
      bool good = true;
      for (const pairfi &s : keep) {
        float d = dist(vec(s.second), vec(cur.second));
        if (d < dq) { good = false; break; }
      }
      if (good) keep.push_back(cur);
    }
    for (const pairfi &p : keep) top.emplace(-p.first, p.second);
  }

  // mutuallyConnectNewElement (hnswalg.h:549-687), isUpdate == false
  uint32_t connect(uint32_t cur_c, MaxQ &top, int level) {
    size_t Mcurmax = level ? maxM : maxM0;
    heuristic(top, M);
    if (top.size() > M) throw std::runtime_error("Should be not be more than M_ candidates returned by the heuristic");
    std::vector<uint32_t> sel;
    sel.reserve(M);
    while (!top.empty()) { sel.push_back(top.top().second); top.pop(); }
    uint32_t next_ep = sel.back();
    {
      uint32_t *l = list_at(cur_c, level);
      if (*l) throw std::runtime_error("The newly inserted element should have blank link list");
      set_cnt(l, sel.size());
      for (size_t i = 0; i < sel.size(); i++) {
        if (l[1 + i]) throw std::runtime_error("Possible memory corruption");
        if (level > levels[sel[i]]) throw std::runtime_error("Trying to make a link on a non-existent level");
        l[1 + i] = sel[i];
      }
    }
    for (size_t i = 0; i < sel.size(); i++) {
      std::unique_lock<std::mutex> lk(locks[sel[i]]);
      uint32_t *lo = list_at(sel[i], level);
      size_t sz = cnt_of(lo);
      if (sz > Mcurmax) throw std::runtime_error("Bad value of sz_link_list_other");
      if (sel[i] == cur_c) throw std::runtime_error("Trying to connect an element to itself");
      if (level > levels[sel[i]]) throw std::runtime_error("Trying to make a link on a non-existent level");
      uint32_t *data = lo + 1;
      if (sz < Mcurmax) {
        data[sz] = cur_c;
        set_cnt(lo, sz + 1);
      } else {
        float dmax = dist(vec(cur_c), vec(sel[i]));
        MaxQ cands;
        cands.emplace(dmax, cur_c);
        for (size_t j = 0; j < sz; j++) cands.emplace(dist(vec(data[j]), vec(sel[i])), data[j]);
        heuristic(cands, Mcurmax);
        int idx = 0;
        while (!cands.empty()) { data[idx++] = cands.top().second; cands.pop(); }
        set_cnt(lo, idx);
      }
    }
    return next_ep;
  }

  // addPoint(data, label, level=-1) for a NEW label (hnswalg.h:1248-1376). cur_c and its level are
  // supplied by the caller so that serial and parallel builds assign ids == insertion index.
  void add_point(const float *x, uint64_t label, uint32_t cur_c, int curlevel, Visited &vl) {
    std::unique_lock<std::mutex> lock_el(locks[cur_c]);
    levels[cur_c] = curlevel;
    std::unique_lock<std::mutex> templock(global);
    int maxlevelcopy = maxlevel;
    if (curlevel <= maxlevelcopy) templock.unlock();
    uint32_t currObj = enterpoint;
    uint32_t ep_copy = enterpoint;
    memset(el(cur_c), 0, size_per_el);
    memcpy(el(cur_c) + label_offset, &label, 8);
    memcpy(el(cur_c) + offsetData, x, 4 * dim);
    if (curlevel) links[cur_c].assign(size_links_up * curlevel + 1, 0);
    if ((int32_t)currObj != -1) {
      if (curlevel < maxlevelcopy) {
        float curdist = dist(x, vec(currObj));
        for (int level = maxlevelcopy; level > curlevel; level--) {
          bool changed = true;
          while (changed) {
            changed = false;
            std::unique_lock<std::mutex> lk(locks[currObj]);
            const uint32_t *l = list_at(currObj, level);
            int n = cnt_of(l);
            for (int i = 0; i < n; i++) {
              uint32_t cand = l[1 + i];
              if (cand > max_elements) throw std::runtime_error("cand error");
              float d = dist(x, vec(cand));
              if (d < curdist) { curdist = d; currObj = cand; changed = true; }
            }
          }
        }
      }
      bool epDeleted = deleted(ep_copy);
      for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
        MaxQ top = search_layer(currObj, x, level, vl);
        if (epDeleted) {
          top.emplace(dist(x, vec(ep_copy)), ep_copy);
          if (top.size() > efC) top.pop();
        }
        currObj = connect(cur_c, top, level);
      }
    } else {
      enterpoint = 0;
      maxlevel = curlevel;
    }
    if (curlevel > maxlevelcopy) {
      enterpoint = cur_c;
      maxlevel = curlevel;
    }
  }

  // Build over rows 0..n-1 with labels == row index.  Levels are drawn up front from the reference's
  // generator in row order (std::default_random_engine seeded `seed`, hnswalg.h:113,217-221), so a
  // threads==1 build reproduces the reference's serial addPoint loop bit for bit; threads>1 keeps
  // ids == labels (the reference's parallel build does not, hnswalg.h:1279-1281) but the graph then
  // depends on thread timing, exactly as upstream hnswlib.
  void build(const float *base, size_t n, size_t d, Metric m, size_t M_, size_t efC_, const std::string &bf,
             size_t seed, int threads, const uint64_t *labels = nullptr) {
    init(n, d, m, M_, efC_, bf);
    std::default_random_engine gen;
    gen.seed(seed);
    std::vector<int> lv(n);
    for (size_t i = 0; i < n; i++) {
      std::uniform_real_distribution<double> distribution(0.0, 1.0);
      double r = -log(distribution(gen)) * mult;
      lv[i] = (int)r;
    }
    if (threads < 1) threads = 1;
    Visited v0;
    size_t serial_head = threads > 1 ? std::min<size_t>(n, 1) : n;
    for (size_t i = 0; i < serial_head; i++) { count = i + 1; add_point(base + i * d, i, labels ? labels[i] : i, lv[i], v0); }
    if (serial_head < n) {
      std::atomic<size_t> next(serial_head);
      std::atomic<bool> failed(false);
      std::string err;
      std::mutex err_mu;
      count = n;
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
          Visited vl;
          while (true) {
            size_t i = next.fetch_add(1);
            if (i >= n || failed) break;
            try {
              add_point(base + i * d, i, labels ? labels[i] : i, lv[i], vl);
            } catch (std::exception &e) {
              std::lock_guard<std::mutex> g(err_mu);
              err = e.what();
              failed = true;
            }
          }
        });
      for (auto &th : pool) th.join();
      if (failed) throw std::runtime_error(err);
    }
    count = n;
  }

  void save(const std::string &path) const {  // hnswalg.h:748-779
    std::ofstream o(path, std::ios::binary);
    if (!o.is_open()) throw std::runtime_error("Cannot open file");
    put<uint64_t>(o, 0);  // offsetLevel0_
    put<uint64_t>(o, max_elements);
    put<uint64_t>(o, count);
    put<uint64_t>(o, size_per_el);
    put<uint64_t>(o, label_offset);
    put<uint64_t>(o, offsetData);
    put<int32_t>(o, maxlevel);
    put<uint32_t>(o, enterpoint);
    put<uint64_t>(o, maxM);
    put<uint64_t>(o, maxM0);
    put<uint64_t>(o, M);
    put<double>(o, mult);
    put<uint64_t>(o, efC);
    o.write(level0.data(), count * size_per_el);
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = levels[i] > 0 ? size_links_up * levels[i] : 0;
      put<uint32_t>(o, sz);
      if (sz) o.write(links[i].data(), sz);
    }
  }

  void load(const BinSource &src, Metric m, size_t d, size_t max_elements_i = 0) {  // hnswalg.h:781-893
    BinReader r(src);
    r.in.seekg(0, r.in.end);
    std::streamoff total = r.in.tellg();
    r.in.seekg(0, r.in.beg);
    uint64_t off0 = r.pod<uint64_t>();
    uint64_t file_max = r.pod<uint64_t>();
    count = r.pod<uint64_t>();
    max_elements = max_elements_i < count ? file_max : max_elements_i;
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
    metric = m; dim = d;
    size_links0 = maxM0 * 4 + 4;
    size_links_up = maxM * 4 + 4;
    if (off0 != 0 || offsetData != size_links0 || size_per_el != size_links0 + 4 * d + 8 ||
        (std::streamoff)(96 + count * size_per_el) > total)
      throw std::runtime_error("Index seems to be corrupted or unsupported");
    level0.assign(std::max<size_t>(max_elements, count) * size_per_el, 0);
    r.bytes(level0.data(), count * size_per_el);
    links.assign(std::max<size_t>(max_elements, count), {});
    levels.assign(std::max<size_t>(max_elements, count), 0);
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = r.pod<uint32_t>();
      if (sz) {
        if (sz % size_links_up) throw std::runtime_error("Index seems to be corrupted or unsupported");
        levels[i] = sz / size_links_up;
        links[i].resize(sz + 1);
        r.bytes(links[i].data(), sz);
      }
    }
    if (r.in.tellg() != total) throw std::runtime_error("Index seems to be corrupted or unsupported");  // :835-836
    locks.reset(new std::mutex[std::max<size_t>(max_elements, count)]);
  }
  size_t num_deleted() const { size_t c = 0; for (size_t i = 0; i < count; i++) c += deleted(i); return c; }
};

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// The base graph HNSW-SlimQ is converted from: rabitqlib::hnsw::HierarchicalNSW::construct
// (/root/reference/third_party/rabitqlib/index/hnsw/hnsw.hpp:667-704, add_point :706-825, search_base_layer :827-905,
// mutually_connect_new_element :907-1014, get_neighbors_by_heuristic2 :1016-1054), as include/strategy/hnsw_slimq_strategy.h:101-133
// drives it (M = 32, ef_construction = 128, seed 100).  It "builds edges with non-quantized vectors" (:696): every distance is
// get_data_dist on the RAW rows (:381-387), so this is hnswlib's insertion algorithm with three differences that change the graph:
// the heaps order (distance, id) PAIRS lexicographically (maxheap = std::priority_queue<pair>, minheap with std::greater, :33-36)
// instead of by distance alone, maxM = M and maxM0 = 2 M with mult = 1 / ln M (:445-493), and a neighbour that already lists the
// new node is not connected twice (:973-980).  Storage and file layout are VanillaGraph's (hnswalg.h:748-779), so the Slim / SlimQ
// converters read it like any vanilla index.
struct RqGraph : VanillaGraph {
  typedef std::pair<float, uint32_t> P;
  typedef std::priority_queue<P> MaxH;
  typedef std::priority_queue<P, std::vector<P>, std::greater<P>> MinH;
  float ddist(uint32_t a, uint32_t b) const { return rabitq_raw_dist(metric, vec(a), vec(b), dim); }

  MaxH rq_search_layer(uint32_t ep, uint32_t cur_c, int layer, Visited &vl) {   // :827-905
    vl.begin(max_elements);
    MaxH top;
    MinH cand;
    float lower = ddist(ep, cur_c);
    top.emplace(lower, ep);
    cand.emplace(lower, ep);
    vl.mass[ep] = vl.cur;
    while (!cand.empty()) {
      const P c = cand.top();
      if (c.first > lower && top.size() == efC) break;
      cand.pop();
      const uint32_t node = c.second;
      std::unique_lock<std::mutex> lk(locks[node]);
      const uint32_t *l = list_at(node, layer);
      const size_t n = cnt_of(l);
      for (size_t j = 0; j < n; j++) {
        const uint32_t id = l[1 + j];
        if (vl.mass[id] == vl.cur) continue;
        vl.mass[id] = vl.cur;
        const float d = ddist(id, cur_c);
        if (top.size() < efC || lower > d) {
          cand.emplace(d, id);
          top.emplace(d, id);
          if (top.size() > efC) top.pop();
          if (!top.empty()) lower = top.top().first;
        }
      }
    }
    return top;
  }
  void rq_heuristic(MaxH &top, size_t Mlim) const {   // :1016-1054
    if (top.size() < Mlim) return;
    MinH closest;
    std::vector<P> keep;
    while (!top.empty()) { closest.emplace(top.top()); top.pop(); }
    while (!closest.empty()) {
      if (keep.size() >= Mlim) break;
      const P cur = closest.top();
      closest.pop();
      bool good = true;
      for (const P &sp : keep)
        if (ddist(sp.second, cur.second) < cur.first) { good = false; break; }
      if (good) keep.push_back(cur);
    }
    for (const P &pp : keep) top.emplace(pp);
  }
  uint32_t rq_connect(uint32_t cur_c, MaxH &top, int level) {   // :907-1014
    const size_t max_m = level > 0 ? maxM : maxM0;
    rq_heuristic(top, M);
    if (top.size() > M) throw std::runtime_error("Should be not be more than M_ candidates returned by the heuristic");
    std::vector<uint32_t> sel;
    sel.reserve(M);
    while (!top.empty()) { sel.push_back(top.top().second); top.pop(); }
    const uint32_t next_ep = sel.back();
    {
      uint32_t *l = list_at(cur_c, level);
      if (*l) throw std::runtime_error("The newly inserted element should have blank link list");
      set_cnt(l, sel.size());
      for (size_t i = 0; i < sel.size(); i++) {
        if (l[1 + i]) throw std::runtime_error("Possible memory corruption");
        if (level > levels[sel[i]]) throw std::runtime_error("Trying to make a link on a non-existent level");
        l[1 + i] = sel[i];
      }
    }
    for (uint32_t nb : sel) {
      std::unique_lock<std::mutex> lk(locks[nb]);
      uint32_t *lo = list_at(nb, level);
      const size_t sz = cnt_of(lo);
      if (sz > max_m) throw std::runtime_error("Bad value of sz_link_list_other");
      if (nb == cur_c) throw std::runtime_error("Trying to connect an element to itself");
      if (level > levels[nb]) throw std::runtime_error("Trying to make a link on a non-existent level");
      uint32_t *data = lo + 1;
      bool present = false;
      for (size_t j = 0; j < sz; j++)
        if (data[j] == cur_c) { present = true; break; }
      if (present) continue;
      if (sz < max_m) {
        data[sz] = cur_c;
        set_cnt(lo, sz + 1);
      } else {
        MaxH cands;
        cands.emplace(ddist(nb, cur_c), cur_c);
        for (size_t j = 0; j < sz; j++) cands.emplace(ddist(data[j], nb), data[j]);
        rq_heuristic(cands, max_m);
        int idx = 0;
        while (!cands.empty()) { data[idx++] = cands.top().second; cands.pop(); }
        set_cnt(lo, idx);
      }
    }
    return next_ep;
  }
  void rq_add_point(const float *x, uint32_t cur_c, int curlevel, Visited &vl) {   // :706-825 (label == internal id)
    std::unique_lock<std::mutex> lock_el(locks[cur_c]);
    levels[cur_c] = curlevel;
    std::unique_lock<std::mutex> templock(global);
    const int maxlevelcopy = maxlevel;
    if (curlevel <= maxlevelcopy) templock.unlock();
    uint32_t curr = enterpoint;
    memset(el(cur_c), 0, size_per_el);
    const uint64_t label = cur_c;
    memcpy(el(cur_c) + label_offset, &label, 8);
    memcpy(el(cur_c) + offsetData, x, 4 * dim);
    if (curlevel) links[cur_c].assign(size_links_up * curlevel + 1, 0);
    if ((int32_t)curr != -1) {
      if (curlevel < maxlevelcopy) {
        float curdist = ddist(curr, cur_c);
        for (int level = maxlevelcopy; level > curlevel; level--) {
          bool changed = true;
          while (changed) {
            changed = false;
            std::unique_lock<std::mutex> lk(locks[curr]);
            const uint32_t *l = list_at(curr, level);
            const int n = cnt_of(l);
            for (int i = 0; i < n; i++) {
              const uint32_t c = l[1 + i];
              if (c > max_elements) throw std::runtime_error("cand error");
              const float d = ddist(c, cur_c);
              if (d < curdist) { curdist = d; curr = c; changed = true; }
            }
          }
        }
      }
      for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
        MaxH top = rq_search_layer(curr, cur_c, level, vl);
        curr = rq_connect(cur_c, top, level);
      }
    } else {
      enterpoint = 0;
      maxlevel = curlevel;
    }
    if (curlevel > maxlevelcopy) {
      enterpoint = cur_c;
      maxlevel = curlevel;
    }
  }
  // rows 0..n-1, labels == row index; levels drawn up front in row order from std::default_random_engine(seed) (:222, 324-328, 477),
  // so threads == 1 reproduces the reference's serial construct(); threads > 1 is as timing-dependent as the reference's own
  // parallel_for (:698-703) but keeps ids == labels (the SlimQ rerank indexes raw rows by internal id, hnswalg_slimq.h:748).
  void rq_build(const float *base, size_t n, size_t d, Metric m, size_t M_, size_t efC_, size_t seed, int threads) {
    init(n, d, m, M_, efC_, "e");
    mult = 1 / log(1.0 * (double)M);   // :493
    std::default_random_engine gen;
    gen.seed(seed);
    std::vector<int> lv(n);
    for (size_t i = 0; i < n; i++) {
      std::uniform_real_distribution<double> distribution(0.0, 1.0);
      lv[i] = (int)(-log(distribution(gen)) * mult);
    }
    if (threads < 1) threads = 1;
    // the new element's row must be in place before anyone measures against it: rows are copied by rq_add_point itself, and
    // get_data_dist(x, cur_c) reads row cur_c -> copy every row first (the reference reads rawDataPtr_, which is complete)
    for (size_t i = 0; i < n; i++) memcpy(el((uint32_t)i) + offsetData, base + i * d, 4 * d);
    Visited v0;
    const size_t serial_head = threads > 1 ? std::min<size_t>(n, 1) : n;
    auto add = [&](size_t i, Visited &vl) {
      // (rq_add_point clears the element: keep the row)
      rq_add_point(base + i * d, (uint32_t)i, lv[i], vl);
    };
    for (size_t i = 0; i < serial_head; i++) { count = i + 1; add(i, v0); }
    if (serial_head < n) {
      std::atomic<size_t> next(serial_head);
      std::atomic<bool> failed(false);
      std::string err;
      std::mutex err_mu;
      count = n;
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; t++)
        pool.emplace_back([&]() {
          Visited vl;
          while (true) {
            const size_t i = next.fetch_add(1);
            if (i >= n || failed) break;
            try { add(i, vl); }
            catch (std::exception &e) { std::lock_guard<std::mutex> g(err_mu); err = e.what(); failed = true; }
          }
        });
      for (auto &th : pool) th.join();
      if (failed) throw std::runtime_error(err);
    }
    count = n;
  }
};

struct SlimParams {
  int threshold_level = 0;
  float top_pct0 = 0.02f, top_pct = 0.02f;                 // alpha_0, alpha
  size_t top_M0 = 32, low_m0 = 8, top_M = 16, low_m = 4;   // M_h0, M_l0, M_h, M_l
};

struct SlimGraph {
  size_t count = 0, dim = 0, size_per_el = 0;
  size_t maxM = 0, maxM0 = 0, M = 0, efC = 0;
  int maxlevel = 0, threshold_level = 0;
  uint32_t enterpoint = 0;
  bool has_deleted = false;
  Metric metric = METRIC_L2;
  std::vector<char> elements;            // count * (24 + 4*dim): [i32 level][u32 total][u64 label][8B ptr][data]
  std::vector<std::vector<char>> blobs;  // [u16 cum_off[level]][u32 ids[total]]

  const char *el(uint32_t i) const { return elements.data() + (size_t)i * size_per_el; }
  int32_t level(uint32_t i) const { int32_t v; memcpy(&v, el(i), 4); return v; }
  uint32_t total(uint32_t i) const { uint32_t v; memcpy(&v, el(i) + 4, 4); return v; }
  uint64_t label(uint32_t i) const { uint64_t v; memcpy(&v, el(i) + 8, 8); return v; }
  const float *vec(uint32_t i) const { return (const float *)(el(i) + 24); }
  bool deleted(uint32_t i) const { return (((const unsigned char *)el(i))[6] & 1) != 0; }  // hnswalg_slim.h:1776-1781

  // PruneByHeuristic (hnswalg_slim.h:836-865): input sorted ascending by distance.
  // HierarchicalNSWSlimQ's own PruneByHeuristic (hnswalg_slimq.h:1334-1362), mirrored as written: the occlusion test measures the
  // candidate against the node whose INTERNAL ID is the loop index i over the sorted candidates (`hnsw->get_data_dist(i, ...)`,
  // :1349), not against the neighbours kept so far, and only once something has been kept.
  static void prune_slimq(const VanillaGraph &g, const std::vector<pairfi> &sorted, std::vector<uint32_t> &out, size_t Mlim) {
    out.clear();
    for (size_t i = 0; i < sorted.size(); i++) {
      if (out.size() >= Mlim) break;
      const pairfi &cur = sorted[i];
      bool good = true;
      if (!out.empty()) {
        float d = rabitq_raw_dist(g.metric, g.vec((uint32_t)i), g.vec(cur.second), g.dim);
        if (d < cur.first) good = false;
      }
      if (good) out.push_back(cur.second);
    }
  }
  static void prune(const VanillaGraph &g, const std::vector<pairfi> &sorted, std::vector<uint32_t> &out, size_t Mlim) {
    out.clear();
    for (const pairfi &cur : sorted) {
      if (out.size() >= Mlim) break;
      bool good = true;
      for (uint32_t kept : out) {
        float d = g.dist(g.vec(kept), g.vec(cur.second));
        if (d < cur.first) { good = false; break; }
      }
      if (good) out.push_back(cur.second);
    }
  }

  // degree histograms per level (:904-922) and hub thresholds (:923-945)
  static std::vector<size_t> hub_thresholds(const VanillaGraph &g, const SlimParams &p) {
    const int maxlevel = g.maxlevel;
    const size_t n = g.count, maxM0 = g.maxM0;
    std::vector<std::vector<size_t>> hist(maxlevel + 1, std::vector<size_t>(maxM0 + 2, 0));
    std::vector<size_t> level_cnts(maxlevel + 1, 0);
    for (size_t i = 0; i < n; i++) {
      for (int l = 1; l <= g.levels[i]; l++) { level_cnts[l]++; hist[l][VanillaGraph::cnt_of(g.list_at(i, l))]++; }
      hist[0][VanillaGraph::cnt_of(g.list_at(i, 0))]++;
    }
    std::vector<size_t> thr(maxlevel + 1, 0);
    // NB: level_cnts[0] is never incremented by the reference (:910-912 start at l=1), so topN==0
    // at level 0, `acc >= topN` holds at the first bucket and thr[0] = maxM0+1: no level-0 list is
    // ever classed as a hub (every node is pruned to M_l0).  Mirrored, not fixed.
    size_t acc = 0, topN = (size_t)(level_cnts[0] * p.top_pct0 + 0.5);
    for (size_t d = hist[0].size() - 1; d > 0; --d) { acc += hist[0][d]; if (acc >= topN) { thr[0] = d; break; } }
    for (int l = 1; l <= maxlevel; l++) {
      acc = 0; topN = (size_t)(level_cnts[l] * p.top_pct + 0.5);
      for (size_t d = hist[l].size() - 1; d > 0; --d) { acc += hist[l][d]; if (acc >= topN) { thr[l] = d; break; } }
    }
    return thr;
  }
  void take_header(const VanillaGraph &g, const SlimParams &p) {
    count = g.count; dim = g.dim; metric = g.metric;
    has_deleted = g.num_deleted() > 0;
    maxM = g.maxM; maxM0 = g.maxM0; M = g.M; efC = g.efC;
    maxlevel = g.maxlevel; enterpoint = g.enterpoint; threshold_level = p.threshold_level;
    size_per_el = 24 + 4 * dim;
  }
  // Element i from its final per-level lists (after the re-prune): hierarchical filter (:1063-1084), offsets, blob (:1085-1106).
  template <class GetList>
  void assemble_node(const VanillaGraph &g, size_t i, GetList list_of) {
    char *e = elements.data() + i * size_per_el;
    int32_t L = g.levels[i];
    memcpy(e, &L, 4);
    memcpy(e + 24, g.vec(i), 4 * dim);
    uint64_t lab = g.label(i);
    memcpy(e + 8, &lab, 8);
    std::vector<uint32_t> nbrs_out;
    std::vector<uint16_t> offs;
    for (int l = 0; l <= L; l++) {
      size_t cnt = 0;
      const uint32_t *nbrs = list_of(i, l, cnt);
      if (l == threshold_level) {
        nbrs_out.insert(nbrs_out.end(), nbrs, nbrs + cnt);
      } else {  // hierarchical pruning: keep neighbours whose own top level == l (:1072-1083)
        for (size_t j = 0; j < cnt; j++)
          if (g.levels[nbrs[j]] == l) nbrs_out.push_back(nbrs[j]);
      }
      offs.push_back((uint16_t)nbrs_out.size());
    }
    uint32_t total = nbrs_out.size();
    memcpy(e + 4, &total, 4);
    if (total == 0) return;  // neighbors pointer stays null (:1091-1094)
    blobs[i].resize(2 * (size_t)L + 4 * (size_t)total);
    memcpy(blobs[i].data(), offs.data(), 2 * (size_t)L);
    memcpy(blobs[i].data() + 2 * (size_t)L, nbrs_out.data(), 4 * (size_t)total);
  }

  // convertFromHNSW (hnswalg_slim.h:867-1108); slimq_prune: HierarchicalNSWSlimQ::convertFromHNSW's graph part
  // (hnswalg_slimq.h:1471-1762: the same passes with its own PruneByHeuristic).
  void convert(const VanillaGraph &g, const SlimParams &p, int threads, bool slimq_prune = false) {
    auto prune = [&](const VanillaGraph &gg, const std::vector<pairfi> &sorted, std::vector<uint32_t> &out, size_t Mlim) {
      if (slimq_prune) SlimGraph::prune_slimq(gg, sorted, out, Mlim);
      else SlimGraph::prune(gg, sorted, out, Mlim);
    };
    // distances: hnsw->fstdistfunc_ (hnswalg_slim.h:977-979) / hnsw->get_data_dist (hnswalg_slimq.h:1623, 1706)
    auto D = [&](uint32_t a, uint32_t b) {
      return slimq_prune ? rabitq_raw_dist(g.metric, g.vec(a), g.vec(b), g.dim) : g.dist(g.vec(a), g.vec(b));
    };
    take_header(g, p);
    const size_t n = count;
    const std::vector<size_t> thr = hub_thresholds(g, p);
    std::vector<std::vector<std::vector<uint32_t>>> nn(n), rev(n);
    auto par = [&](auto fn) {
      int T = std::max(1, threads);
      std::atomic<size_t> next(0);
      std::vector<std::thread> pool;
      for (int t = 0; t < T; t++)
        pool.emplace_back([&]() { for (size_t v; (v = next.fetch_add(64)) < n;) for (size_t u = v; u < std::min(n, v + 64); u++) fn(u); });
      for (auto &th : pool) th.join();
    };
    par([&](size_t v) {  // :951-986
      int L = g.levels[v];
      nn[v].resize(L + 1); rev[v].resize(L + 1);
      std::vector<pairfi> heap;
      for (int l = 0; l <= L; l++) {
        const uint32_t *ll = g.list_at(v, l);
        size_t size = VanillaGraph::cnt_of(ll);
        size_t M0 = l == 0 ? (size > thr[l] ? p.top_M0 : p.low_m0) : (size > thr[l] ? p.top_M : p.low_m);
        heap.resize(size);
        for (size_t j = 0; j < size; j++) heap[j] = {D(v, ll[1 + j]), ll[1 + j]};
        std::sort(heap.begin(), heap.end(), CmpFirst());
        prune(g, heap, nn[v][l], M0);
      }
    });
    for (size_t v = 0; v < n; v++)  // reverse edges (:988-998); serial: order of emplace_back is irrelevant after the sort below
      for (int l = 0; l <= g.levels[v]; l++)
        for (uint32_t u : nn[v][l]) rev[u][l].push_back(v);
    par([&](size_t v) {  // union + sort + unique (:999-1012)
      for (int l = 0; l <= g.levels[v]; l++) {
        auto &a = nn[v][l];
        a.insert(a.end(), rev[v][l].begin(), rev[v][l].end());
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
      }
    });
    elements.assign(n * size_per_el, 0);
    blobs.assign(n, {});
    par([&](size_t i) {  // :1014-1107
      std::vector<pairfi> heap;
      for (int l = 0; l <= g.levels[i]; l++) {
        auto &nbrs = nn[i][l];
        size_t limit = l == 0 ? maxM0 : maxM;
        if (nbrs.size() > limit) {  // re-prune (:1038-1062)
          heap.resize(nbrs.size());
          for (size_t j = 0; j < nbrs.size(); j++) heap[j] = {D((uint32_t)i, nbrs[j]), nbrs[j]};
          std::sort(heap.begin(), heap.end(), CmpFirst());
          prune(g, heap, nbrs, limit);
        }
      }
      assemble_node(g, i, [&](size_t v, int l, size_t &cnt) { cnt = nn[v][l].size(); return nn[v][l].data(); });
    });
  }

#ifdef HS_HAVE_GPU_CONVERT
  // The same conversion with the list-level work on the GPU (convert_gpu.hip).  Returns false when the shape is outside the
  // device path (lists longer than 64 / capacities above 32 ids, or a reverse-edge list that outgrew the on-chip buffers):
  // the caller then runs convert().  kernel_ms: device time of the kernels.
  bool convert_gpu(const VanillaGraph &g, const SlimParams &p, int device, int threads, double *kernel_ms, std::string *err) {
    if (g.maxM0 > 32 || g.maxM > 32 || p.top_M0 > 32 || p.low_m0 > 32 || p.top_M > 32 || p.low_m > 32) return false;
    take_header(g, p);
    const size_t n = count;
    const std::vector<size_t> thr = hub_thresholds(g, p);
    ConvertInput in;
    in.vec = nullptr; in.n = (uint32_t)n; in.dim = (uint32_t)dim; in.metric = (int)metric;
    std::vector<float> rows(n * dim);
    for (size_t i = 0; i < n; i++) memcpy(&rows[i * dim], g.vec(i), 4 * dim);
    in.vec = rows.data();
    in.upb.assign(n, 0);
    size_t nup = 0;
    for (size_t i = 0; i < n; i++) { in.upb[i] = (uint32_t)nup; nup += g.levels[i]; }
    const size_t nt = n + nup;
    in.t_node.resize(nt); in.t_level.resize(nt); in.t_off.resize(nt); in.t_size.resize(nt); in.t_mlim.resize(nt); in.t_limit.resize(nt);
    auto add = [&](size_t t, size_t v, int l) {
      const uint32_t *ll = g.list_at(v, l);
      const size_t size = VanillaGraph::cnt_of(ll);
      in.t_node[t] = (uint32_t)v; in.t_level[t] = (uint32_t)l; in.t_off[t] = (uint32_t)in.lists.size(); in.t_size[t] = (uint32_t)size;
      in.t_mlim[t] = (uint32_t)(l == 0 ? (size > thr[l] ? p.top_M0 : p.low_m0) : (size > thr[l] ? p.top_M : p.low_m));   // :957-961, 969-973
      in.t_limit[t] = (uint32_t)(l == 0 ? maxM0 : maxM);
      in.lists.insert(in.lists.end(), ll + 1, ll + 1 + size);
    };
    in.lists.reserve(n * 16);
    for (size_t v = 0; v < n; v++) add(v, v, 0);
    for (size_t v = 0; v < n; v++)
      for (int l = 1; l <= g.levels[v]; l++) add(n + in.upb[v] + l - 1, v, l);
    for (size_t t = 0; t < nt; t++)
      if (in.t_size[t] > 64) return false;
    // what the device kernels index with: every list entry a node of at least the list's level (the reverse-edge kernels address
    // task n + upb[u] + l - 1 for a level-l edge to u), 32-bit prefix sums.  A file that breaks this takes the CPU conversion.
    if (nt >= 0xFFFFFFFFull || in.lists.size() >= 0xFFFFFFFFull || 2 * in.lists.size() >= 0xFFFFFFFFull) return false;
    for (size_t t = 0; t < nt; t++) {
      const uint32_t l = in.t_level[t];
      for (uint32_t j = 0; j < in.t_size[t]; j++) {
        const uint32_t u = in.lists[in.t_off[t] + j];
        if (u >= n || (uint32_t)g.levels[u] < l) return false;
      }
    }
    std::vector<uint32_t> fin, fin_cnt;
    bool needs_host = false;
    hipError_t e = gpu_convert_lists(in, device, fin, fin_cnt, needs_host, kernel_ms);
    if (e != hipSuccess) {
      if (err) *err = std::string("GPU convert: ") + hipGetErrorString(e);
      return false;
    }
    if (needs_host) return false;
    elements.assign(n * size_per_el, 0);
    blobs.assign(n, {});
    int T = std::max(1, threads);
    std::atomic<size_t> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++)
      pool.emplace_back([&]() {
        for (size_t v0; (v0 = next.fetch_add(256)) < n;)
          for (size_t i = v0; i < std::min(n, v0 + 256); i++)
            assemble_node(g, i, [&](size_t v, int l, size_t &cnt) {
              const size_t t2 = l == 0 ? v : n + in.upb[v] + l - 1;
              cnt = fin_cnt[t2];
              return fin.data() + t2 * 32;
            });
      });
    for (auto &th : pool) th.join();
    return true;
  }
#endif

  void save(const std::string &path) const {  // hnswalg_slim.h:717-751
    std::ofstream o(path, std::ios::binary);
    if (!o.is_open()) throw std::runtime_error("Cannot open file");
    put<uint64_t>(o, count); put<uint64_t>(o, size_per_el);
    put<uint64_t>(o, 8); put<uint64_t>(o, 4); put<uint64_t>(o, 24); put<uint64_t>(o, 16);
    put<int32_t>(o, maxlevel); put<int32_t>(o, threshold_level); put<uint32_t>(o, enterpoint);
    put<uint64_t>(o, maxM); put<uint64_t>(o, maxM0); put<uint64_t>(o, M); put<uint64_t>(o, efC);
    put<uint8_t>(o, has_deleted ? 1 : 0);
    o.write(elements.data(), count * size_per_el);
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = 2 * (uint32_t)level(i) + 4 * total(i);  // get_neighbor_size (:652-661)
      put<uint32_t>(o, sz);
      if (sz && total(i) != 0) o.write(blobs[i].data(), sz);
    }
  }

  void load(const BinSource &src, Metric m, size_t d) {  // hnswalg_slim.h:753-815
    BinReader r(src);
    metric = m; dim = d;
    count = r.pod<uint64_t>();
    size_per_el = r.pod<uint64_t>();
    uint64_t label_off = r.pod<uint64_t>(), off_total = r.pod<uint64_t>(), off_data = r.pod<uint64_t>(), off_nb = r.pod<uint64_t>();
    maxlevel = r.pod<int32_t>();
    threshold_level = r.pod<int32_t>();
    enterpoint = r.pod<uint32_t>();
    maxM = r.pod<uint64_t>(); maxM0 = r.pod<uint64_t>(); M = r.pod<uint64_t>(); efC = r.pod<uint64_t>();
    has_deleted = r.pod<uint8_t>() != 0;
    if (label_off != 8 || off_total != 4 || off_data != 24 || off_nb != 16 || size_per_el != 24 + 4 * d)
      throw std::runtime_error("Index seems to be corrupted or unsupported");
    elements.resize(count * size_per_el);
    r.bytes(elements.data(), elements.size());
    blobs.assign(count, {});
    for (size_t i = 0; i < count; i++) {
      uint32_t sz = r.pod<uint32_t>();
      if (sz == 0 || total(i) == 0) continue;
      if (sz != 2 * (uint32_t)level(i) + 4 * total(i)) throw std::runtime_error("Index seems to be corrupted or unsupported");
      blobs[i].resize(sz);
      r.bytes(blobs[i].data(), sz);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Device-facing packed form.  Level-0 adjacency is plain CSR (row_ptr0[n+1], cols); nodes with upper
// levels own (level+1) consecutive entries of up_ptr starting at up_base[i]: the level-l slice
// (l >= 1) is cols[up_ptr[up_base[i]+l-1] .. up_ptr[up_base[i]+l]).
struct PackedIndex {
  int kind = 0;  // 0 = HierarchicalNSW, 1 = HierarchicalNSWSlim
  Metric metric = METRIC_L2;
  size_t n = 0, dim = 0;
  int maxlevel = 0, threshold_level = 0;
  uint32_t enterpoint = 0;
  bool has_deleted = false;
  size_t max_deg0 = 0;
  size_t index_size = 0;   // what the reference's indexSize() reports for this index (graph structure without vectors)
  std::vector<float> vec;
  std::vector<uint32_t> row_ptr0, cols, up_base, up_ptr;
  std::vector<uint64_t> labels;
  std::vector<uint8_t> deleted;

  static constexpr uint32_t NONE = 0xFFFFFFFFu;

  void from_vanilla(const VanillaGraph &g) {
    kind = 0; metric = g.metric; n = g.count; dim = g.dim;
    // HierarchicalNSW::indexSize() (hnswalg.h:1533-1547): level-0 block without vectors and labels, element_levels_,
    // one pointer per element + its upper link lists (+1 as malloc'd)
    index_size = g.max_elements * (g.size_per_el - g.dim * 4 - 8) + g.max_elements * sizeof(int);
    for (size_t i = 0; i < g.count; i++) index_size += 8 + (g.levels[i] > 0 ? g.size_links_up * (size_t)g.levels[i] + 1 : 0);
    maxlevel = g.maxlevel; threshold_level = 0; enterpoint = g.enterpoint;
    vec.resize(n * dim); labels.resize(n); deleted.resize(n);
    row_ptr0.assign(n + 1, 0); up_base.assign(n, NONE); up_ptr.clear(); cols.clear();
    size_t nd = 0;
    for (size_t i = 0; i < n; i++) {
      memcpy(&vec[i * dim], g.vec(i), 4 * dim);
      labels[i] = g.label(i);
      deleted[i] = g.deleted(i);
      nd += deleted[i];
      const uint32_t *l = g.list_at(i, 0);
      size_t c = VanillaGraph::cnt_of(l);
      max_deg0 = std::max(max_deg0, c);
      cols.insert(cols.end(), l + 1, l + 1 + c);
      row_ptr0[i + 1] = cols.size();
    }
    has_deleted = nd > 0;
    for (size_t i = 0; i < n; i++) {
      int L = g.levels[i];
      if (L <= 0) continue;
      up_base[i] = up_ptr.size();
      for (int l = 1; l <= L; l++) {
        up_ptr.push_back(cols.size());
        const uint32_t *ll = g.list_at(i, l);
        cols.insert(cols.end(), ll + 1, ll + 1 + VanillaGraph::cnt_of(ll));
      }
      up_ptr.push_back(cols.size());
    }
    if (cols.size() >= NONE) throw std::runtime_error("adjacency too large for 32-bit CSR offsets");
  }

  // CHAL blobs ([u16 cumulative offsets x level][u32 ids x total], hnswalg_slim.h:1981-2000) -> CSR.
  // level_of(i), total_of(i), blob_of(i) -> const std::vector<char>&
  template <class LF, class TF, class BF>
  void pack_chal(LF level_of, TF total_of, BF blob_of) {
    row_ptr0.assign(n + 1, 0); up_base.assign(n, NONE); up_ptr.clear(); cols.clear();
    auto slice = [&](size_t i, int lvl, const uint32_t *&ids, size_t &cnt) {
      const int L = level_of(i);
      const std::vector<char> &blob = blob_of(i);
      const uint16_t *off = (const uint16_t *)blob.data();
      size_t s = lvl == 0 ? 0 : off[lvl - 1];
      size_t e = lvl == L ? total_of(i) : off[lvl];
      if (e < s || e > total_of(i) || blob.size() < 2 * (size_t)L + 4 * (size_t)total_of(i))
        throw std::runtime_error("Index seems to be corrupted or unsupported");
      ids = (const uint32_t *)(blob.data() + 2 * (size_t)L) + s;
      cnt = e - s;
    };
    for (size_t i = 0; i < n; i++) {
      if (!blob_of(i).empty()) {
        const uint32_t *ids; size_t c;
        slice(i, 0, ids, c);
        max_deg0 = std::max(max_deg0, c);
        cols.insert(cols.end(), ids, ids + c);
      }
      row_ptr0[i + 1] = cols.size();
    }
    for (size_t i = 0; i < n; i++) {
      const int L = level_of(i);
      if (L <= 0 || blob_of(i).empty()) continue;
      up_base[i] = up_ptr.size();
      for (int l = 1; l <= L; l++) {
        up_ptr.push_back(cols.size());
        const uint32_t *ids; size_t c;
        slice(i, l, ids, c);
        cols.insert(cols.end(), ids, ids + c);
      }
      up_ptr.push_back(cols.size());
    }
    for (uint32_t c : cols) if (c >= n) throw std::runtime_error("Index seems to be corrupted or unsupported");
    if (cols.size() >= NONE) throw std::runtime_error("adjacency too large for 32-bit CSR offsets");
  }

  void from_slim(const SlimGraph &g) {
    kind = 1; metric = g.metric; n = g.count; dim = g.dim;
    // HierarchicalNSWSlim::indexSize() (hnswalg_slim.h:2435-2444): 16 bytes per element + every neighbour blob
    index_size = g.count * 16;
    for (size_t i = 0; i < g.count; i++) index_size += 2 * (size_t)g.level((uint32_t)i) + 4 * (size_t)g.total((uint32_t)i);
    maxlevel = g.maxlevel; threshold_level = g.threshold_level; enterpoint = g.enterpoint;
    has_deleted = g.has_deleted;
    vec.resize(n * dim); labels.resize(n); deleted.resize(n);
    for (size_t i = 0; i < n; i++) {
      memcpy(&vec[i * dim], g.vec(i), 4 * dim);
      labels[i] = g.label(i);
      deleted[i] = g.deleted(i);
    }
    pack_chal([&](size_t i) { return (int)g.level(i); }, [&](size_t i) { return (size_t)g.total(i); },
              [&](size_t i) -> const std::vector<char> & { return g.blobs[i]; });
  }
};

}  // namespace hs
