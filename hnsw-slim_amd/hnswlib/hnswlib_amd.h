// hnswlib_amd.h -- header-only C++ facade that re-creates the part of the reference's hnswlib API that
// sits on the searchKnn hot path, on top of the C ABI in include/hnsw_slim_amd.h (libhnsw_slim_amd.so).
//
// A caller written against the reference (include/strategy/hnsw_slim_strategy.h:38-114,
// hnsw_slim_server.cc:59-81) keeps its code: same namespace, class names, method names, argument meaning,
// exception type and messages.  What changes is where the work runs: loadIndex uploads the index to one
// MI355X, searchKnn runs the HIP beam-search kernel, and the new searchKnnBatch() is the fast entry
// (one wavefront per query; a single-query searchKnn is a batch of one and pays a kernel launch).
//
// Mirrors (paths relative to /root/reference/third_party/hnswlib/):
//   hnswlib.h:125-221        labeltype, BaseFilterFunctor, DISTFUNC, SpaceInterface, AlgorithmInterface
//   space_l2.h:208-253       L2Space            space_ip.h:342-398  InnerProductSpace
//   hnswalg.h:17-19,78-83,184,781,1378          HierarchicalNSW<float>
//   hnswalg_slim.h:28-30,83-87,149-152,193,753,867,1907,2030   HierarchicalNSWSlim<float>
// Filter functors are host callbacks: the facade evaluates one once per element into an allowed-array
// (cached per functor object) and calls hs_search_batch_filtered.
// Not provided (outside the search path): addPoint/updatePoint/markDelete, the diff/patch
// protocol, stop conditions.
#pragma once
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <unistd.h>

#include "../../include/hnsw_slim_amd.h"
#include "../csrc/dist_recipe.hpp"

namespace hnswlib {

typedef size_t labeltype;
typedef unsigned int tableint;

class BaseFilterFunctor {
 public:
  virtual bool operator()(hnswlib::labeltype) { return true; }
  virtual ~BaseFilterFunctor() {}
};

template <typename MTYPE>
using DISTFUNC = MTYPE (*)(const void *, const void *, const void *);

template <typename MTYPE>
class SpaceInterface {
 public:
  virtual size_t get_data_size() = 0;
  virtual DISTFUNC<MTYPE> get_dist_func() = 0;
  virtual void *get_dist_func_param() = 0;
  virtual ~SpaceInterface() {}
};

// The host-side distance functions are the same operation-for-operation recipes the kernel uses
// (dist_recipe.hpp), so values computed through the space agree bit for bit with search results.
class L2Space : public SpaceInterface<float> {
  size_t data_size_, dim_;
  static float fn(const void *a, const void *b, const void *p) {
    return hs::host_dist(hs::METRIC_L2, (const float *)a, (const float *)b, *(const size_t *)p);  // every dim
  }
 public:
  explicit L2Space(size_t dim) : data_size_(dim * sizeof(float)), dim_(dim) {}
  size_t get_data_size() override { return data_size_; }
  DISTFUNC<float> get_dist_func() override { return fn; }
  void *get_dist_func_param() override { return &dim_; }
};
class InnerProductSpace : public SpaceInterface<float> {
  size_t data_size_, dim_;
  static float fn(const void *a, const void *b, const void *p) {
    return hs::host_dist(hs::METRIC_IP, (const float *)a, (const float *)b, *(const size_t *)p);  // every dim
  }
 public:
  explicit InnerProductSpace(size_t dim) : data_size_(dim * sizeof(float)), dim_(dim) {}
  size_t get_data_size() override { return data_size_; }
  DISTFUNC<float> get_dist_func() override { return fn; }
  void *get_dist_func_param() override { return &dim_; }
};

template <typename dist_t>
class AlgorithmInterface {
 public:
  virtual void addPoint(const void *datapoint, labeltype label, bool replace_deleted = false) = 0;
  virtual std::priority_queue<std::pair<dist_t, labeltype>> searchKnn(const void *, size_t,
                                                                     BaseFilterFunctor *isIdAllowed = nullptr) const = 0;
  virtual std::vector<std::pair<dist_t, labeltype>> searchKnnCloserFirst(const void *query_data, size_t k,
                                                                        BaseFilterFunctor *isIdAllowed = nullptr) const {
    std::vector<std::pair<dist_t, labeltype>> result;  // hnswlib.h:203-221
    auto ret = searchKnn(query_data, k, isIdAllowed);
    size_t sz = ret.size();
    result.resize(sz);
    while (!ret.empty()) {
      result[--sz] = ret.top();
      ret.pop();
    }
    return result;
  }
  virtual void saveIndex(const std::string &location) = 0;
  virtual ~AlgorithmInterface() {}
};

namespace detail {
inline void check(hs_status s) {
  if (s != HS_OK) throw std::runtime_error(hs_last_error());  // same exception type/messages as the reference
}
inline int metric_of(SpaceInterface<float> *s) { return dynamic_cast<InnerProductSpace *>(s) ? HS_METRIC_IP : HS_METRIC_L2; }
inline size_t dim_of(SpaceInterface<float> *s) { return s->get_data_size() / sizeof(float); }

// Common device-index holder.
class DeviceIndex {
 protected:
  hs_index *h_ = nullptr;
  std::string path_;
  int metric_ = HS_METRIC_L2;
  size_t dim_ = 0;
  int device_ = 0;
  // more than one device: the index is replicated and searchKnnBatch shards the batch (hs_search_batch_sharded);
  // replicas_[0] == h_.  Single-query searchKnn stays on the first device.
  std::vector<int> devices_;
  std::vector<hs_index *> replicas_;
  hs_comm *comm_ = nullptr;
  void free_replicas() {
    for (size_t r = 1; r < replicas_.size(); r++) hs_index_free(replicas_[r]);
    replicas_.clear();
    hs_comm_free(comm_);
    comm_ = nullptr;
  }

 public:
  size_t ef_ = 10;
  ~DeviceIndex() {
    free_replicas();
    hs_index_free(h_);
  }
  void setDevice(int device) { device_ = device; devices_.clear(); }
  // Call before loadIndex: replicate the index on these devices (e.g. {0,1,...,7} on one MI355X node).
  void setDevices(const std::vector<int> &devices) {
    devices_ = devices;
    if (!devices.empty()) device_ = devices[0];
  }
  void setEf(size_t ef) {
    ef_ = ef;
    if (h_) check(hs_set_ef(h_, ef));
    for (size_t r = 1; r < replicas_.size(); r++) check(hs_set_ef(replicas_[r], ef));
  }
  bool sharded() const { return comm_ != nullptr; }
  hs_index *handle() const { return h_; }
  const std::string &path() const { return path_; }
  // indexSize(): the bytes the reference reports for this index's graph structure in ITS layout (hnswalg.h:1533-1547,
  // hnswalg_slim.h:2435-2444, hnswalg_slimq.h:2047-2057), so that the strategies' "index size" lines keep their meaning
  // (hnsw_slim_strategy.h:83,99); deviceBytes() is what the index holds in HBM here.
  size_t indexSize() const {
    if (!h_) return 0;
    hs_info info;
    check(hs_index_info(h_, &info));
    return (size_t)info.index_size;
  }
  size_t deviceBytes() const {
    if (!h_) return 0;
    hs_info info;
    check(hs_index_info(h_, &info));
    return (size_t)info.device_bytes;
  }

 protected:
  void load(const std::string &location, int kind, SpaceInterface<float> *s, size_t max_elements) {
    hs_index_free(h_);
    h_ = nullptr;
    metric_ = metric_of(s);
    dim_ = dim_of(s);
    path_ = location;
    free_replicas();
    check(hs_index_load(location.c_str(), kind, metric_, dim_, max_elements, device_, &h_));
    if (devices_.size() > 1) {
      replicas_.assign(1, h_);
      for (size_t r = 1; r < devices_.size(); r++) {
        hs_index *h = nullptr;
        check(hs_index_load(location.c_str(), kind, metric_, dim_, max_elements, devices_[r], &h));
        replicas_.push_back(h);
      }
      check(hs_comm_init((int)devices_.size(), devices_.data(), &comm_));
    }
    ef_ = 10;  // hnswalg.h:864, hnswalg_slim.h:793
  }
  mutable BaseFilterFunctor *cached_filter_ = nullptr;
  mutable std::vector<uint8_t> allowed_;
  std::priority_queue<std::pair<float, labeltype>> search_pq(const void *q, size_t k, BaseFilterFunctor *f = nullptr) const {
    std::priority_queue<std::pair<float, labeltype>> result;
    if (!h_) return result;
    std::vector<uint64_t> labels(k);
    std::vector<float> dists(k);
    uint32_t cnt = 0;
    if (f) {
      if (f != cached_filter_) {  // evaluate the functor once per element (hs_labels: label of each internal id)
        hs_info info;
        check(hs_index_info(h_, &info));
        std::vector<uint64_t> all(info.n);
        check(hs_labels(h_, all.data()));
        allowed_.resize(info.n);
        for (size_t i = 0; i < info.n; i++) allowed_[i] = (*f)((labeltype)all[i]) ? 1 : 0;
        cached_filter_ = f;
      }
      check(hs_search_batch_filtered(h_, (const float *)q, 1, k, allowed_.data(), labels.data(), dists.data(), &cnt, nullptr));
    } else
    check(hs_search_batch(h_, (const float *)q, 1, k, HS_MODE_PQ, nullptr, labels.data(), dists.data(), &cnt, nullptr));
    for (uint32_t i = 0; i < cnt; i++) result.emplace(dists[i], (labeltype)labels[i]);
    return result;
  }
};
}  // namespace detail

template <typename dist_t>
class HierarchicalNSW;

template <>
class HierarchicalNSW<float> : public AlgorithmInterface<float>, public detail::DeviceIndex {
  // build-then-search callers (include/strategy/hnsw_strategy.h:24-40: ctor, addPoint loop, saveIndex, setEf, searchKnn):
  // points are collected on the host, the graph is built by the CPU harness (hs_build_hnsw_labeled: the reference's
  // addPoint loop, serial => the same bytes) the first time it is needed, written to `saveIndex`'s path (or a temporary
  // file) and loaded onto the device.
  SpaceInterface<float> *space_ = nullptr;
  size_t max_elements_ = 0, M_ = 16, efc_ = 200, seed_ = 100;
  std::string branching_ = "16";
  std::vector<float> rows_;
  std::vector<uint64_t> row_labels_;
  bool dirty_ = false;
  int build_threads_ = 1;
  std::string tmp_path_;   // index file of a graph that was built here and never saved by the caller
  void materialize(const std::string &location) {
    if (row_labels_.empty()) throw std::runtime_error("hnswlib_amd: nothing to build (no addPoint calls)");
    detail::check(hs_build_hnsw_labeled(rows_.data(), row_labels_.data(), row_labels_.size(), detail::dim_of(space_), detail::metric_of(space_),
                                        M_, efc_, branching_.c_str(), seed_, build_threads_, location.c_str()));
    const size_t ef_keep = ef_;
    load(location, HS_KIND_HNSW, space_, max_elements_);
    setEf(ef_keep);
    dirty_ = false;
  }
  void ensure_built() const {
    if (!dirty_) return;
    char tmpl[] = "/tmp/hnswlib_amd_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) throw std::runtime_error("Cannot open file");
    close(fd);
    auto *self = const_cast<HierarchicalNSW *>(this);
    if (!self->tmp_path_.empty()) unlink(self->tmp_path_.c_str());
    self->tmp_path_ = tmpl;
    self->materialize(tmpl);
  }

 public:
  ~HierarchicalNSW() { if (!tmp_path_.empty()) unlink(tmp_path_.c_str()); }
  void build() const { ensure_built(); }   // force the deferred build (convertFromHNSW calls it)
  size_t indexSize() const { ensure_built(); return detail::DeviceIndex::indexSize(); }
  explicit HierarchicalNSW(SpaceInterface<float> *s) : space_(s) {}
  HierarchicalNSW(SpaceInterface<float> *s, const std::string &location, bool /*nmslib*/ = false, size_t max_elements = 0,
                  bool /*allow_replace_deleted*/ = false) : space_(s) {
    loadIndex(location, s, max_elements);
  }
  // hnswalg.h:85-159
  HierarchicalNSW(SpaceInterface<float> *s, size_t max_elements, size_t M = 16, size_t ef_construction = 200,
                  std::string branching_factor = "16", size_t random_seed = 100, bool /*allow_replace_deleted*/ = false)
      : space_(s), max_elements_(max_elements), M_(M), efc_(ef_construction), seed_(random_seed), branching_(branching_factor) {
    rows_.reserve(max_elements * detail::dim_of(s));
    row_labels_.reserve(max_elements);
  }
  void setBuildThreads(int t) { build_threads_ = t < 1 ? 1 : t; }   // 1 = the reference's serial addPoint loop, byte for byte
  void loadIndex(const std::string &location, SpaceInterface<float> *s, size_t max_elements_i = 0) {
    space_ = s;
    rows_.clear(); row_labels_.clear(); dirty_ = false;
    load(location, HS_KIND_HNSW, s, max_elements_i);
  }
  void addPoint(const void *datapoint, labeltype label, bool = false) override {
    if (!space_ || max_elements_ == 0)
      throw std::runtime_error("hnswlib_amd: addPoint needs the (space, max_elements, M, ef_construction, ..) constructor");
    if (row_labels_.size() >= max_elements_)
      throw std::runtime_error("The number of elements exceeds the specified limit");  // hnswalg.h:1289-1291
    const size_t d = detail::dim_of(space_);
    rows_.insert(rows_.end(), (const float *)datapoint, (const float *)datapoint + d);
    row_labels_.push_back(label);
    dirty_ = true;
  }
  void saveIndex(const std::string &location) override {
    if (row_labels_.empty()) throw std::runtime_error("hnswlib_amd: the device index is read-only; the file it was loaded from is unchanged");
    materialize(location);
  }
  std::priority_queue<std::pair<float, labeltype>> searchKnn(const void *query_data, size_t k,
                                                             BaseFilterFunctor *isIdAllowed = nullptr) const override {
    ensure_built();
    return search_pq(query_data, k, isIdAllowed);
  }
  // Batched searchKnn: nq x dim queries; out_labels / out_dists nq x k (unused slots: UINT64_MAX / +inf).
  void searchKnnBatch(const float *queries, size_t nq, size_t k, uint64_t *out_labels, float *out_dists,
                      uint32_t *out_counts) const {
    ensure_built();
    if (comm_) detail::check(hs_search_batch_sharded(comm_, replicas_.data(), queries, nq, k, HS_MODE_PQ, nullptr, out_labels, out_dists, out_counts));
    else detail::check(hs_search_batch(h_, queries, nq, k, HS_MODE_PQ, nullptr, out_labels, out_dists, out_counts, nullptr));
  }
};

template <typename dist_t>
class HierarchicalNSWSlim;

template <>
class HierarchicalNSWSlim<float> : public AlgorithmInterface<float>, public detail::DeviceIndex {
  SpaceInterface<float> *space_ = nullptr;
  int thr_ = 0;
  float p0_ = 0.02f, p_ = 0.02f;
  size_t M0h_ = 32, m0l_ = 8, Mh_ = 16, ml_ = 4;
  std::string tmp_path_;

 public:
  explicit HierarchicalNSWSlim(SpaceInterface<float> *s) : space_(s) {}
  HierarchicalNSWSlim(SpaceInterface<float> *s, const std::string &location, bool /*nmslib*/ = false,
                      size_t max_elements = 0, bool /*allow_replace_deleted*/ = false) : space_(s) {
    loadIndex(location, s, max_elements);
  }
  // hnswalg_slim.h:89-143: the pruning parameters live in the object, convertFromHNSW(hnsw) uses them
  HierarchicalNSWSlim(SpaceInterface<float> *s, size_t /*max_elements*/, size_t /*M*/ = 16, size_t /*ef_construction*/ = 200,
                      size_t threshold_level = 0, float top_degree_percent0 = 0.02f, float top_degree_percent = 0.02f,
                      size_t top_degree_M0 = 32, size_t low_degree_m0 = 8, size_t top_degree_M = 16, size_t low_degree_m = 4,
                      size_t /*random_seed*/ = 100, bool /*allow_replace_deleted*/ = false)
      : space_(s), thr_((int)threshold_level), p0_(top_degree_percent0), p_(top_degree_percent), M0h_(top_degree_M0),
        m0l_(low_degree_m0), Mh_(top_degree_M), ml_(low_degree_m) {}
  ~HierarchicalNSWSlim() { if (!tmp_path_.empty()) unlink(tmp_path_.c_str()); }
  void loadIndex(const std::string &location, SpaceInterface<float> *s, size_t max_elements_i = 0) {
    space_ = s;
    load(location, HS_KIND_SLIM, s, max_elements_i);
  }
  // convertFromHNSW(hnsw) (hnswalg_slim.h:867-1108) with the object's parameters; the Slim file goes to a temporary
  // path until saveIndex names one
  void convertFromHNSW(HierarchicalNSW<float> *hnsw, int threads = 1) {
    hnsw->build();
    char tmpl[] = "/tmp/hnswlib_amd_slim_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) throw std::runtime_error("Cannot open file");
    close(fd);
    if (!tmp_path_.empty()) unlink(tmp_path_.c_str());
    tmp_path_ = tmpl;
    convertFromHNSW(hnsw, space_, tmp_path_, thr_, p0_, p_, M0h_, m0l_, Mh_, ml_, threads);
  }
  // convertFromHNSW + saveIndex to `slim_location`, then load it on the device.
  void convertFromHNSW(HierarchicalNSW<float> *hnsw, SpaceInterface<float> *s, const std::string &slim_location,
                       int threshold_level = 0, float top_degree_percent0 = 0.02f, float top_degree_percent = 0.02f,
                       size_t top_degree_M0 = 32, size_t low_degree_m0 = 8, size_t top_degree_M = 16,
                       size_t low_degree_m = 4, int threads = 1) {
    hnsw->build();
    detail::check(hs_convert_slim(hnsw->path().c_str(), detail::metric_of(s), detail::dim_of(s), threshold_level,
                                  top_degree_percent0, top_degree_percent, top_degree_M0, low_degree_m0, top_degree_M,
                                  low_degree_m, threads, slim_location.c_str()));
    const size_t ef_keep = ef_;
    loadIndex(slim_location, s);
    setEf(ef_keep);
  }
  void addPoint(const void *, labeltype, bool = false) override {
    throw std::runtime_error("HierarchicalNSWSlim does not support addPoint");  // hnswalg_slim.h:149-152
  }
  // saveIndex (hnswalg_slim.h:717-751): the Slim file this object holds (written by convertFromHNSW or loaded), copied
  void saveIndex(const std::string &location) override {
    if (path_.empty()) throw std::runtime_error("hnswlib_amd: nothing to save");
    if (location == path_) return;
    std::ifstream in(path_, std::ios::binary);
    std::ofstream out(location, std::ios::binary);
    if (!in.is_open() || !out.is_open()) throw std::runtime_error("Cannot open file");
    out << in.rdbuf();
  }
  // searchKnn(q, k, filter) / searchKnn(q, k): hnswalg_slim.h:1783-1905 / 1907-2028
  std::priority_queue<std::pair<float, labeltype>> searchKnn(const void *query_data, size_t k,
                                                             BaseFilterFunctor *isIdAllowed = nullptr) const override {
    return search_pq(query_data, k, isIdAllowed);
  }
  // searchKnn(q, k, tableint* result): hnswalg_slim.h:2030-2131.  k labels; same k-subset as the reference,
  // sorted by distance (call setExactOrder(true) to also reproduce the reference's array order).
  void searchKnn(const void *query_data, size_t k, tableint *result) const {
    if (!h_) return;  // cur_element_count_ == 0 (hnswalg_slim.h:2031-2032)
    detail::check(hs_search_batch(h_, (const float *)query_data, 1, k, HS_MODE_SLIM_IDS, result, nullptr, nullptr, nullptr, nullptr));
  }
  // The fast entry: every row of `queries` in one launch.
  // With setDevices({...}) before loadIndex the batch is sharded over the devices and gathered (RCCL over xGMI).
  void searchKnnBatch(const float *queries, size_t nq, size_t k, tableint *results) const {
    if (comm_) detail::check(hs_search_batch_sharded(comm_, replicas_.data(), queries, nq, k, HS_MODE_SLIM_IDS, results, nullptr, nullptr, nullptr));
    else detail::check(hs_search_batch(h_, queries, nq, k, HS_MODE_SLIM_IDS, results, nullptr, nullptr, nullptr, nullptr));
  }
  void setExactOrder(bool on) {
    detail::check(hs_set_exact_order(h_, on ? 1 : 0));
    for (size_t r = 1; r < replicas_.size(); r++) detail::check(hs_set_exact_order(replicas_[r], on ? 1 : 0));
  }
};

// BruteforceSearch (bruteforce.h): exhaustive scan.  addPoint keeps the rows on the host; searchKnn / searchKnnBatch
// run the exhaustive-scan kernel over them (hs_brute_force: same distances, same (dist, label) tie order).
template <typename dist_t>
class BruteforceSearch;

template <>
class BruteforceSearch<float> : public AlgorithmInterface<float> {
  std::vector<float> rows_;
  std::vector<uint64_t> labels_;
  size_t dim_ = 0, max_ = 0;
  int metric_ = HS_METRIC_L2;
 public:
  BruteforceSearch(SpaceInterface<float> *s, size_t maxElements) : dim_(detail::dim_of(s)), max_(maxElements), metric_(detail::metric_of(s)) {
    rows_.reserve(maxElements * dim_);
    labels_.reserve(maxElements);
  }
  void addPoint(const void *datapoint, labeltype label, bool = false) override {
    for (size_t i = 0; i < labels_.size(); i++)
      if (labels_[i] == label) {  // bruteforce.h:66-70: an existing label is overwritten in place
        std::copy((const float *)datapoint, (const float *)datapoint + dim_, rows_.begin() + i * dim_);
        return;
      }
    if (labels_.size() >= max_) throw std::runtime_error("The number of elements exceeds the specified limit\n");  // :72-74
    rows_.insert(rows_.end(), (const float *)datapoint, (const float *)datapoint + dim_);
    labels_.push_back(label);
  }
  void saveIndex(const std::string &) override { throw std::runtime_error("hnswlib_amd: BruteforceSearch::saveIndex is not provided"); }
  std::priority_queue<std::pair<float, labeltype>> searchKnn(const void *query_data, size_t k, BaseFilterFunctor *f = nullptr) const override {
    if (f) throw std::runtime_error("hnswlib_amd: BruteforceSearch with a filter is not provided");
    std::priority_queue<std::pair<float, labeltype>> r;
    if (labels_.empty()) return r;
    std::vector<uint64_t> lab(k);
    std::vector<float> d(k);
    uint32_t cnt = 0;
    detail::check(hs_brute_force(rows_.data(), labels_.size(), dim_, metric_, labels_.data(), (const float *)query_data, 1, k, 0, lab.data(),
                                 d.data(), &cnt));
    for (uint32_t i = 0; i < cnt; i++) r.emplace(d[i], (labeltype)lab[i]);
    return r;
  }
  // every row of `queries` in one launch; out_* are nq x k, ascending by (dist, label)
  void searchKnnBatch(const float *queries, size_t nq, size_t k, uint64_t *out_labels, float *out_dists, uint32_t *out_counts = nullptr) const {
    detail::check(hs_brute_force(rows_.data(), labels_.size(), dim_, metric_, labels_.data(), queries, nq, k, 0, out_labels, out_dists, out_counts));
  }
};

// HierarchicalNSWSlimQ (hnswalg_slimq.h): the RaBitQ-quantised variant.  Same call sequence as the reference's
// strategy (include/strategy/hnsw_slimq_strategy.h:72,142-156): loadIndex, setDataset, setEf, searchKnn(q, K, result).
template <typename dist_t>
class HierarchicalNSWSlimQ;

template <>
class HierarchicalNSWSlimQ<float> : public AlgorithmInterface<float>, public detail::DeviceIndex {
 public:
  explicit HierarchicalNSWSlimQ(SpaceInterface<float> *) {}
  HierarchicalNSWSlimQ(SpaceInterface<float> *s, const std::string &location, bool /*nmslib*/ = false,
                       size_t max_elements = 0, bool /*allow_replace_deleted*/ = false) {
    loadIndex(location, s, max_elements);
  }
  void loadIndex(const std::string &location, SpaceInterface<float> *s, size_t max_elements_i = 0) {
    load(location, HS_KIND_SLIMQ, s, max_elements_i);
  }
  // setDataset (hnswalg_slimq.h:303-305): the raw rows, indexed by internal id, for the exact re-rank
  void setDataset(std::vector<std::vector<float>> *data_set) {
    if (!h_ || !data_set || data_set->empty()) return;
    const size_t n = data_set->size(), d = (*data_set)[0].size();
    std::vector<float> flat(n * d);
    for (size_t i = 0; i < n; i++) std::copy((*data_set)[i].begin(), (*data_set)[i].end(), flat.begin() + i * d);
    detail::check(hs_slimq_set_dataset(h_, flat.data(), n, d));
  }
  void setDataset(const float *rows, size_t n, size_t d) { detail::check(hs_slimq_set_dataset(h_, rows, n, d)); }
  void setTConst(double t) { detail::check(hs_slimq_set_tconst(h_, t)); }
  void addPoint(const void *, labeltype, bool = false) override {
    throw std::runtime_error("HierarchicalNSWSlimQ does not support addPoint");  // hnswalg_slimq.h:298-301
  }
  void saveIndex(const std::string &) override {
    throw std::runtime_error("hnswlib_amd: the device index is read-only; the file it was loaded from is unchanged");
  }
  // the priority_queue overloads of the reference print "todo: searchKnn()" and return nothing (:1795-1808)
  std::priority_queue<std::pair<float, labeltype>> searchKnn(const void *, size_t, BaseFilterFunctor * = nullptr) const override {
    return {};
  }
  // searchKnn(q, k, tableint* result): hnswalg_slimq.h:1810-1924 -- k labels in the reference's heap-array order
  void searchKnn(const void *query_data, size_t k, tableint *result) const {
    if (!h_) return;
    std::vector<uint64_t> lab(k);
    detail::check(hs_slimq_search_batch(h_, (const float *)query_data, 1, k, lab.data(), nullptr, nullptr, nullptr));
    for (size_t i = 0; i < k; i++) result[i] = (tableint)lab[i];
  }
  // The fast entry: every row of `queries` in one launch; results nq x k.
  void searchKnnBatch(const float *queries, size_t nq, size_t k, tableint *results) const {
    std::vector<uint64_t> lab(nq * k);
    detail::check(hs_slimq_search_batch(h_, queries, nq, k, lab.data(), nullptr, nullptr, nullptr));
    for (size_t i = 0; i < nq * k; i++) results[i] = (tableint)lab[i];
  }
};

}  // namespace hnswlib
