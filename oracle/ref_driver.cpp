// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Driver around the *unmodified* reference headers under
// /root/reference/third_party/hnswlib (vanilla hnswlib::HierarchicalNSW, L2Space,
// InnerProductSpace).  It is compiled by oracle/Makefile from the sources where
// they lie (no copies, no stand-in headers) into oracle/_ref/ref_hnsw and is used to
//   (1) pin the distance-function recipes (space_l2.h:25-54, space_ip.h:146-199),
//   (2) build vanilla HNSW index files single-threaded (hnswalg.h:1248-1376, 748-779),
//   (3) dump searchKnn results (hnswalg.h:1378-1440) + per-query distance-call counts,
//   (4) dump BruteforceSearch::searchKnn results (bruteforce.h:106-135)
// as golden fixtures under tests/golden/.
//
// NOT buildable here: hnswalg_slim.h / hnswalg_slimq.h (they include
// <folly/concurrency/container/atomic_grow_array.h>, absent from this image), so the Slim
// classes have no compiled reference; see DESIGN.md "Oracle pinning".
#include "hnswlib.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

static long g_dist_calls = 0;
static hnswlib::DISTFUNC<float> g_real_fn = nullptr;
static float counting_fn(const void *a, const void *b, const void *p) {
  g_dist_calls++;
  return g_real_fn(a, b, p);
}

static std::vector<float> read_fvecs(const std::string &path, size_t &n, size_t &d) {
  std::ifstream in(path, std::ios::binary);
  if (!in) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  std::vector<float> out;
  n = 0; d = 0;
  int32_t dim;
  while (in.read((char *)&dim, 4)) {
    d = dim;
    size_t old = out.size();
    out.resize(old + d);
    in.read((char *)(out.data() + old), 4 * d);
    n++;
  }
  return out;
}

static hnswlib::SpaceInterface<float> *make_space(const std::string &metric, size_t d) {
  if (metric == "l2") return new hnswlib::L2Space(d);
  if (metric == "ip") return new hnswlib::InnerProductSpace(d);
  fprintf(stderr, "bad metric %s\n", metric.c_str());
  exit(2);
}

// dist <metric> <a.fvecs> <b.fvecs> <out.f32>
static int cmd_dist(int argc, char **argv) {
  if (argc < 6) return 2;
  size_t na, da, nb, db;
  auto A = read_fvecs(argv[3], na, da);
  auto B = read_fvecs(argv[4], nb, db);
  if (na != nb || da != db) return 3;
  auto *space = make_space(argv[2], da);
  auto fn = space->get_dist_func();
  void *param = space->get_dist_func_param();
  std::vector<float> out(na);
  for (size_t i = 0; i < na; i++) out[i] = fn(A.data() + i * da, B.data() + i * da, param);
  std::ofstream o(argv[5], std::ios::binary);
  o.write((char *)out.data(), 4 * na);
  return 0;
}

// build <metric> <base.fvecs> <out.index> <M> <efC> <branching> <seed>
static int cmd_build(int argc, char **argv) {
  if (argc < 9) return 2;
  size_t n, d;
  auto X = read_fvecs(argv[3], n, d);
  auto *space = make_space(argv[2], d);
  hnswlib::HierarchicalNSW<float> index(space, n, atoi(argv[5]), atoi(argv[6]), argv[7],
                                        atoi(argv[8]));
  for (size_t i = 0; i < n; i++) index.addPoint(X.data() + i * d, i);  // serial: ids == labels
  index.saveIndex(argv[4]);
  printf("built n=%zu d=%zu maxlevel=%d ep=%u\n", n, d, index.maxlevel_, index.enterpoint_node_);
  return 0;
}

// Filter used by `searchf`: allows every label with label % mod != rem  (BaseFilterFunctor, hnswlib.h:128-133).
struct ModFilter : public hnswlib::BaseFilterFunctor {
  size_t mod, rem;
  ModFilter(size_t m, size_t r) : mod(m), rem(r) {}
  bool operator()(hnswlib::labeltype id) override { return id % mod != rem; }
};
static ModFilter *g_filter = nullptr;

// search <metric> <index> <query.fvecs> <out.bin> <k> <ef> [ef...]
// out.bin: u32 nq, u32 k, u32 n_ef, then per ef: u32 ef, per query: u32 cnt, u32 n_dist_calls,
//          cnt x {f32 dist, u64 label} in priority_queue pop order (farthest first).
static int cmd_search(int argc, char **argv) {
  if (argc < 8) return 2;
  size_t nq, d;
  auto Q = read_fvecs(argv[4], nq, d);
  auto *space = make_space(argv[2], d);
  hnswlib::HierarchicalNSW<float> index(space, argv[3]);
  g_real_fn = index.fstdistfunc_;
  index.fstdistfunc_ = counting_fn;  // public member (hnswalg.h:57)
  uint32_t k = atoi(argv[6]);
  std::ofstream o(argv[5], std::ios::binary);
  uint32_t nq32 = nq, nef = argc - 7;
  o.write((char *)&nq32, 4); o.write((char *)&k, 4); o.write((char *)&nef, 4);
  for (int a = 7; a < argc; a++) {
    uint32_t ef = atoi(argv[a]);
    index.setEf(ef);
    o.write((char *)&ef, 4);
    for (size_t i = 0; i < nq; i++) {
      g_dist_calls = 0;
      auto res = index.searchKnn(Q.data() + i * d, k, g_filter);  // hnswalg.h:1378-1380
      uint32_t cnt = res.size(), calls = g_dist_calls;
      o.write((char *)&cnt, 4); o.write((char *)&calls, 4);
      while (!res.empty()) {
        float dist = res.top().first;
        uint64_t label = res.top().second;
        o.write((char *)&dist, 4); o.write((char *)&label, 8);
        res.pop();
      }
    }
  }
  return 0;
}

// markdel <metric> <dim> <index_in> <index_out> <every>
// markDelete (hnswalg.h:923-1010) every `every`-th label, then saveIndex: an index file carrying delete marks,
// which drives the !bare_bone_search branch of searchBaseLayerST (hnswalg.h:347-349, 441-444).
static int cmd_markdel(int argc, char **argv) {
  if (argc < 7) return 2;
  auto *space = make_space(argv[2], atoi(argv[3]));
  hnswlib::HierarchicalNSW<float> index(space, argv[4]);
  size_t every = atoi(argv[6]);
  size_t n = index.cur_element_count;
  for (size_t l = every / 2; l < n; l += every) index.markDelete(l);
  index.saveIndex(argv[5]);
  printf("marked %zu deleted\n", (size_t)index.num_deleted_);
  return 0;
}

// bf <metric> <base.fvecs> <query.fvecs> <out.bin> <k>
// hnswlib::BruteforceSearch (bruteforce.h): addPoint loop (label = row), searchKnn (:106-135) per query.
// out.bin: u32 nq, u32 k, per query k x {f32 dist, u64 label} in priority_queue pop order (farthest first).
static int cmd_bf(int argc, char **argv) {
  if (argc < 7) return 2;
  size_t n, d, nq, dq;
  auto X = read_fvecs(argv[3], n, d);
  auto Q = read_fvecs(argv[4], nq, dq);
  if (d != dq) return 3;
  auto *space = make_space(argv[2], d);
  hnswlib::BruteforceSearch<float> bf(space, n);
  for (size_t i = 0; i < n; i++) bf.addPoint(X.data() + i * d, i);
  uint32_t k = atoi(argv[6]), nq32 = nq;
  std::ofstream o(argv[5], std::ios::binary);
  o.write((char *)&nq32, 4); o.write((char *)&k, 4);
  for (size_t i = 0; i < nq; i++) {
    auto res = bf.searchKnn(Q.data() + i * d, k);
    if (res.size() != k) return 4;
    while (!res.empty()) {
      float dist = res.top().first;
      uint64_t label = res.top().second;
      o.write((char *)&dist, 4); o.write((char *)&label, 8);
      res.pop();
    }
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: ref_hnsw dist|build|search|bf ...\n"); return 2; }
  std::string c = argv[1];
  if (c == "dist") return cmd_dist(argc, argv);
  if (c == "build") return cmd_build(argc, argv);
  if (c == "search") return cmd_search(argc, argv);
  if (c == "markdel") return cmd_markdel(argc, argv);
  if (c == "bf") return cmd_bf(argc, argv);
  if (c == "searchf") {  // searchf <mod> <rem> <metric> <index> <query.fvecs> <out.bin> <k> <ef>...
    if (argc < 10) return 2;
    g_filter = new ModFilter(atoi(argv[2]), atoi(argv[3]));
    std::vector<char *> a2 = {argv[0], argv[1]};
    for (int i = 4; i < argc; i++) a2.push_back(argv[i]);
    return cmd_search((int)a2.size(), a2.data());
  }
  return 2;
}
