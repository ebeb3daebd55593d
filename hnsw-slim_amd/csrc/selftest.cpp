// selftest.cpp -- host-only check that heap_emul.hpp reproduces libstdc++'s heap / nth_element
// mechanics bit for bit (array layout after every operation), on tie-heavy random sequences.
// Exit code 0 = identical.  Run by tests/test_heap_emul.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

#include "heap_emul.hpp"
#include "host_graph.hpp"

using P = std::pair<float, uint32_t>;
struct LessP { bool operator()(const P &a, const P &b) const { return a.first < b.first; } };
struct GreaterP { bool operator()(const P &a, const P &b) const { return a.first > b.first; } };

static bool same(const std::vector<P> &s, const std::vector<hs::Pair> &e, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (s[i].first != e[i].d || s[i].second != e[i].id) return false;
  return true;
}

template <class SC, class EC>
static int run_heap(unsigned seed, int range, SC sc, EC ec) {
  std::mt19937 rng(seed);
  std::vector<P> s;
  std::vector<hs::Pair> e;
  uint32_t next_id = 0;
  for (int op = 0; op < 4000; op++) {
    bool push = s.empty() || (rng() % 100) < 60;
    if (push) {
      float d = (float)(rng() % range);
      s.emplace_back(d, next_id);
      e.push_back({d, next_id});
      next_id++;
      std::push_heap(s.begin(), s.end(), sc);
      hs::push_heap(e.data(), (long)e.size(), ec);
    } else {
      std::pop_heap(s.begin(), s.end(), sc);
      hs::pop_heap(e.data(), (long)e.size(), ec);
      if (!same(s, e, s.size())) return 1;
      s.pop_back();
      e.pop_back();
    }
    if (!same(s, e, s.size())) return 1;
    if (op % 500 == 499) {  // make_heap on a shuffled copy
      std::vector<P> s2 = s;
      std::shuffle(s2.begin(), s2.end(), rng);
      std::vector<hs::Pair> e2(s2.size());
      for (size_t i = 0; i < s2.size(); i++) e2[i] = {s2[i].first, s2[i].second};
      std::make_heap(s2.begin(), s2.end(), sc);
      hs::make_heap(e2.data(), (long)e2.size(), ec);
      if (!same(s2, e2, s2.size())) return 2;
    }
  }
  return 0;
}

static int run_nth(unsigned seed, int range) {
  std::mt19937 rng(seed);
  for (int it = 0; it < 3000; it++) {
    size_t n = 1 + rng() % 300;
    size_t k = rng() % (n + 1);
    std::vector<P> s(n);
    std::vector<hs::Pair> e(n);
    bool sorted_input = (it % 7 == 0), organ = (it % 11 == 0);
    for (size_t i = 0; i < n; i++) {
      float d = (float)(rng() % range);
      if (sorted_input) d = (float)i;
      if (organ) d = (float)(i < n / 2 ? i : n - i);
      s[i] = {d, (uint32_t)i};
      e[i] = {d, (uint32_t)i};
    }
    std::nth_element(s.begin(), s.begin() + k, s.end(), LessP());
    hs::nth_element(e.data(), (long)k, (long)n, hs::LessD());
    if (!same(s, e, n)) return 3;
  }
  return 0;
}

// hs::std_sort against std::sort with a by-key-only comparator on tie-heavy, sorted, reversed and organ-pipe inputs
static int run_sort(unsigned seed, int range) {
  std::mt19937 rng(seed);
  for (int it = 0; it < 2000; it++) {
    size_t n = 1 + rng() % (it % 5 == 0 ? 3000 : 200);
    std::vector<P> s(n);
    std::vector<hs::Pair> e(n);
    for (size_t i = 0; i < n; i++) {
      float d = (float)(rng() % range);
      if (it % 7 == 0) d = (float)i;
      if (it % 11 == 0) d = (float)(i < n / 2 ? i : n - i);
      if (it % 13 == 0) d = (float)(n - i);
      if (it % 17 == 0) d = (float)((i * 7919u) % 5);
      s[i] = {d, (uint32_t)i};
      e[i] = {d, (uint32_t)i};
    }
    std::sort(s.begin(), s.end(), LessP());
    if (!hs::std_sort(e.data(), (long)n, hs::LessD())) return 5;
    if (!same(s, e, n)) return 4;
  }
  return 0;
}

// selftest loadmem <vanilla index> <slim index> <dim>: the loaders read the serialized bytes from host memory
// (BinSource / MemBuf, behind hs_index_load_mem) exactly as they read the files; a truncated buffer is rejected.
static std::vector<char> slurp(const char *path) {
  std::ifstream in(path, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static int run_loadmem(const char *vanilla, const char *slim, size_t dim) {
  const std::vector<char> vb = slurp(vanilla), sb = slurp(slim);
  if (vb.empty() || sb.empty()) return 1;
  hs::VanillaGraph a, b;
  a.load(vanilla, hs::METRIC_L2, dim);
  b.load(hs::BinSource(vb.data(), vb.size()), hs::METRIC_L2, dim);
  if (a.count != b.count || a.maxlevel != b.maxlevel || a.enterpoint != b.enterpoint || a.level0 != b.level0 || a.links != b.links ||
      a.levels != b.levels)
    return 2;
  hs::SlimGraph c, d;
  c.load(slim, hs::METRIC_L2, dim);
  d.load(hs::BinSource(sb.data(), sb.size()), hs::METRIC_L2, dim);
  if (c.count != d.count || c.maxlevel != d.maxlevel || c.enterpoint != d.enterpoint || c.elements != d.elements || c.blobs != d.blobs)
    return 3;
  for (size_t cut : {sb.size() / 2, sb.size() - 1, (size_t)40}) {
    try {
      hs::SlimGraph e;
      e.load(hs::BinSource(sb.data(), cut), hs::METRIC_L2, dim);
      return 4;   // a truncated image must not load
    } catch (std::runtime_error &ex) {
      if (std::string(ex.what()) != "Index seems to be corrupted or unsupported") return 5;
    }
  }
  try {
    hs::VanillaGraph e;
    e.load(hs::BinSource(vb.data(), vb.size() / 3), hs::METRIC_L2, dim);
    return 6;
  } catch (std::runtime_error &) {
  }
  printf("loaders: memory image == file (vanilla n=%zu, slim n=%zu); truncated images rejected\n", a.count, c.count);
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 5 && std::string(argv[1]) == "loadmem") return run_loadmem(argv[2], argv[3], (size_t)atoll(argv[4]));
  int rc = 0;
  for (unsigned seed = 1; seed <= 8 && !rc; seed++)
    for (int range : {3, 17, 1000, 1 << 20}) {
      if ((rc = run_heap(seed, range, LessP(), hs::LessD()))) break;
      if ((rc = run_heap(seed + 100, range, GreaterP(), hs::GreaterD()))) break;
      if ((rc = run_nth(seed + 200, range))) break;
      if ((rc = run_sort(seed + 300, range))) break;
    }
  printf(rc ? "heap_emul MISMATCH rc=%d\n" : "heap_emul identical to libstdc++ (rc=%d)\n", rc);
  return rc;
}
