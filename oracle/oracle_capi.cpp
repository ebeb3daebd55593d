// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY: C entry points over hs_oracle.hpp for ctypes
// (tests/, smoke(), bench.py cpu_baseline).  Never linked into the product library.
#include "hs_oracle.hpp"
#include <omp.h>

using namespace hso;

struct Handle {
  int kind;  // 0 vanilla, 1 slim
  VanillaIndex v;
  SlimIndex s;
};
static thread_local std::string g_err;

extern "C" {

const char *hso_last_error() { return g_err.c_str(); }

void *hso_load(const char *path, int kind, int metric, size_t dim) {
  try {
    auto *h = new Handle();
    h->kind = kind;
    if (kind == 0) h->v.load(path, (Metric)metric, dim);
    else h->s.load(path, (Metric)metric, dim);
    return h;
  } catch (std::exception &e) { g_err = e.what(); return nullptr; }
}
void hso_free(void *p) { delete (Handle *)p; }
void hso_set_ef(void *p, size_t ef) { auto *h = (Handle *)p; h->v.ef = ef; h->s.ef = ef; }
// filter as an allowed[internal id] byte array (null = remove the filter)
void hso_set_filter(void *p, const uint8_t *allowed) {
  auto *h = (Handle *)p;
  size_t n = h->kind == 0 ? h->v.count : h->s.count;
  auto &dst = h->kind == 0 ? h->v.allowed : h->s.allowed;
  if (allowed) dst.assign(allowed, allowed + n);
  else dst.clear();
}
size_t hso_count(void *p) { auto *h = (Handle *)p; return h->kind == 0 ? h->v.count : h->s.count; }
int hso_maxlevel(void *p) { auto *h = (Handle *)p; return h->kind == 0 ? h->v.maxlevel : h->s.maxlevel; }

static void put_raw(const SlimResult &r, size_t cap, float *rd, uint32_t *ri, uint32_t *rs, uint32_t *cnt, size_t i) {
  if (rs) rs[i] = r.top.size();
  if (rd && ri)
    for (size_t j = 0; j < r.top.size() && j < cap; j++) { rd[i * cap + j] = r.top[j].first; ri[i * cap + j] = r.top[j].second; }
  if (cnt) { cnt[i * 5 + 0] = r.c.n_dist; cnt[i * 5 + 1] = r.c.n_hops; cnt[i * 5 + 2] = r.c.n_nbr; cnt[i * 5 + 3] = r.c.n_accept; cnt[i * 5 + 4] = r.c.max_cand; }
}

// HierarchicalNSWSlim::searchKnn(q,k,tableint*) for nq queries. raw_* (nullable): top_candidates
// arrays, row stride raw_cap.  counters (nullable): nq x 5 u32.
int hso_slim_search_ids(void *p, const float *q, size_t nq, size_t k, uint32_t *out, size_t raw_cap,
                        float *raw_d, uint32_t *raw_i, uint32_t *raw_sz, uint32_t *counters, int threads) {
  auto *h = (Handle *)p;
  int rc = 0;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    Scratch s;
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < (long)nq; i++) {
      try {
        SlimResult r = slim_search_ids(h->s, q + i * h->s.dim, k, s, out + i * k);
        put_raw(r, raw_cap, raw_d, raw_i, raw_sz, counters, i);
      } catch (std::exception &e) {
#pragma omp critical
        { g_err = e.what(); rc = 1; }
      }
    }
  }
  return rc;
}

// pq-returning overloads: out_d/out_l are nq x k in pop order (farthest first), out_cnt = sizes.
int hso_search_pq(void *p, const float *q, size_t nq, size_t k, float *out_d, uint64_t *out_l, uint32_t *out_cnt,
                  size_t raw_cap, float *raw_d, uint32_t *raw_i, uint32_t *raw_sz, uint32_t *counters, int threads) {
  auto *h = (Handle *)p;
  int rc = 0;
  size_t dim = h->kind == 0 ? h->v.dim : h->s.dim;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
  {
    Scratch s;
    std::vector<std::pair<float, uint64_t>> res;
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < (long)nq; i++) {
      try {
        SlimResult r = h->kind == 0 ? vanilla_search_pq(h->v, q + i * dim, k, s, res)
                                    : slim_search_pq(h->s, q + i * dim, k, s, res);
        out_cnt[i] = res.size();
        for (size_t j = 0; j < res.size(); j++) { out_d[i * k + j] = res[j].first; out_l[i * k + j] = res[j].second; }
        put_raw(r, raw_cap, raw_d, raw_i, raw_sz, counters, i);
      } catch (std::exception &e) {
#pragma omp critical
        { g_err = e.what(); rc = 1; }
      }
    }
  }
  return rc;
}

int hso_dist(int metric, const float *a, const float *b, size_t n, size_t d, float *out) {
  try {
    for (size_t i = 0; i < n; i++) out[i] = dist((Metric)metric, a + i * d, b + i * d, d);
    return 0;
  } catch (std::exception &e) { g_err = e.what(); return 1; }
}

int hso_brute_force(int metric, const float *base, size_t n, size_t d, const float *q, size_t nq, size_t k,
                    uint32_t *out, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
  for (long i = 0; i < (long)nq; i++) brute_force((Metric)metric, base, n, d, q + i * d, k, out + i * k);
  return 0;
}
}
