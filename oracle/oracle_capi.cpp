// oracle/oracle_capi.cpp -- TEST INFRASTRUCTURE ONLY: C entry points over hs_oracle.hpp for ctypes
// (tests/, smoke(), bench.py cpu_baseline).  Never linked into the product library.
#include "hs_oracle.hpp"
#include "hs_oracle_slimq.hpp"
#include <omp.h>

using namespace hso;

struct Handle {
  int kind;  // 0 vanilla, 1 slim
  int mark_ep = -1;  // -1: as the overload does; 0 / 1: forced (cross-pin knob, see slim_search_pq)
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
void hso_set_mark_ep(void *p, int v) { ((Handle *)p)->mark_ep = v; }
// level-0 entry node of each query (Slim index)
void hso_slim_entry(void *p, const float *q, size_t nq, uint32_t *out) {
  auto *h = (Handle *)p;
  for (size_t i = 0; i < nq; i++) out[i] = slim_entry(h->s, q + i * h->s.dim);
}
// SearchBuffer restatement driven by an op sequence: op[i] = 1 -> `if (!is_full(d)) insert(id, d)`, 0 -> `if (has_next()) pop()`.
// ev[i] = 1/0 (inserted or not) resp. the popped id (0xFFFFFFFF when nothing to pop); final array -> out_id/out_d, returns size.
size_t hso_pool_run(size_t cap, const uint8_t *op, const uint32_t *ids, const float *d, size_t n, uint32_t *ev, uint32_t *out_id,
                    float *out_d) {
  Pool pool(cap);
  for (size_t i = 0; i < n; i++) {
    if (op[i]) {
      if (pool.is_full(d[i])) { ev[i] = 0; continue; }
      pool.insert(ids[i], d[i]);
      ev[i] = 1;
    } else {
      ev[i] = pool.has_next() ? pool.pop() : 0xFFFFFFFFu;
    }
  }
  for (size_t i = 0; i < pool.size; i++) { out_id[i] = pool.d[i].second; out_d[i] = pool.d[i].first; }
  return pool.size;
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
                                    : slim_search_pq(h->s, q + i * dim, k, s, res, h->mark_ep != 0);
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

// ---- HNSW-SlimQ ---------------------------------------------------------------------------------------------
void *hso_slimq_load(const char *path) {
  try { auto *ix = new SlimQIndex(); ix->load(path); return ix; }
  catch (std::exception &e) { g_err = e.what(); return nullptr; }
}
void hso_slimq_free(void *p) { delete (SlimQIndex *)p; }
void hso_slimq_set(void *p, size_t ef, double t_const, const float *raw) {
  auto *ix = (SlimQIndex *)p; ix->ef = ef; ix->t_const = t_const; ix->raw = raw;
}
void hso_slimq_perturb(void *p, const int *ulps) { auto *ix = (SlimQIndex *)p; for (int i = 0; i < 4; i++) ix->perturb_ulps[i] = ulps[i]; }
void hso_slimq_info(void *p, uint64_t *out) {
  auto *ix = (SlimQIndex *)p;
  out[0] = ix->count; out[1] = ix->dim; out[2] = ix->padded; out[3] = ix->ncl; out[4] = (uint64_t)ix->maxlevel;
  out[5] = ix->enterpoint; out[6] = (uint64_t)ix->metric; out[7] = (uint64_t)ix->threshold_level;
}
// Piece-wise entry points with explicit operands (pinned against tests/golden/rabitq_ref.npz).
void hso_rq_rotate(size_t dim, const uint8_t *flips, const float *x, size_t n, float *out) {
  SlimQIndex ix;
  ix.dim = dim; ix.padded = (dim + 63) / 64 * 64;
  ix.flips.assign(flips, flips + 4 * ix.padded / 8);
  for (size_t i = 0; i < n; i++) ix.rotate(x + i * dim, out + i * ix.padded);
}
// q3 n x {delta, vl, k1xsumq}; planes n x padded/64*4; q2c n x (ncl or 2 ncl for IP)
void hso_rq_prepare(size_t padded, int metric, double t_const, const float *rq, size_t n, const float *cent, size_t ncl,
                    float *q3, uint64_t *planes, float *q2c) {
  SlimQIndex ix;
  ix.padded = padded; ix.metric = metric; ix.t_const = t_const; ix.ncl = ncl;
  ix.cent.assign(cent, cent + ncl * padded);
  SlimQIndex::Query Q;
  for (size_t i = 0; i < n; i++) {
    ix.prepare(rq + i * padded, Q);
    q3[i * 3] = Q.delta; q3[i * 3 + 1] = Q.vl; q3[i * 3 + 2] = Q.k1xsumq;
    std::copy(Q.planes.begin(), Q.planes.end(), planes + i * Q.planes.size());
    std::copy(Q.q2c.begin(), Q.q2c.end(), q2c + i * Q.q2c.size());
  }
}
// out nq x nd estimated distances with the query-side g_add given per query
void hso_rq_est(size_t padded, int metric, const uint64_t *codes, const float *fac, size_t nd, const float *q3,
                const uint64_t *planes, const float *g_add, size_t nq, float *out) {
  SlimQIndex ix;
  ix.padded = padded; ix.metric = metric; ix.ncl = 1;
  ix.code.assign(codes, codes + nd * padded / 64);
  ix.fac.assign(fac, fac + nd * 3);
  ix.cid.assign(nd, 0);
  SlimQIndex::Query Q;
  for (size_t i = 0; i < nq; i++) {
    Q.delta = q3[i * 3]; Q.vl = q3[i * 3 + 1]; Q.k1xsumq = q3[i * 3 + 2];
    Q.planes.assign(planes + i * padded / 64 * 4, planes + (i + 1) * padded / 64 * 4);
    for (size_t j = 0; j < nd; j++) out[i * nd + j] = ix.est_g(Q, (uint32_t)j, g_add[i]);
  }
}
// out_ids/out_d: nq x k in the reference's heap-array order (labels), out_cnt = entries found, counters nq x 4.
int hso_slimq_search(void *p, const float *q, size_t nq, size_t k, uint64_t *out_l, float *out_d, uint32_t *out_cnt,
                     uint64_t *counters, int threads) {
  auto *ix = (SlimQIndex *)p;
  if (!ix->raw || ix->t_const <= 0) { g_err = "dataset / t_const not set"; return 1; }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
  for (long i = 0; i < (long)nq; i++) {
    std::vector<std::pair<float, uint32_t>> heap;
    SlimQCounters c;
    size_t f = slimq_search(*ix, q + i * ix->dim, k, heap, &c);
    out_cnt[i] = (uint32_t)f;
    for (size_t j = 0; j < k; j++) {
      out_l[i * k + j] = j < f ? ix->label[heap[j].second] : ~0ull;
      out_d[i * k + j] = j < f ? heap[j].first : std::numeric_limits<float>::infinity();
    }
    if (counters) { counters[i * 4] = c.n_hops; counters[i * 4 + 1] = c.n_est; counters[i * 4 + 2] = c.n_insert; counters[i * 4 + 3] = c.n_revisit; }
  }
  return 0;
}
// the SearchBuffer pops of one query in order (bit 31 = revisit); returns the number of pops
size_t hso_slimq_trace(void *p, const float *q, size_t k, uint32_t *out, size_t cap) {
  auto *ix = (SlimQIndex *)p;
  std::vector<std::pair<float, uint32_t>> heap;
  std::vector<uint32_t> tr;
  slimq_search(*ix, q, k, heap, nullptr, &tr);
  for (size_t i = 0; i < tr.size() && i < cap; i++) out[i] = tr[i];
  return tr.size();
}
// rotation + preparation of queries against a loaded index: rq n x padded, q3 n x 3, planes, g_add n x ncl
void hso_slimq_prepare(void *p, const float *q, size_t n, float *rq, float *q3, uint64_t *planes, float *g_add) {
  auto *ix = (SlimQIndex *)p;
  SlimQIndex::Query Q;
  for (size_t i = 0; i < n; i++) {
    float *r = rq + i * ix->padded;
    ix->rotate(q + i * ix->dim, r);
    ix->prepare(r, Q);
    q3[i * 3] = Q.delta; q3[i * 3 + 1] = Q.vl; q3[i * 3 + 2] = Q.k1xsumq;
    std::copy(Q.planes.begin(), Q.planes.end(), planes + i * Q.planes.size());
    for (size_t c = 0; c < ix->ncl; c++) g_add[i * ix->ncl + c] = ix->metric == METRIC_IP ? -Q.q2c[c] : Q.q2c[c] * Q.q2c[c];
  }
}
}
