// oracle/ref_rabitq.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Driver around the *unmodified* vendored rabitqlib (/root/reference/third_party/rabitqlib, with the vendored
// Eigen), compiled by oracle/Makefile from the sources where they lie.  It pins the pieces of the HNSW-SlimQ
// search path that are separable from hnswalg_slimq.h (which itself needs folly and is unbuildable here):
//   rotate : FhtKacRotator::rotate                     rabitqlib/utils/rotator.hpp:370-423
//   query  : SplitSingleQuery ctor (4-bit scalar quantisation + bit-plane transpose)
//                                                        rabitqlib/index/query.hpp:112-156
//            + split_single_estdist / warmup_ip_x0_q     rabitqlib/index/estimator.hpp:164-188,
//                                                        rabitqlib/utils/warmup_space.hpp:8-102
//   data   : one_bit_compact_code (1-bit code + f_add, f_rescale, f_error)
//                                                        rabitqlib/quantization/rabitq_impl.hpp:75-187
//   cent   : euclidean_sqr / dot_product (q_to_centroids) rabitqlib/utils/space.hpp:226-253
//   buffer : rabitqlib::buffer::SearchBuffer            rabitqlib/utils/buffer.hpp:16-100
//   hnsw   : rabitqlib::hnsw::HierarchicalNSW ctor + construct (the graph HNSW-SlimQ is converted from), serial
//                                                        rabitqlib/index/hnsw/hnsw.hpp:427-500, 667-1054
// All arrays are raw little-endian binaries; shapes are passed on the command line (tests/golden/make_golden.py).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "rabitqlib/index/estimator.hpp"
#include "rabitqlib/index/hnsw/hnsw.hpp"
#include "rabitqlib/index/query.hpp"
#include "rabitqlib/quantization/rabitq.hpp"
#include "rabitqlib/utils/buffer.hpp"
#include "rabitqlib/utils/rotator.hpp"
#include "rabitqlib/utils/space.hpp"

template <typename T>
static std::vector<T> rd(const char *p, size_t n) {
  std::vector<T> v(n);
  std::ifstream in(p, std::ios::binary);
  in.read((char *)v.data(), n * sizeof(T));
  if (!in) { fprintf(stderr, "short read %s\n", p); exit(2); }
  return v;
}
template <typename T>
static void wr(const char *p, const std::vector<T> &v) {
  std::ofstream o(p, std::ios::binary);
  o.write((const char *)v.data(), v.size() * sizeof(T));
}

int main(int argc, char **argv) {
  std::string c = argc > 1 ? argv[1] : "";
  if (c == "rotate") {  // rotate <dim> <flip.bin> <in.f32> <n> <out.f32>   (flip.bin is created if missing)
    size_t dim = atoi(argv[2]), n = atoi(argv[5]);
    size_t padded = rabitqlib::round_up_to_multiple(dim, 64);
    auto *rot = rabitqlib::choose_rotator<float>(dim, rabitqlib::RotatorType::FhtKacRotator, padded);
    {
      std::ifstream f(argv[3], std::ios::binary);
      if (f.good()) rot->load(f);
      else { std::ofstream o(argv[3], std::ios::binary); rot->save(o); }
    }
    auto in = rd<float>(argv[4], n * dim);
    std::vector<float> out(n * padded);
    for (size_t i = 0; i < n; i++) rot->rotate(in.data() + i * dim, out.data() + i * padded);
    wr(argv[6], out);
    return 0;
  }
  if (c == "data") {  // data <padded> <metric 0=L2 1=IP> <rot.f32> <n> <centroid.f32> <codes.u64> <factors.f32>
    size_t padded = atoi(argv[2]), n = atoi(argv[5]);
    auto metric = atoi(argv[3]) ? rabitqlib::METRIC_IP : rabitqlib::METRIC_L2;
    auto x = rd<float>(argv[4], n * padded);
    auto cen = rd<float>(argv[6], padded);
    std::vector<uint64_t> codes(n * padded / 64);
    std::vector<float> fac(n * 3);
    for (size_t i = 0; i < n; i++)
      rabitqlib::quant::rabitq_impl::one_bit::one_bit_compact_code<float, uint64_t>(
          x.data() + i * padded, cen.data(), padded, codes.data() + i * padded / 64, fac[i * 3], fac[i * 3 + 1], fac[i * 3 + 2], metric);
    wr(argv[7], codes);
    wr(argv[8], fac);
    return 0;
  }
  if (c == "query") {
    // query <padded> <metric> <t_const> <rotq.f32> <nq> <codes.u64> <factors.f32> <nd> <gadd.f32 nq> <gerr.f32 nq>
    //       <out_q.f32: nq x 3 {delta, vl, k1xsumq}> <out_bins.u64: nq x padded/64*4> <out_est.f32: nq x nd x 3>
    size_t padded = atoi(argv[2]), nq = atoi(argv[6]), nd = atoi(argv[9]);
    size_t metric = atoi(argv[3]) ? rabitqlib::METRIC_IP : rabitqlib::METRIC_L2;
    rabitqlib::quant::RabitqConfig cfg;
    cfg.t_const = atof(argv[4]);
    auto q = rd<float>(argv[5], nq * padded);
    auto codes = rd<uint64_t>(argv[7], nd * padded / 64);
    auto fac = rd<float>(argv[8], nd * 3);
    auto gadd = rd<float>(argv[10], nq), gerr = rd<float>(argv[11], nq);
    const size_t rec = padded / 8 + 12;
    std::vector<char> bin(nd * rec);  // ConstBinDataMap layout: code bytes, then f_add, f_rescale, f_error
    for (size_t j = 0; j < nd; j++) {
      memcpy(bin.data() + j * rec, codes.data() + j * padded / 64, padded / 8);
      memcpy(bin.data() + j * rec + padded / 8, fac.data() + j * 3, 12);
    }
    std::vector<float> oq(nq * 3), oe(nq * nd * 3);
    std::vector<uint64_t> ob(nq * padded / 64 * 4);
    for (size_t i = 0; i < nq; i++) {
      rabitqlib::SplitSingleQuery<float> w(q.data() + i * padded, padded, 3, cfg, metric);
      oq[i * 3] = w.delta(); oq[i * 3 + 1] = w.vl(); oq[i * 3 + 2] = w.k1xsumq();
      memcpy(ob.data() + i * padded / 64 * 4, w.query_bin(), padded / 64 * 4 * 8);
      for (size_t j = 0; j < nd; j++) {
        float ip, est, low;
        rabitqlib::split_single_estdist(bin.data() + j * rec, w, padded, ip, est, low, gadd[i], gerr[i]);
        oe[(i * nd + j) * 3] = ip; oe[(i * nd + j) * 3 + 1] = est; oe[(i * nd + j) * 3 + 2] = low;
      }
    }
    wr(argv[12], oq); wr(argv[13], ob); wr(argv[14], oe);
    return 0;
  }
  if (c == "cent") {
    // cent <padded> <rotq.f32> <nq> <cent.f32> <ncl> <out.f32: nq x ncl x 2 {euclidean_sqr, dot_product}>
    // the two reductions HierarchicalNSWSlimQ::searchKnn takes q_to_centroids from (hnswalg_slimq.h:1823-1848):
    // rabitqlib::euclidean_sqr / dot_product (rabitqlib/utils/space.hpp:226-253, Eigen .dot())
    size_t padded = atoi(argv[2]), nq = atoi(argv[4]), ncl = atoi(argv[6]);
    auto q = rd<float>(argv[3], nq * padded);
    auto ce = rd<float>(argv[5], ncl * padded);
    std::vector<float> out(nq * ncl * 2);
    for (size_t i = 0; i < nq; i++)
      for (size_t j = 0; j < ncl; j++) {
        out[(i * ncl + j) * 2] = rabitqlib::euclidean_sqr(q.data() + i * padded, ce.data() + j * padded, padded);
        out[(i * ncl + j) * 2 + 1] = rabitqlib::dot_product(q.data() + i * padded, ce.data() + j * padded, padded);
      }
    wr(argv[7], out);
    return 0;
  }
  if (c == "tconst") {  // tconst <padded> : one draw of faster_config(padded, 4).t_const (random, rabitq.hpp:27-34)
    auto cfg = rabitqlib::quant::faster_config(atoi(argv[2]), 4);
    printf("%.17g\n", cfg.t_const);
    return 0;
  }
  if (c == "buffer") {
    // buffer <cap> <n> <op.u8> <ids.u32> <dist.f32> <out_ev.u32> <out_final.bin>
    // rabitqlib::buffer::SearchBuffer (rabitqlib/utils/buffer.hpp:16-100; hnswalg_slimq.h:80-151 is the same class with the
    // is_full test hoisted to the caller) driven by an op sequence: 1 = insert (no-op when is_full(dist)), 0 = pop if has_next.
    // out_ev[i]: 1/0 inserted or not, resp. the popped id (0xFFFFFFFF when nothing to pop); out_final: u32 size, then size x {u32 id, f32 dist}.
    size_t cap = atoi(argv[2]), n = atoi(argv[3]);
    auto op = rd<uint8_t>(argv[4], n);
    auto ids = rd<uint32_t>(argv[5], n);
    auto ds = rd<float>(argv[6], n);
    rabitqlib::buffer::SearchBuffer<float> buf(cap);
    std::vector<uint32_t> ev(n);
    size_t size = 0;
    for (size_t i = 0; i < n; i++) {
      if (op[i]) {
        if (buf.is_full(ds[i])) { ev[i] = 0; continue; }
        buf.insert(ids[i], ds[i]);
        ev[i] = 1;
        if (size < cap) size++;
      } else {
        ev[i] = buf.has_next() ? buf.pop() : 0xFFFFFFFFu;
      }
    }
    wr(argv[7], ev);
    std::ofstream o(argv[8], std::ios::binary);
    uint32_t sz = (uint32_t)size;
    o.write((char *)&sz, 4);
    for (size_t i = 0; i < size; i++) {
      uint32_t id = buf.data()[i].id;
      float dd = buf.data()[i].distance;
      o.write((char *)&id, 4); o.write((char *)&dd, 4);
    }
    return 0;
  }
  if (c == "hnsw") {
    // hnsw <n> <dim> <metric 0=L2 1=IP> <M> <efC> <seed> <base.f32> <out.u32>
    // One centroid (row 0) and cluster id 0 for every point: the edges come from the raw rows alone (hnsw.hpp:381-387, 696);
    // num_threads = 1 -> parallel_for's serial loop (ivf/initializer.hpp:23-26).  out: maxlevel, enterpoint, then per node in id
    // order: label, level, and per level 0..level: count, ids.
    size_t n = atol(argv[2]), dim = atol(argv[3]);
    int metric = atoi(argv[4]);
    size_t M = atol(argv[5]), efc = atol(argv[6]), seed = atol(argv[7]);
    auto base = rd<float>(argv[8], n * dim);
    rabitqlib::hnsw::HierarchicalNSW h(n, dim, 4, M, efc, seed, metric ? rabitqlib::METRIC_IP : rabitqlib::METRIC_L2);
    std::vector<rabitqlib::PID> cid(n, 0);
    h.construct(1, base.data(), n, base.data(), cid.data(), 1, true);
    std::vector<uint32_t> out;
    out.push_back((uint32_t)h.maxlevel_);
    out.push_back((uint32_t)h.enterpoint_node_);
    for (size_t i = 0; i < n; i++) {
      out.push_back((uint32_t)h.get_external_label(i));
      const int lv = h.element_levels_[i];
      out.push_back((uint32_t)lv);
      for (int l = 0; l <= lv; l++) {
        const rabitqlib::PID *ll = l == 0 ? h.get_linklist0(i) : h.get_linklist(i, l);
        const size_t cnt = h.get_list_count(ll);
        out.push_back((uint32_t)cnt);
        for (size_t j = 0; j < cnt; j++) out.push_back(ll[1 + j]);
      }
    }
    wr(argv[9], out);
    return 0;
  }
  fprintf(stderr, "usage: ref_rabitq rotate|data|query|cent|tconst|buffer|hnsw ...\n");
  return 2;
}
