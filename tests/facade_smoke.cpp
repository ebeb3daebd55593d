// tests/facade_smoke.cpp -- a caller written against the reference's hnswlib API, compiled against the
// facade instead (hnsw-slim_amd/hnswlib/hnswlib_amd.h).  Mirrors include/strategy/hnsw_slim_strategy.h:
// L2Space(dim) -> HierarchicalNSWSlim(space, path) -> setEf -> searchKnn(q, K, out) per query.
// usage: facade_smoke <mode> ...
//   errors                      : CPU-safe checks of the reference's error conventions
//   slim  <slim.bin> <dim> <queries.f32> <nq> <k> <ef> <out.u32>      : per-query searchKnn + batch
//   hnsw  <hnsw.bin> <dim> <queries.f32> <nq> <k> <ef> <out.bin>      : priority_queue overload
//   bf    <base.f32> <dim> <queries.f32> <nq> <k> <n> <out.bin>         : BruteforceSearch addPoint + searchKnn
//   build <base.f32> <dim> <queries.f32> <nq> <k> <ef> <out.u32> <n> <hnsw_out> <slim_out>   : the build-then-search
//         sequence of include/strategy/hnsw_strategy.h:24-40 + hnsw_slim_strategy.h:83-103 (ctor, addPoint loop,
//         saveIndex, convertFromHNSW, saveIndex, setEf, searchKnn)
//   slimq <slimq.bin> <dim> <queries.f32> <nq> <k> <ef> <out.u32> <base.f32> <n>   : HierarchicalNSWSlimQ, the
//         call sequence of include/strategy/hnsw_slimq_strategy.h:72,142-156
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../hnsw-slim_amd/hnswlib/hnswlib_amd.h"

static std::vector<float> read_f32(const char *p, size_t n) {
  std::vector<float> v(n);
  std::ifstream in(p, std::ios::binary);
  in.read((char *)v.data(), n * 4);
  return v;
}

int main(int argc, char **argv) {
  std::string mode = argc > 1 ? argv[1] : "";
  if (mode == "errors") {
    hnswlib::L2Space space(32);
    int ok = 0;
    try {
      hnswlib::HierarchicalNSWSlim<float> ix(&space, "/nonexistent/slim.bin");
    } catch (std::runtime_error &e) {
      // without a GPU the device check comes first; with one, the reference's message
      ok += (std::string(e.what()) == "Cannot open file" || std::string(e.what()).find("no HIP device") != std::string::npos);
    }
    try {
      hnswlib::HierarchicalNSWSlim<float> ix(&space);
      ix.addPoint(nullptr, 0);
    } catch (std::runtime_error &e) {
      ok += std::string(e.what()) == "HierarchicalNSWSlim does not support addPoint";
    }
    {
      hnswlib::InnerProductSpace ip7(7);  // every dim, as in the reference (space_ip.h:374-382): 1 - sum i*2i, i<7
      float u[7], v[7];
      for (int i = 0; i < 7; i++) { u[i] = i; v[i] = 2 * i; }
      ok += ip7.get_dist_func()(u, v, ip7.get_dist_func_param()) == 1.0f - 182.0f;
    }
    float a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = i; b[i] = 2 * i; }
    hnswlib::L2Space s16(16);
    ok += s16.get_dist_func()(a, b, s16.get_dist_func_param()) == 1240.0f;  // sum i^2, i<16
    printf("errors ok=%d/4\n", ok);
    return ok == 4 ? 0 : 1;
  }
  if (argc < 9) return 2;
  const char *path = argv[2];
  size_t dim = atoi(argv[3]), nq = atoi(argv[5]), k = atoi(argv[6]), ef = atoi(argv[7]);
  auto Q = read_f32(argv[4], nq * dim);
  hnswlib::L2Space space(dim);
  std::ofstream out(argv[8], std::ios::binary);
  if (mode == "build") {
    if (argc < 12) return 2;
    const size_t n = atoll(argv[9]);
    auto B = read_f32(path, n * dim);
    hnswlib::HierarchicalNSW<float> hnsw(&space, n, 16, 100, "4");
    for (size_t i = 0; i < n; i++) hnsw.addPoint(B.data() + i * dim, i);
    hnsw.saveIndex(argv[10]);
    hnswlib::HierarchicalNSWSlim<float> slim(&space, n, 16, 100);
    slim.convertFromHNSW(&hnsw);
    slim.saveIndex(argv[11]);
    slim.setEf(ef);
    slim.setExactOrder(true);
    std::vector<hnswlib::tableint> one(k);
    for (size_t i = 0; i < nq; i++) {
      slim.searchKnn(Q.data() + i * dim, k, one.data());
      out.write((char *)one.data(), 4 * k);
    }
    // the vanilla object answers too (farthest-first priority_queue)
    auto r = hnsw.searchKnn(Q.data(), k);
    uint32_t c = r.size();
    out.write((char *)&c, 4);
    return 0;
  }
  if (mode == "bf") {
    const size_t n = ef;  // 7th argument is the row count in this mode
    auto B = read_f32(path, n * dim);
    hnswlib::BruteforceSearch<float> bf(&space, n);
    for (size_t i = 0; i < n; i++) bf.addPoint(B.data() + i * dim, 1000 + 3 * i);
    for (size_t i = 0; i < nq; i++) {
      auto r = bf.searchKnn(Q.data() + i * dim, k);   // farthest on top, as the reference's priority_queue
      while (!r.empty()) { uint64_t l = r.top().second; float d = r.top().first; out.write((char *)&d, 4); out.write((char *)&l, 8); r.pop(); }
    }
    return 0;
  }
  if (mode == "slim2") {
    // two replicas (here: device 0 listed twice, the one-GPU rehearsal of a multi-device node): searchKnnBatch shards the
    // batch over them and gathers; the strict order of the single-device call must come back
    hnswlib::HierarchicalNSWSlim<float> ix(&space);
    ix.setDevices({0, 0});
    ix.loadIndex(path, &space);
    ix.setEf(ef);
    ix.setExactOrder(true);
    if (!ix.sharded()) return 3;
    std::vector<hnswlib::tableint> all(nq * k);
    ix.searchKnnBatch(Q.data(), nq, k, all.data());
    out.write((char *)all.data(), 4 * nq * k);
    return 0;
  }
  if (mode == "slim") {
    hnswlib::HierarchicalNSWSlim<float> ix(&space, path);
    ix.setEf(ef);
    ix.setExactOrder(true);
    std::vector<hnswlib::tableint> one(k), all(nq * k);
    for (size_t i = 0; i < nq; i++) {  // the reference's serial query loop (hnsw_slim_strategy.h:112-114)
      ix.searchKnn(Q.data() + i * dim, k, one.data());
      out.write((char *)one.data(), 4 * k);
    }
    ix.searchKnnBatch(Q.data(), nq, k, all.data());
    out.write((char *)all.data(), 4 * nq * k);
  } else if (mode == "slimq") {
    if (argc < 11) return 2;
    const size_t n = atoll(argv[10]);
    auto B = read_f32(argv[9], n * dim);
    std::vector<std::vector<float>> data_set(n);
    for (size_t i = 0; i < n; i++) data_set[i].assign(B.begin() + i * dim, B.begin() + (i + 1) * dim);
    hnswlib::HierarchicalNSWSlimQ<float> ix(&space, path);
    ix.setDataset(&data_set);
    ix.setEf(ef);
    std::vector<hnswlib::tableint> one(k), all(nq * k);
    for (size_t i = 0; i < nq; i++) {  // hnsw_slimq_strategy.h:151-153
      ix.searchKnn(Q.data() + i * dim, k, one.data());
      out.write((char *)one.data(), 4 * k);
    }
    ix.searchKnnBatch(Q.data(), nq, k, all.data());
    out.write((char *)all.data(), 4 * nq * k);
  } else {
    hnswlib::HierarchicalNSW<float> ix(&space, path);
    ix.setEf(ef);
    for (size_t i = 0; i < nq; i++) {
      auto r = ix.searchKnnCloserFirst(Q.data() + i * dim, k);
      uint32_t c = r.size();
      out.write((char *)&c, 4);
      for (auto &p : r) { uint64_t l = p.second; out.write((char *)&p.first, 4); out.write((char *)&l, 8); }
    }
  }
  return 0;
}
