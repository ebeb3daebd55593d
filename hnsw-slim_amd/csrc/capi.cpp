// capi.cpp -- the C ABI (include/hnsw_slim_amd.h) over the HIP search engine.
// Compiled by hipcc together with beam_search.hip into libhnsw_slim_amd.so.  No CPU search path
// exists in this library: without a HIP device every search call returns HS_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hnsw_slim_amd.h"
#include "engine.hpp"
#define HS_HAVE_GPU_CONVERT 1
#include "convert_engine.hpp"
#include "host_graph.hpp"
#include "rabitq_est.hpp"
#include "rabitq_host.hpp"
#include "slimq_engine.hpp"
#include "bf_engine.hpp"

using namespace hs;

static thread_local std::string g_err;
static hs_status fail(hs_status s, const std::string &msg) {
  g_err = msg;
  return s;
}
static hs_status from_exception(const std::exception &e) {
  std::string m = e.what();
  if (m == "Cannot open file") return fail(HS_ERR_IO, m);
  if (m.find("corrupted") != std::string::npos) return fail(HS_ERR_CORRUPT, m);
  if (m.find("Not enough memory") != std::string::npos) return fail(HS_ERR_NOMEM, m);
  if (m.find("supports dim") != std::string::npos || m.find("SlimQ supports") != std::string::npos) return fail(HS_ERR_UNSUPPORTED, m);
  return fail(HS_ERR_INVALID, m);
}
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return fail(HS_ERR_DEVICE, std::string(#expr ": ") + hipGetErrorString(_e)); \
  } while (0)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    if (count == 0) return hipSuccess;
    const hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;   // (a failed allocation leaves the buffer empty: the next ensure() tries again)
    else p = nullptr;
    return e;
  }
  hipError_t ensure(size_t count) { return count <= n ? hipSuccess : alloc(count); }
  hipError_t upload(const std::vector<T> &v) {
    hipError_t e = alloc(std::max<size_t>(v.size(), 1));
    if (e != hipSuccess) return e;
    return v.empty() ? hipSuccess : hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

struct hs_index {
  int device = 0;
  hs_info info{};
  size_t ef = 10;  // hnswalg.h:864, hnswalg_slim.h:793
  uint32_t user_cand_cap = 0, user_hash_slots = 0;
  uint32_t grow_cand = 0, grow_hash = 0;  // adaptive: doublings learnt from earlier batches' overflow counts
  bool exact_order = false;               // always use the strict kernel (reference output order)
  const char *last_kernel = "";           // the kernel that served pass 0 of the most recent search call (hs_last_kernel)
  // patching (hs_index_patch): a Slim index loaded with max_elements > count keeps its host image and has row capacity
  std::unique_ptr<SlimGraph> host_slim;
  size_t cap_rows = 0;
  DevIndex dev{};
  DevBuf<float> vec;
  DevBuf<uint32_t> row_ptr0, cols, up_base, up_ptr, tile0, uptile;
  DevBuf<uint64_t> labels;
  DevBuf<uint8_t> deleted;
  // per-stream scratch (grow-only): calls on different HIP streams may be in flight together
  struct StreamWs {
    DevBuf<uint32_t> spill;             // visited-set tier 2, nq x kSpillSlots
    DevBuf<uint32_t> prep;              // SlimQ: per-query preparation records
    DevBuf<uint32_t> status, counters;  // counters: 3 passes x 4 {visited overflow, candidate overflow, tie hazard, tier-2 spills}
    DevBuf<uint32_t> entry, order;      // two-launch fast pass: level-0 entries (nq x 4 words) and the start order
    DevBuf<uint32_t> fb;                // last-resort pass: kFbGrid x (candidate heap + tier-2 visited set)
    uint32_t oflip = 0;
    size_t last_nq = 0;
    // hs_search_batch_async: device staging of the queries and outputs of the call in flight on this stream
    DevBuf<float> aq, adist;
    DevBuf<uint32_t> al32, acnt, astats;
    DevBuf<uint64_t> al64;
  };
  std::map<hipStream_t, std::unique_ptr<StreamWs>> ws;
  std::mutex ws_mu;
  StreamWs *stream_ws(hipStream_t st) {
    std::lock_guard<std::mutex> g(ws_mu);
    auto &p = ws[st];
    if (!p) p.reset(new StreamWs());
    return p.get();
  }
  // host-pointer API staging (default stream)
  DevBuf<float> wq, wdist;
  DevBuf<uint32_t> wl32, wcnt, wstats, wrawsz;
  DevBuf<uint64_t> wl64;
  DevBuf<Pair> wraw;
  std::vector<uint64_t> host_labels;   // external labels by internal id
  std::vector<uint8_t> host_deleted;   // delete marks by internal id
  DevBuf<uint8_t> wexcl;               // deleted | !allowed of the current filtered call
  // HNSW-SlimQ (kind == HS_KIND_SLIMQ): RaBitQ records, rotated centroids, rotator flips; `vec` then holds the
  // dataset rows of hs_slimq_set_dataset()
  DevBuf<uint32_t> q_rec, q_ftile, q_uptile;
  DevBuf<float> q_cent;
  DevBuf<uint8_t> q_flips;
  DevSlimQ sq{};
  bool has_dataset = false;
  uint32_t *trace_ptr = nullptr;   // hs_slimq_trace only
  uint32_t trace_cap = 0;
};

static uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

struct Shape {
  uint32_t ef, cand_cap, cand_cap_fast, hash_slots;
  uint32_t q_hash_slots, q_bits;        // fast kernel: visited-set tier 1 in 16-bit slots (LDS words, id-space width; 0 = not applicable)
  uint32_t l_cand_cap, l_hash_slots;    // lean kernel
  uint32_t fb_cand_cap, fb_hash_slots;  // last-resort pass (one workgroup per CU, whole LDS)
};
static constexpr size_t kLdsPerCU = 160 * 1024;
// Last-resort pass (strict kernel): a few workgroups, each with a small visited hash in LDS and its candidate heap + a large
// tier-2 visited set in its own region of global memory (engine.hpp fb_cand / fb_spill): 24 MiB per stream, any query fits.
static constexpr uint32_t kFbGrid = 32, kFbCand = 32768, kFbSpill = 131072, kFbHash = 2048;
static constexpr uint32_t kLeanMinEf = 64;    // HS_KERNEL=lean: the lean kernel answers from this ef upwards (HS_LEAN_MIN_EF overrides), the fast kernel below
static constexpr uint32_t kSpillSlots = 8192;  // 32 KiB per query of tier-2 visited set
static constexpr uint32_t kCand2Cap = 4096;    // 32 KiB per query of tier-2 candidate heap
static constexpr uint32_t kLogCap = 4096;      // 32 KiB per query: result-set insertion log (tie replay)
static constexpr uint32_t kHopCap = 4096;      // 4 KiB per query: accepted neighbours per expansion (flat start of the fast kernel)
static constexpr uint32_t kParkWords = 2048;   // 8 KiB per query: where the flat kernel parks the head of its visited set during a heap replay
// words of scratch per query; beyond ef = 256 (the flat kernel's S = 6, 8 shapes: up to ~3 k accepted neighbours and > 1 k hops per
// query at ef = 512) the insertion and hop logs are twice / four times as long
static uint32_t log_cap_for(uint32_t ef) { return ef <= 256 ? kLogCap : 2 * kLogCap; }
static uint32_t hop_cap_for(uint32_t ef) { return ef <= 256 ? kHopCap : 4 * kHopCap; }
static uint32_t spill_stride_for(uint32_t ef) { return kSpillSlots + 2 * kCand2Cap + 2 * log_cap_for(ef) + hop_cap_for(ef) / 4 + kParkWords; }

static hs_status plan_shape(const hs_index *ix, size_t k, Shape &s) {
  const size_t ef = std::max(ix->ef, k);
  if (ef > (1u << 20)) return fail(HS_ERR_INVALID, "ef too large");
  s.ef = (uint32_t)ef;
  // LDS share of the candidate heap: must cover essentially every query (peak heap size on the bench data:
  // 2.4 ef median, 4.4 ef + 40 at p99.9) -- tier 2 is a safety net, a few %% of queries living in it already
  // cost 15-40 %% of throughput (profiles/r01_tier2_cost.txt)
  s.cand_cap_fast = ix->user_cand_cap ? ix->user_cand_cap : (uint32_t)(4.4 * ef + 64);
  s.cand_cap_fast = (s.cand_cap_fast + 1) & ~1u;
  // the strict kernel keeps its whole heap in LDS: cover the observed maximum (4.9 ef + margin)
  s.cand_cap = ix->user_cand_cap ? ix->user_cand_cap : (uint32_t)((3 * ef + 256) << ix->grow_cand);
  s.cand_cap = (s.cand_cap + 1) & ~1u;
  // tier-1 visited set: sized so that most queries never leave LDS (75 % fill); the rest spill to tier 2
  const uint32_t want = (uint32_t)((450 + 5 * ef) * (1.0 + 0.25 * ix->grow_hash) / 0.75);
  s.hash_slots = ix->user_hash_slots ? (ix->user_hash_slots + 63) / 64 * 64 : (want + 63) / 64 * 64;
  // Fast kernel: the same LDS holds twice the ids as 16-bit remainders of a bijective hash, eight to a 16-byte bucket, no
  // probing (csrc/search_common.hpp).  Buckets: a power of two with the expected number of visited ids filling them to
  // 5 of 8 on average (4.5 would double the table at ef=384 for nothing: measured); usable while the id space is at most 16 bits
  // wider than the bucket index.
  {
    const uint32_t n_vis = ix->user_hash_slots ? ix->user_hash_slots : (uint32_t)((450 + 5 * ef) * (1.0 + 0.25 * ix->grow_hash));
    uint32_t nb = 4;
    while (nb * 5.0 < n_vis && nb < (1u << 14)) nb <<= 1;
    uint32_t bbits = 0, idbits = 1;
    while ((1u << bbits) < nb) bbits++;
    while (idbits < 32 && ((uint64_t)1 << idbits) < (uint64_t)std::max<size_t>(ix->info.n, 2)) idbits++;
    const uint32_t B = std::max(idbits, bbits);
    static const bool off = getenv("HS_VIS16") && atoi(getenv("HS_VIS16")) == 0;   // diagnostic: the 32-bit form everywhere
    s.q_hash_slots = nb * 4;
    s.q_bits = (!off && B - bbits <= 16 && B < 32) ? B : 0;
  }
  const uint32_t dim = (uint32_t)ix->info.dim;
  // Lean kernel (large ef): candidate heap ~p99 of its peak size, visited set ~p90 of the distance evaluations at an 87.5 % fill
  // limit (measured on the 1M SIFT-like bench index, ef 32..256); the rest continue in their tier-2 regions.
  {
    uint32_t lc = ix->user_cand_cap ? ix->user_cand_cap : (uint32_t)((2.5 * ef + 130) * (1u << ix->grow_cand));
    s.l_cand_cap = std::max<uint32_t>((lc + 1) & ~1u, 16);
    const uint32_t lh = ix->user_hash_slots ? ix->user_hash_slots : (uint32_t)((520 + 5 * ef) * (1.0 + 0.25 * ix->grow_hash) / 0.875);
    s.l_hash_slots = (lh + 63) / 64 * 64;
  }
  // shrink the first-pass shape if it does not fit one CU at all
  while (strict_lds_bytes(dim, s.ef, s.cand_cap, s.hash_slots) > kLdsPerCU && s.hash_slots > 256) s.hash_slots >>= 1;
  while (strict_lds_bytes(dim, s.ef, s.cand_cap, s.hash_slots) > kLdsPerCU && s.cand_cap > 128) s.cand_cap = (s.cand_cap / 2 + 1) & ~1u;
  if (strict_lds_bytes(dim, s.ef, s.cand_cap, s.hash_slots) > kLdsPerCU)
    return fail(HS_ERR_CAPACITY, "ef/dim do not fit the 160 KiB LDS of one CU");
  // last resort (see kFbGrid): LDS = query + result array + a kFbHash-slot visited hash; everything else in global memory.
  // (It goes out with every batch and is normally empty; when it asked for 64 KiB -- before that for a whole CU -- it waited 0.56 ms
  // on average for that much LDS to drain behind the other streams' launches, profiles/r03_kernel_stats_pipelined.csv.)
  s.fb_cand_cap = kFbCand;
  s.fb_hash_slots = kFbHash;
  while (strict_lds_bytes(dim, s.ef, 0, s.fb_hash_slots) > kLdsPerCU && s.fb_hash_slots > 256) s.fb_hash_slots >>= 1;
  return HS_OK;
}

// Flat kernel (flat_search.hip): visited-set buckets, the division constants of bucket = h mod nb, and the LDS share of its
// (lazily replayed) candidate heap.  The bucket count takes whatever LDS the wave's residency granule leaves unused: 5 wavefronts
// per SIMD = 20 workgroups per CU = 8 KiB each on the common shapes (flatk_waves_per_cu).
struct FlatPlan { uint32_t nb, mul, sh, vis_bits; bool ok; };
static FlatPlan plan_flat(const hs_index *ix, uint32_t ef, size_t nq) {
  FlatPlan f{};
  const uint32_t dim = (uint32_t)ix->info.dim;
  uint32_t idbits = 1;
  while (idbits < 32 && ((uint64_t)1 << idbits) < (uint64_t)std::max<size_t>(ix->info.n, 2)) idbits++;
  f.vis_bits = idbits;
  // expected visited ids per query (distance evaluations, measured on the 1M SIFT-like bench index: 450 + 5 ef), 3.2 per bucket of 7
  uint32_t nb = ix->user_hash_slots ? std::max<uint32_t>(ix->user_hash_slots / 4, 8) : (uint32_t)((450 + 5.0 * ef) * (1.0 + 0.25 * ix->grow_hash) / 3.2);
  nb = std::max<uint32_t>(nb, 8);
  if (!ix->user_hash_slots) {
    const size_t total = flatk_lds_bytes(dim, ef, nb);
    static const size_t env_waves = getenv("HS_FLAT_WAVES_PER_CU") ? (size_t)atoi(getenv("HS_FLAT_WAVES_PER_CU")) : 0;   // diagnostic: builds with another residency
    const size_t max_waves = env_waves ? env_waves : flatk_waves_per_cu(dim, ef);
    size_t waves = std::min<size_t>(max_waves, kLdsPerCU / std::max<size_t>(total, 1));
    // (Until round 3 a launch smaller than the wave slots took fewer, larger shares.  Measured, profiles/r03_small_launch_lds_share_ab.log:
    //  nothing gained on a single small launch -- 1250 SIFT queries 0.705 vs 0.655 ms, 1000 GIST queries 3.420 vs 3.416 ms -- and with
    //  16 such launches in flight the larger shares cap the residency: 416 k vs 520 k q/s.  The share is the full-residency one.)
    if (waves >= 1) {
      const size_t share = std::min<size_t>((kLdsPerCU / waves) & ~size_t(15), 64 * 1024);
      if (share > total) nb += (uint32_t)((share - total) / 16);
    }
  }
  nb = std::min<uint32_t>(nb, 1u << 14);
  // remainders h div nb must fit 15 bits
  while (((uint64_t)1 << idbits) / nb > 32767 && nb < (1u << 16)) nb += nb / 2;
  if (((uint64_t)1 << idbits) / nb > 32767 || idbits > 31) return f;
  uint32_t sh = 0;
  while ((2u << sh) <= nb) sh++;   // floor(log2(nb))
  uint64_t m = (((uint64_t)1 << (32 + sh)) + nb - 1) / nb;
  if (m >> 32) { sh--; m = (((uint64_t)1 << (32 + sh)) + nb - 1) / nb; }
  f.nb = nb; f.mul = (uint32_t)m; f.sh = sh;
  f.ok = flatk_lds_bytes(dim, ef, nb) <= kLdsPerCU;
  return f;
}

template <typename T>
static hipError_t upload_cap(DevBuf<T> &b, const std::vector<T> &v, size_t cap) {
  hipError_t e = b.alloc(std::max<size_t>(std::max(v.size(), cap), 1));
  if (e != hipSuccess) return e;
  return v.empty() ? hipSuccess : hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

extern "C" {

const char *hs_last_error(void) { return g_err.c_str(); }

int hs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// upper-level tiles for the greedy descent (see engine.hpp): slot t = up_ptr entry t, {neighbour id, the neighbour's up_base}
static std::vector<uint32_t> build_uptile(const PackedIndex &p, uint32_t &up_stride) {
  up_stride = 0;
  std::vector<uint32_t> ut;
  size_t max_up = 0;
  for (size_t t = 0; t + 1 < p.up_ptr.size(); t++)
    if (p.up_ptr[t + 1] > p.up_ptr[t]) max_up = std::max<size_t>(max_up, p.up_ptr[t + 1] - p.up_ptr[t]);
  if (p.up_ptr.empty() || max_up > 64) return ut;
  up_stride = std::max<uint32_t>(16, (uint32_t)((max_up + 15) / 16 * 16));
  ut.assign(p.up_ptr.size() * (size_t)up_stride * 2, 0xFFFFFFFFu);
  // a node with L upper levels owns L+1 consecutive up_ptr entries (L list starts + one terminator), in id order
  std::vector<uint32_t> owners;
  for (size_t i = 0; i < p.n; i++)
    if (p.up_base[i] != PackedIndex::NONE) owners.push_back((uint32_t)i);
  for (size_t o = 0; o < owners.size(); o++) {
    const uint32_t b0 = p.up_base[owners[o]];
    const uint32_t end = o + 1 < owners.size() ? p.up_base[owners[o + 1]] : (uint32_t)p.up_ptr.size();
    for (uint32_t t = b0; t + 1 < end; t++) {
      const uint32_t s0 = p.up_ptr[t], e0 = p.up_ptr[t + 1];
      for (uint32_t j = 0; j < e0 - s0; j++) {
        const uint32_t nb = p.cols[s0 + j];
        ut[((size_t)t * up_stride + j) * 2] = nb;
        ut[((size_t)t * up_stride + j) * 2 + 1] = p.up_base[nb];
      }
    }
  }
  return ut;
}
static uint32_t tile_stride_for(size_t max_deg0) { return max_deg0 <= 64 ? std::max<uint32_t>(16, (uint32_t)((max_deg0 + 15) / 16 * 16)) : 0; }

// the graph-structure arrays that are small next to the vectors and tiles: uploaded whole (also after a patch)
static hs_status upload_small(hs_index *ix, const PackedIndex &p) {
  HIP_TRY(ix->row_ptr0.upload(p.row_ptr0));
  HIP_TRY(ix->cols.upload(p.cols));
  HIP_TRY(upload_cap(ix->up_base, p.up_base, ix->cap_rows));
  HIP_TRY(ix->up_ptr.upload(p.up_ptr));
  uint32_t up_stride = 0;
  const std::vector<uint32_t> ut = build_uptile(p, up_stride);
  if (up_stride) HIP_TRY(ix->uptile.upload(ut));
  DevIndex &d = ix->dev;
  d.row_ptr0 = ix->row_ptr0.p; d.cols = ix->cols.p; d.up_base = ix->up_base.p; d.up_ptr = ix->up_ptr.p;
  d.uptile = up_stride ? reinterpret_cast<const uint2 *>(ix->uptile.p) : nullptr; d.up_stride = up_stride;
  d.ep_base = p.n ? p.up_base[p.enterpoint] : 0xFFFFFFFFu;
  d.n = (uint32_t)p.n; d.dim = (uint32_t)p.dim; d.maxlevel = p.maxlevel; d.threshold_level = p.threshold_level;
  d.enterpoint = p.enterpoint; d.has_deleted = p.has_deleted; d.kind = p.kind; d.metric = p.metric;
  hs_info &i = ix->info;
  i.n = p.n; i.dim = p.dim; i.kind = p.kind; i.metric = p.metric; i.maxlevel = p.maxlevel;
  i.threshold_level = p.threshold_level; i.enterpoint = p.enterpoint; i.has_deleted = p.has_deleted;
  i.n_edges = p.cols.size(); i.max_degree0 = p.max_deg0; i.index_size = p.index_size;
  i.device_bytes = p.vec.size() * 4 + (p.row_ptr0.size() + p.cols.size() + p.up_base.size() + p.up_ptr.size()) * 4 +
                   p.labels.size() * 8 + p.deleted.size() + (size_t)p.n * ix->dev.tile_stride * 4;
  return HS_OK;
}

static hs_status upload(hs_index *ix, const PackedIndex &p) {
  HIP_TRY(hipSetDevice(ix->device));
  const size_t cap = std::max(ix->cap_rows, p.n);
  HIP_TRY(upload_cap(ix->vec, p.vec, cap * p.dim));
  // level-0 adjacency tiles: node i's ids padded with 0xFFFFFFFF to a fixed, 64-byte-multiple stride
  const uint32_t stride = tile_stride_for(p.max_deg0);
  if (stride) {
    std::vector<uint32_t> tile((size_t)p.n * stride, 0xFFFFFFFFu);
    for (size_t i = 0; i < p.n; i++)
      std::copy(p.cols.begin() + p.row_ptr0[i], p.cols.begin() + p.row_ptr0[i + 1], tile.begin() + i * stride);
    HIP_TRY(upload_cap(ix->tile0, tile, cap * stride));
  }
  HIP_TRY(upload_cap(ix->labels, p.labels, cap));
  HIP_TRY(upload_cap(ix->deleted, p.deleted, cap));
  ix->host_labels = p.labels;
  ix->host_deleted = p.deleted;
  DevIndex &d = ix->dev;
  d.vec = ix->vec.p; d.labels = ix->labels.p; d.deleted = ix->deleted.p;
  d.tile0 = stride ? ix->tile0.p : nullptr; d.tile_stride = stride;
  return upload_small(ix, p);
}

// HierarchicalNSWSlimQ::loadIndex (hnswalg_slimq.h:1218-1313): graph -> CSR/tiles, element records -> 16-byte header
// {f_add, f_rescale, cluster id, f_error} + sign code, rotated centroids and rotator flips as they are.
static hs_status load_slimq(const BinSource &src, int metric, size_t dim, int device, hs_index **out) {
  SlimQGraph q;
  PackedIndex p;
  try {
    q.load(src, metric, dim);
    if (q.rot.trunc < 64) return fail(HS_ERR_UNSUPPORTED, "SlimQ supports dim >= 64");
    p.kind = HS_KIND_SLIMQ; p.metric = (Metric)metric; p.n = q.count; p.dim = dim;
    p.maxlevel = q.maxlevel; p.threshold_level = q.threshold_level; p.enterpoint = q.enterpoint;
    p.index_size = q.count * 20;   // HierarchicalNSWSlimQ::indexSize() (hnswalg_slimq.h:2047-2057): 20 B per element + blobs
    for (size_t i = 0; i < q.count; i++) p.index_size += 2 * (size_t)q.level[i] + 4 * (size_t)q.total(i);
    p.labels = q.label;
    p.deleted.assign(q.count, 0);
    p.pack_chal([&](size_t i) { return (int)q.level[i]; }, [&](size_t i) { return (size_t)q.total(i); },
                [&](size_t i) -> const std::vector<char> & { return q.blobs[i]; });
    if (q.count && q.enterpoint >= q.count) return fail(HS_ERR_CORRUPT, "Index seems to be corrupted or unsupported");
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory: loadIndex failed to allocate");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  hs_index *ix = new hs_index();
  ix->device = device;
  hs_status s = upload(ix, p);
  if (s != HS_OK) { delete ix; return s; }
  const uint32_t nblk = (uint32_t)(q.padded / 64), rw = (4 + 2 * nblk + 3) & ~3u;  // 16-byte multiple
  std::vector<uint32_t> rec((size_t)q.count * rw);
  for (size_t i = 0; i < q.count; i++) {
    uint32_t *r = &rec[i * rw];
    memcpy(r, &q.factors[i * 3], 4); memcpy(r + 1, &q.factors[i * 3 + 1], 4);
    r[2] = q.cluster[i];
    memcpy(r + 3, &q.factors[i * 3 + 2], 4);
    memcpy(r + 4, &q.code[i * nblk], 8 * nblk);
  }
  // fused level-0 tiles (see slimq_engine.hpp): 288 GB of HBM buys one dependent access per expansion.
  // HS_SLIMQ_FUSED=0 (debug/test knob) keeps the CSR + record-array layout that wide graphs (degree > 64) fall back to.
  const char *fused_env = getenv("HS_SLIMQ_FUSED");
  const bool fused = !(fused_env && fused_env[0] == '0');
  const uint32_t stride = fused ? ix->dev.tile_stride : 0;
  std::vector<uint32_t> ft;
  if (stride) {
    ft.assign((size_t)q.count * stride * rw, 0u);
    for (size_t i = 0; i < q.count; i++) {
      uint32_t *row = &ft[i * stride * rw];
      const uint32_t deg = p.row_ptr0[i + 1] - p.row_ptr0[i];
      for (uint32_t j = 0; j < stride; j++) {
        uint32_t *r = row + (size_t)j * rw;
        if (j < deg) {
          const uint32_t nb = p.cols[p.row_ptr0[i] + j];
          memcpy(r, &rec[(size_t)nb * rw], 4 * rw);
          r[3] = nb;
        } else {
          r[3] = 0xFFFFFFFFu;
        }
      }
    }
  }
  // fused upper-level tiles: slot up_base[i] + l - 1 holds level l of node i
  uint32_t up_stride = 0;
  std::vector<uint32_t> ut;
  {
    size_t max_up = 0;
    for (size_t t = 0; t + 1 < p.up_ptr.size(); t++) max_up = std::max<size_t>(max_up, p.up_ptr[t + 1] >= p.up_ptr[t] ? p.up_ptr[t + 1] - p.up_ptr[t] : 0);
    if (fused && !p.up_ptr.empty() && max_up <= 64) {
      up_stride = std::max<uint32_t>(16, (uint32_t)((max_up + 15) / 16 * 16));
      const uint32_t urw = rw + 4;
      ut.assign(p.up_ptr.size() * (size_t)up_stride * urw, 0u);
      for (size_t t = 0; t < p.up_ptr.size(); t++)
        for (uint32_t j = 0; j < up_stride; j++) ut[(t * up_stride + j) * urw + 3] = 0xFFFFFFFFu;
      for (size_t i = 0; i < q.count; i++) {
        const uint32_t b = p.up_base[i];
        if (b == PackedIndex::NONE) continue;
        for (int l = 1; l <= q.level[i]; l++) {
          const uint32_t s0 = p.up_ptr[b + l - 1], e0 = p.up_ptr[b + l];
          for (uint32_t j = 0; j < e0 - s0; j++) {
            const uint32_t nb = p.cols[s0 + j];
            uint32_t *r = &ut[((size_t)(b + l - 1) * up_stride + j) * urw];
            memcpy(r, &rec[(size_t)nb * rw], 4 * rw);
            r[3] = nb;
            r[rw] = p.up_base[nb];
          }
        }
      }
    }
  }
  hipError_t e = ix->q_rec.upload(rec);
  if (e == hipSuccess && up_stride) e = ix->q_uptile.upload(ut);
  if (e == hipSuccess && stride) e = ix->q_ftile.upload(ft);
  if (e == hipSuccess) e = ix->q_cent.upload(q.centroids);
  if (e == hipSuccess) e = ix->q_flips.upload(q.rot.flip);
  if (e != hipSuccess) { delete ix; return fail(HS_ERR_DEVICE, std::string("SlimQ upload: ") + hipGetErrorString(e)); }
  DevSlimQ &d = ix->sq;
  d.rec = ix->q_rec.p; d.ftile = stride ? ix->q_ftile.p : nullptr; d.raw = nullptr;
  d.uptile = up_stride ? ix->q_uptile.p : nullptr; d.up_stride = up_stride;
  d.ep_base = q.count ? p.up_base[q.enterpoint] : 0xFFFFFFFFu; d.cent = ix->q_cent.p; d.flips = ix->q_flips.p;
  d.rec_words = rw; d.padded = (uint32_t)q.padded; d.trunc = (uint32_t)q.rot.trunc; d.ncl = (uint32_t)q.num_cluster;
  d.fht_scale = q.rot.fac;
  d.t_const = rq_default_tconst(q.padded, 1);
  ix->info.device_bytes += (rec.size() + ft.size() + ut.size()) * 4 + q.centroids.size() * 4 + q.rot.flip.size();
  *out = ix;
  return HS_OK;
}

static hs_status load_from(const BinSource &src, int kind, int metric, size_t dim, size_t max_elements, int device, hs_index **out) {
  if (metric != HS_METRIC_L2 && metric != HS_METRIC_IP) return fail(HS_ERR_INVALID, "bad metric");
  if (dim == 0) return fail(HS_ERR_INVALID, "dim must be > 0");
  if (hs_device_count() <= device) return fail(HS_ERR_DEVICE, "no HIP device (this library has no CPU search path)");
  PackedIndex p;
  std::unique_ptr<SlimGraph> keep_slim;
  try {
    if (kind == HS_KIND_HNSW) {
      VanillaGraph g;
      g.load(src, (Metric)metric, dim, max_elements);
      p.from_vanilla(g);
    } else if (kind == HS_KIND_SLIM) {
      std::unique_ptr<SlimGraph> g(new SlimGraph());
      g->load(src, (Metric)metric, dim);
      p.from_slim(*g);
      if (max_elements > g->count) keep_slim = std::move(g);   // room for patchFromStream (hnswalg_slim.h:760, 784)
    } else if (kind == HS_KIND_SLIMQ) {
      return load_slimq(src, metric, dim, device, out);
    } else {
      return fail(HS_ERR_INVALID, "bad index kind");
    }
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory: loadIndex failed to allocate");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  hs_index *ix = new hs_index();
  ix->device = device;
  if (keep_slim) { ix->cap_rows = max_elements; ix->host_slim = std::move(keep_slim); }
  hs_status s = upload(ix, p);
  if (s != HS_OK) { delete ix; return s; }
  *out = ix;
  return HS_OK;
}

// patchFromStream(std::istream &in, bool to_add) (hnswalg_slim.h:2292-2340) on a device-resident index.  Stream, as the
// reference's server assembles it (hnsw_slim_server_patch.cc:280-290 after its `finished` word; records by genPatch,
// hnswalg_slim.h:1427-1476):  u64 cur_element_count, u64 changed_old_cnt, u64 changed_new_cnt, then per changed node
//   u32 id | old node: 8 bytes {i32 level, u32 total_neighbor}; new node: 16 bytes {level, total, u64 label} |
//   u32 neighborsSize | the neighbour blob | new node and to_add: data_size bytes of vector.
// The host image takes the records exactly as the reference's does; on the device the two big arrays (vectors, level-0
// tiles) are rewritten only where a node changed, the small structure arrays (CSR, upper-level tiles) are rebuilt whole.
// As in the reference the stream carries no enter point / max level: they stay what they were.
hs_status hs_index_patch(hs_index *ix, const void *bytes, size_t len, int to_add) {
  if (!ix || !bytes) return fail(HS_ERR_INVALID, "null argument");
  if (!ix->host_slim) return fail(HS_ERR_INVALID, "index not patchable: load a Slim index with max_elements > its element count");
  SlimGraph &g = *ix->host_slim;
  const size_t spe = g.size_per_el, dim = g.dim;
  std::vector<uint32_t> changed;
  size_t new_count = 0;
  try {
    BinReader r(BinSource(bytes, len));
    new_count = r.pod<uint64_t>();
    const uint64_t n_old = r.pod<uint64_t>(), n_new = r.pod<uint64_t>();
    if (new_count > ix->cap_rows || new_count < g.count || n_old + n_new > (1ull << 32)) return fail(HS_ERR_CAPACITY, "patch exceeds max_elements");
    // stage the records first: a malformed stream must leave the index untouched
    struct Rec { uint32_t id; char head[16]; std::vector<char> blob; std::vector<char> vec; bool is_new; };
    std::vector<Rec> recs((size_t)(n_old + n_new));
    for (size_t i = 0; i < recs.size(); i++) {
      Rec &rc = recs[i];
      rc.is_new = i >= n_old;
      rc.id = r.pod<uint32_t>();
      r.bytes(rc.head, rc.is_new ? 16 : 8);
      const uint32_t nsz = r.pod<uint32_t>();
      int32_t level; uint32_t total;
      memcpy(&level, rc.head, 4); memcpy(&total, rc.head + 4, 4);
      if (rc.id >= new_count || level < 0 || level > 64 || (nsz != 0 && nsz != 2 * (uint32_t)level + 4 * total))
        return fail(HS_ERR_CORRUPT, "Index seems to be corrupted or unsupported");
      rc.blob.resize(nsz);
      if (nsz) r.bytes(rc.blob.data(), nsz);
      if (to_add && rc.is_new) { rc.vec.resize(dim * 4); r.bytes(rc.vec.data(), dim * 4); }
    }
    g.elements.resize(new_count * spe, 0);
    g.blobs.resize(new_count);
    for (Rec &rc : recs) {
      char *e = g.elements.data() + (size_t)rc.id * spe;
      memcpy(e, rc.head, rc.is_new ? 16 : 8);
      uint32_t total; memcpy(&total, e + 4, 4);
      g.blobs[rc.id] = (rc.blob.empty() || total == 0) ? std::vector<char>() : std::move(rc.blob);
      if (!rc.vec.empty()) memcpy(e + 24, rc.vec.data(), dim * 4);
      changed.push_back(rc.id);
    }
    g.count = new_count;
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory: patchFromStream failed to allocate linklist");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  PackedIndex p;
  try {
    p.from_slim(g);
  } catch (std::exception &e) {
    return from_exception(e);
  }
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipDeviceSynchronize());   // no search may be in flight on this index while it is rewritten
  const uint32_t stride = tile_stride_for(p.max_deg0);
  if (stride != ix->dev.tile_stride || !ix->dev.tile0) return upload(ix, p);   // a list outgrew the tile stride: re-tile everything
  std::vector<uint32_t> row(stride);
  for (uint32_t id : changed) {
    std::fill(row.begin(), row.end(), 0xFFFFFFFFu);
    std::copy(p.cols.begin() + p.row_ptr0[id], p.cols.begin() + p.row_ptr0[id + 1], row.begin());
    HIP_TRY(hipMemcpy(ix->tile0.p + (size_t)id * stride, row.data(), stride * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ix->vec.p + (size_t)id * dim, &p.vec[(size_t)id * dim], dim * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ix->labels.p + id, &p.labels[id], 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ix->deleted.p + id, &p.deleted[id], 1, hipMemcpyHostToDevice));
  }
  ix->host_labels = p.labels;
  ix->host_deleted = p.deleted;
  return upload_small(ix, p);
}

// An index the host has already parsed: node i owns levels[i] + 1 consecutive neighbour lists (level 0 first),
// list t = list_ids[list_ptr[t] .. list_ptr[t+1]).  kind selects the searchKnn semantics (HS_KIND_HNSW / HS_KIND_SLIM).
hs_status hs_index_from_host_arrays(int kind, int metric, size_t n, size_t dim, const float *vectors, const uint64_t *labels,
                                    const uint8_t *deleted, const int32_t *levels, const uint64_t *list_ptr, const uint32_t *list_ids,
                                    uint32_t enterpoint, int32_t maxlevel, int32_t threshold_level, int device, hs_index **out) {
  if (!out || (n && (!vectors || !levels || !list_ptr || !list_ids))) return fail(HS_ERR_INVALID, "null argument");
  if (kind != HS_KIND_HNSW && kind != HS_KIND_SLIM) return fail(HS_ERR_INVALID, "bad index kind");
  if (metric != HS_METRIC_L2 && metric != HS_METRIC_IP) return fail(HS_ERR_INVALID, "bad metric");
  if (dim == 0) return fail(HS_ERR_INVALID, "dim must be > 0");
  if (n && enterpoint >= n) return fail(HS_ERR_INVALID, "enter point out of range");
  if (hs_device_count() <= device) return fail(HS_ERR_DEVICE, "no HIP device (this library has no CPU search path)");
  PackedIndex p;
  try {
    p.kind = kind; p.metric = (Metric)metric; p.n = n; p.dim = dim;
    p.maxlevel = maxlevel; p.threshold_level = kind == HS_KIND_SLIM ? threshold_level : 0; p.enterpoint = enterpoint;
    p.vec.assign(vectors, vectors + n * dim);
    p.labels.resize(n);
    p.deleted.assign(n, 0);
    size_t nd = 0;
    for (size_t i = 0; i < n; i++) {
      p.labels[i] = labels ? labels[i] : (uint64_t)i;
      if (deleted) { p.deleted[i] = deleted[i] ? 1 : 0; nd += p.deleted[i]; }
    }
    p.has_deleted = nd > 0;
    p.row_ptr0.assign(n + 1, 0);
    p.up_base.assign(n, PackedIndex::NONE);
    size_t t = 0;
    std::vector<size_t> first(n);
    for (size_t i = 0; i < n; i++) {   // level-0 lists first (CSR rows), then the upper levels appended to cols
      if (levels[i] < 0 || levels[i] > maxlevel) return fail(HS_ERR_INVALID, "level out of range");
      first[i] = t;
      const uint64_t s0 = list_ptr[t], e0 = list_ptr[t + 1];
      if (e0 < s0) return fail(HS_ERR_INVALID, "list_ptr not monotone");
      for (uint64_t j = s0; j < e0; j++) {
        if (list_ids[j] >= n) return fail(HS_ERR_INVALID, "neighbour id out of range");
        p.cols.push_back(list_ids[j]);
      }
      p.max_deg0 = std::max<size_t>(p.max_deg0, e0 - s0);
      p.row_ptr0[i + 1] = (uint32_t)p.cols.size();
      t += (size_t)levels[i] + 1;
    }
    for (size_t i = 0; i < n; i++) {
      if (levels[i] <= 0) continue;
      p.up_base[i] = (uint32_t)p.up_ptr.size();
      for (int l = 1; l <= levels[i]; l++) {
        p.up_ptr.push_back((uint32_t)p.cols.size());
        const uint64_t s0 = list_ptr[first[i] + l], e0 = list_ptr[first[i] + l + 1];
        if (e0 < s0) return fail(HS_ERR_INVALID, "list_ptr not monotone");
        for (uint64_t j = s0; j < e0; j++) {
          if (list_ids[j] >= n) return fail(HS_ERR_INVALID, "neighbour id out of range");
          p.cols.push_back(list_ids[j]);
        }
      }
      p.up_ptr.push_back((uint32_t)p.cols.size());
    }
    if (p.cols.size() >= PackedIndex::NONE) return fail(HS_ERR_INVALID, "adjacency too large for 32-bit CSR offsets");
    p.index_size = 16 * n + 4 * p.cols.size();
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  }
  hs_index *ix = new hs_index();
  ix->device = device;
  hs_status s = upload(ix, p);
  if (s != HS_OK) { delete ix; return s; }
  *out = ix;
  return HS_OK;
}

hs_status hs_index_load(const char *path, int kind, int metric, size_t dim, size_t max_elements, int device,
                        hs_index **out) {
  if (!path || !out) return fail(HS_ERR_INVALID, "null argument");
  return load_from(BinSource(path), kind, metric, dim, max_elements, device, out);
}

hs_status hs_index_load_mem(const void *bytes, size_t len, int kind, int metric, size_t dim, size_t max_elements, int device,
                            hs_index **out) {
  if (!bytes || !out) return fail(HS_ERR_INVALID, "null argument");
  return load_from(BinSource(bytes, len), kind, metric, dim, max_elements, device, out);
}

void hs_index_free(hs_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  delete ix;
}

hs_status hs_set_ef(hs_index *ix, size_t ef) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  if (ef != ix->ef) ix->grow_hash = ix->grow_cand = 0;   // what hs_search_check learned about the scratch shares held for the old ef
  ix->ef = ef;
  return HS_OK;
}
hs_status hs_set_capacity(hs_index *ix, uint32_t cand_cap, uint32_t hash_slots) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  ix->user_cand_cap = cand_cap;
  ix->user_hash_slots = hash_slots;
  return HS_OK;
}
hs_status hs_set_exact_order(hs_index *ix, int on) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  ix->exact_order = on != 0;
  return HS_OK;
}
const char *hs_last_kernel(const hs_index *ix) { return ix ? ix->last_kernel : ""; }
hs_status hs_index_info(const hs_index *ix, hs_info *out) {
  if (!ix || !out) return fail(HS_ERR_INVALID, "null argument");
  *out = ix->info;
  return HS_OK;
}

// One launch group serves at most kMaxLaunchQueries queries: the per-query scratch in global memory (96 KiB each) is
// sized for that many, larger batches run as consecutive groups on the same stream (counters accumulate).
static constexpr size_t kMaxLaunchQueries = 32768;
// from this many queries per launch the fast pass runs as descent / order / level-0 search (see search_dev_group)
static constexpr size_t kOrderMinQueries = 6144;

static hs_status search_dev_group(hs_index *ix, const float *d_q, size_t nq, size_t k, int mode, uint32_t *l32,
                                  uint64_t *l64, float *dd, uint32_t *cnt, uint32_t *stats, Pair *raw, uint32_t *rawsz,
                                  hipStream_t stream, bool first_group, size_t nq_total) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  if (k == 0) return fail(HS_ERR_INVALID, "k must be > 0");
  if (mode != HS_MODE_SLIM_IDS && mode != HS_MODE_PQ) return fail(HS_ERR_INVALID, "bad mode");
  if (ix->info.kind == HS_KIND_SLIMQ) return fail(HS_ERR_INVALID, "SlimQ index: use hs_slimq_search_batch");
  if (mode == HS_MODE_SLIM_IDS && ix->info.kind != HS_KIND_SLIM)
    return fail(HS_ERR_INVALID, "HS_MODE_SLIM_IDS needs a Slim index (searchKnn(q,k,tableint*) exists on HierarchicalNSWSlim only)");
  if (nq > 0x7FFFFFFFu) return fail(HS_ERR_INVALID, "nq too large");
  if (nq == 0) return HS_OK;
  Shape sh;
  hs_status ps = plan_shape(ix, k, sh);
  if (ps != HS_OK) return ps;
  HIP_TRY(hipSetDevice(ix->device));
  hs_index::StreamWs *w = ix->stream_ws(stream);
  HIP_TRY(w->status.ensure(nq));
  HIP_TRY(w->spill.ensure(nq * (size_t)spill_stride_for(sh.ef)));
  // (status needs no clearing: pass 0 takes every query and writes each one's final status)
  // counters[0..12): per-pass overflow / hazard counts.  They are STICKY: they accumulate over the launch groups of a call and
  // over every call issued on this stream until hs_search_check reads and clears them, so a capacity failure in any batch of
  // a pipelined sequence is reported by the check that follows it.  [12]: the group kernel's query queue head (per launch group)
  if (w->counters.n < 48) {
    HIP_TRY(w->counters.ensure(48));
    HIP_TRY(hipMemsetAsync(w->counters.p, 0, 48 * sizeof(uint32_t), stream));
  } else {
    HIP_TRY(hipMemsetAsync(w->counters.p + 12, 0, sizeof(uint32_t), stream));
  }
  if (first_group) w->last_nq += nq_total;   // queries since the last hs_search_check on this stream
  SearchArgs a{};
  a.queries = d_q; a.nq = (uint32_t)nq; a.k = (uint32_t)k; a.ef = sh.ef;
  a.cand_cap = sh.cand_cap; a.hash_slots = sh.hash_slots; a.mode = mode;
  a.mark_ep = (ix->info.kind == HS_KIND_SLIM && mode == HS_MODE_PQ) ? 1 : 0;
  a.out_labels32 = l32; a.out_labels64 = l64; a.out_dists = dd; a.out_counts = cnt; a.stats = stats;
  a.raw_top = raw; a.raw_size = rawsz; a.raw_stride = sh.ef;
  a.status = w->status.p;
  a.spill = w->spill.p; a.spill_slots = kSpillSlots; a.spill_stride = spill_stride_for(sh.ef); a.cand2_cap = kCand2Cap;
  a.log_cap = log_cap_for(sh.ef); a.hop_cap = hop_cap_for(sh.ef);
  static const bool flat_off = getenv("HS_FLAT") && atoi(getenv("HS_FLAT")) == 0;   // diagnostic: heap path from the first expansion
  a.flat = flat_off ? 0u : 1u;
  // A launch that cannot fill the GPU anyway (fewer queries than wavefront slots) lasts as long as its longest query, and
  // LDS is not what limits it: the 16-bit visited set then takes up to 4x the buckets, as far as the queries of this launch
  // still all fit on the chip at once -- the longest queries never see a full bucket.
  if (sh.q_bits && !ix->user_hash_slots) {
    const size_t waves_per_cu = std::max<size_t>((nq + 255) / 256, 1);
    const size_t budget = std::min<size_t>(kLdsPerCU / waves_per_cu, 64 * 1024);
    uint32_t bbits = 0;
    while ((1u << bbits) < sh.q_hash_slots / 4) bbits++;
    for (int step = 0; step < 2; step++) {
      const uint32_t bigger = sh.q_hash_slots * 2;
      const size_t lds_fast = fast_lds_bytes((uint32_t)ix->info.dim, sh.ef, sh.cand_cap_fast, bigger);
      const size_t lds_lean = lean_lds_bytes((uint32_t)ix->info.dim, sh.ef, sh.l_cand_cap, bigger);
      if (std::max(lds_fast, lds_lean) > budget || bbits + 1 > sh.q_bits) break;
      sh.q_hash_slots = bigger;
      bbits++;
    }
  }
  const uint32_t fast_hash = sh.q_bits ? sh.q_hash_slots : sh.hash_slots;
  const bool fast = !ix->exact_order && !raw && fast_supported(ix->dev, sh.ef, (uint32_t)k) &&
                    fast_lds_bytes((uint32_t)ix->info.dim, sh.ef, sh.cand_cap_fast, fast_hash) <= kLdsPerCU;
  // scratch shares of the fast kernel's launches / of everything else
  auto fast_scratch = [&](bool on) {
    a.cand_cap = on ? sh.cand_cap_fast : sh.cand_cap;
    a.hash_slots = on ? fast_hash : sh.hash_slots;
    a.vis_bits = on ? sh.q_bits : 0;
  };
  // The flat kernel (lazy candidate heap, flat_search.hip) answers every bare index it supports.  The older kernels run where it
  // does not (filters, delete marks, threshold_level > 0, dim % 16 != 0, ef > 512: fast / strict) and when asked for by name:
  // HS_KERNEL=lean|fast for A/B runs and their parity tests (HS_LEAN_MIN_EF=<ef> asks for the lean kernel from that ef upwards),
  // HS_KERNEL=flat forces the flat kernel.  No property of the DATA enters the choice (until round 2 a strided sample of the rows
  // -- "integer-valued?" -- moved the lean / fast threshold).
  static const char *kernel_env = getenv("HS_KERNEL");
  static const bool lean_forced = getenv("HS_LEAN_MIN_EF") != nullptr;
  static const uint32_t lean_min_ef = lean_forced ? (uint32_t)atoi(getenv("HS_LEAN_MIN_EF")) : kLeanMinEf;
  static const bool lean_asked = lean_forced || (kernel_env && !strcmp(kernel_env, "lean"));
  const bool lean = lean_asked && fast && sh.ef >= lean_min_ef && lean_supported(ix->dev, sh.ef, (uint32_t)k) &&
                    lean_lds_bytes((uint32_t)ix->info.dim, sh.ef, sh.l_cand_cap, sh.q_bits ? sh.q_hash_slots : sh.l_hash_slots) <= kLdsPerCU;
  static const bool flatk_off = kernel_env && (!strcmp(kernel_env, "lean") || !strcmp(kernel_env, "fast"));
  const FlatPlan fp = plan_flat(ix, sh.ef, nq);
  const bool flatk = !flatk_off && !lean_forced && fast && fp.ok && flatk_supported(ix->dev, sh.ef, (uint32_t)k);
  ix->last_kernel = flatk ? "hs::flat_kernel" : lean ? "hs::lean_kernel" : fast ? "hs::fast_kernel" : "hs::strict_kernel";
  a.queue = w->counters.p + 12;
  a.counters = w->counters.p; a.pass_id = 0;
  static const int order_env = getenv("HS_ORDER") ? atoi(getenv("HS_ORDER")) : -1;   // diagnostic: 0 = never, 1 = always
  const bool ordered = fast && (order_env < 0 ? nq >= kOrderMinQueries : order_env != 0);
  if (flatk) {
    // pass 0: the flat kernel; a query that exhausts its scratch is left ST_OVERFLOW, one whose logs did not fit ST_HAZARD
    a.hash_slots = fp.nb * 4; a.vis_bits = fp.vis_bits; a.fl_nb = fp.nb; a.fl_mul = fp.mul; a.fl_sh = fp.sh;
    a.select_mask = 1u << ST_TODO; a.grid = (uint32_t)nq;
    if (ordered) {   // descent / order / level-0 search, as for the fast kernel below
      HIP_TRY(w->entry.ensure(nq * 4));
      HIP_TRY(w->order.ensure(nq));
      a.entry = reinterpret_cast<uint4 *>(w->entry.p); a.order = w->order.p;
      a.phase = 1;
      HIP_TRY(launch_flatk(ix->dev, a, stream));
      HIP_TRY(launch_order(a.entry, w->order.p, (uint32_t)nq, stream));
      a.phase = 2;
      HIP_TRY(launch_flatk(ix->dev, a, stream));
      a.phase = 0;
    } else {
      HIP_TRY(launch_flatk(ix->dev, a, stream));
    }
    a.cand_cap = sh.cand_cap; a.hash_slots = sh.hash_slots; a.vis_bits = 0;
  } else if (lean) {
    // pass 0: the lean kernel; a query that exhausts even its tier-2 regions is left ST_OVERFLOW for the passes below
    a.cand_cap = sh.l_cand_cap; a.hash_slots = sh.q_bits ? sh.q_hash_slots : sh.l_hash_slots; a.hash_fill_shift = 3; a.vis_bits = sh.q_bits;
    a.select_mask = 1u << ST_TODO; a.grid = (uint32_t)nq;
    if (ordered) {   // descent / order / level-0 search, as for the fast kernel below
      HIP_TRY(w->entry.ensure(nq * 4));
      HIP_TRY(w->order.ensure(nq));
      a.entry = reinterpret_cast<uint4 *>(w->entry.p); a.order = w->order.p;
      a.phase = 1;
      HIP_TRY(launch_lean(ix->dev, a, stream));
      HIP_TRY(launch_order(a.entry, w->order.p, (uint32_t)nq, stream));
      a.phase = 2;
      HIP_TRY(launch_lean(ix->dev, a, stream));
      a.phase = 0;
    } else {
      HIP_TRY(launch_lean(ix->dev, a, stream));
    }
    a.cand_cap = sh.cand_cap; a.hash_slots = sh.hash_slots; a.hash_fill_shift = 0; a.vis_bits = 0;
  } else {
    if (fast) fast_scratch(true);
    // pass 0: every query, one wavefront each
    a.select_mask = 1u << ST_TODO; a.grid = (uint32_t)nq;
    // A launch much larger than what the GPU holds at once (4096 wavefronts of this kernel) ends on the queries that
    // started last; if those are long ones the whole chip waits for them.  The distance of the level-0 entry predicts
    // the number of expansions (rank correlation 0.5 on the bench data), so the descent runs as a launch of its own,
    // the queries are ordered by that distance, farthest first, and the level-0 search takes them in that order.
    if (ordered) {
      HIP_TRY(w->entry.ensure(nq * 4));
      HIP_TRY(w->order.ensure(nq));
      a.entry = reinterpret_cast<uint4 *>(w->entry.p); a.order = w->order.p;
      a.phase = 1;
      HIP_TRY(launch_fast(ix->dev, a, stream));
      if (order_env == 2) a.order = nullptr;   // diagnostic: the split without the ordering
      else HIP_TRY(launch_order(a.entry, w->order.p, (uint32_t)nq, stream));
      a.phase = 2;
      HIP_TRY(launch_fast(ix->dev, a, stream));
      a.phase = 0;
    } else {
      HIP_TRY(fast ? launch_fast(ix->dev, a, stream) : launch_strict(ix->dev, a, stream));
    }
    fast_scratch(false);
  }
  // Re-run pass (normally empty: a launch that scans the statuses and exits): tie queries whose insertion log did not fit
  // (ST_HAZARD) and queries that outgrew their scratch (ST_OVERFLOW) -> strict kernel, a few workgroups, candidate heap and a
  // large tier-2 visited set per workgroup in global memory (kFbGrid).
  {
    a.select_mask = (1u << ST_OVERFLOW) | ((fast || lean || flatk) ? (1u << ST_HAZARD) : 0u); a.grid = (uint32_t)std::min<size_t>(nq, kFbGrid);
    HIP_TRY(w->fb.ensure((size_t)kFbGrid * ((size_t)kFbCand * 2 + kFbSpill)));
    a.cand_cap = sh.fb_cand_cap; a.hash_slots = sh.fb_hash_slots; a.vis_bits = 0; a.hash_fill_shift = 0;
    a.fb_cand = w->fb.p; a.fb_spill = w->fb.p + (size_t)kFbGrid * kFbCand * 2; a.spill_slots = kFbSpill;
    a.counters = w->counters.p + 8; a.pass_id = 2;
    HIP_TRY(launch_strict(ix->dev, a, stream));
  }
  return HS_OK;
}

static hs_status search_dev(hs_index *ix, const float *d_q, size_t nq, size_t k, int mode, uint32_t *l32,
                            uint64_t *l64, float *dd, uint32_t *cnt, uint32_t *stats, Pair *raw, uint32_t *rawsz,
                            hipStream_t stream) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  if (nq > 0x7FFFFFFFu) return fail(HS_ERR_INVALID, "nq too large");
  const size_t dim = ix->info.dim, ef = std::max(ix->ef, k);
  for (size_t off = 0; off < nq || off == 0; off += kMaxLaunchQueries) {
    const size_t m = std::min(kMaxLaunchQueries, nq - off);
    hs_status s = search_dev_group(ix, d_q + off * dim, m, k, mode, l32 ? l32 + off * k : nullptr, l64 ? l64 + off * k : nullptr,
                                   dd ? dd + off * k : nullptr, cnt ? cnt + off : nullptr, stats ? stats + off * 4 : nullptr,
                                   raw ? raw + off * ef : nullptr, rawsz ? rawsz + off : nullptr, stream, off == 0, nq);
    if (s != HS_OK || nq == 0) return s;
  }
  return HS_OK;
}

hs_status hs_search_check(hs_index *ix, void *stream) {
  if (!ix) return fail(HS_ERR_INVALID, "null index");
  uint32_t c[12];
  HIP_TRY(hipSetDevice(ix->device));
  hs_index::StreamWs *w = ix->stream_ws((hipStream_t)stream);
  if (!w->counters.p) return HS_OK;  // nothing was launched on this stream
  HIP_TRY(hipMemcpyAsync(c, w->counters.p, sizeof(c), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(w->counters.p, 0, sizeof(c), (hipStream_t)stream));   // read and cleared: see search_dev_group
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  // learn the scratch sizes from the data: if more than 1% of a batch overflowed in the first passes,
  // later batches start with twice the visited-set slots / candidate capacity.
  const size_t nq = std::max<size_t>(w->last_nq, 1);
  w->last_nq = 0;
  static const bool verbose = getenv("HS_VERBOSE") != nullptr;
  if (verbose) fprintf(stderr, "[hs check] nq %zu: visited overflow %u, candidate overflow %u, tie replays %u, visited-set spills %u | pass 2: %u %u | grow_hash %u grow_cand %u\n",
                       nq, c[0], c[1], c[2], c[3], c[8], c[9], ix->grow_hash, ix->grow_cand);
  if ((size_t)c[3] * 10 > nq && ix->grow_hash < 8 && !ix->user_hash_slots) ix->grow_hash++;
  if ((size_t)(c[1] + c[5]) * 100 > nq && ix->grow_cand < 4 && !ix->user_cand_cap) ix->grow_cand++;
  if (ix->info.kind == HS_KIND_SLIMQ) {
    if (c[8] > 0) return fail(HS_ERR_CAPACITY, std::to_string(c[8]) + " queries expanded more nodes than the 64 KiB on-chip set holds");
    return HS_OK;
  }
  if (c[8] + c[9] > 0)
    return fail(HS_ERR_CAPACITY, std::to_string(c[8] + c[9]) + " queries exhausted even a whole CU's on-chip scratch");
  return HS_OK;
}

// Parity/debug: the flat kernel's visited-set plan for an index of n nodes (dim 128) at (ef, nq): {buckets, multiplier, shift, id bits, ok}
hs_status hs_debug_flat_plan(size_t n, size_t ef, size_t nq, uint32_t *out5) {
  if (!out5) return fail(HS_ERR_INVALID, "null argument");
  hs_index tmp;
  tmp.info.n = n; tmp.info.dim = 128;
  const FlatPlan f = plan_flat(&tmp, (uint32_t)ef, nq);
  out5[0] = f.nb; out5[1] = f.mul; out5[2] = f.sh; out5[3] = f.vis_bits; out5[4] = f.ok ? 1u : 0u;
  return HS_OK;
}

hs_status hs_debug_heap_ops(const uint32_t *ops, size_t n_ops, int wave_pop, uint32_t lds_slots, uint32_t *out_heap, uint32_t *out_pops,
                            uint32_t *out_n) {
  if (!ops || !out_heap || !out_pops || !out_n) return fail(HS_ERR_INVALID, "null argument");
  if (lds_slots < 2 || lds_slots > 8192 || (lds_slots & 1)) return fail(HS_ERR_INVALID, "lds_slots: even, 2..8192");
  DevBuf<uint32_t> d_ops, d_n;
  DevBuf<uint2> d_spill, d_heap, d_pops;
  HIP_TRY(d_ops.alloc(std::max<size_t>(3 * n_ops, 1)));
  HIP_TRY(d_n.alloc(2));
  HIP_TRY(d_spill.alloc(n_ops + 2));
  HIP_TRY(d_heap.alloc(n_ops + 2));
  HIP_TRY(d_pops.alloc(n_ops + 2));
  HIP_TRY(hipMemcpy(d_ops.p, ops, 3 * n_ops * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(flat_heap_ops(d_ops.p, (uint32_t)n_ops, d_spill.p, d_heap.p, d_pops.p, d_n.p, wave_pop, lds_slots, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_n, d_n.p, 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_heap, d_heap.p, (size_t)out_n[0] * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_pops, d_pops.p, (size_t)out_n[1] * 8, hipMemcpyDeviceToHost));
  return HS_OK;
}

hs_status hs_search_batch_dev(hs_index *ix, const float *d_queries, size_t nq, size_t k, int mode,
                              uint32_t *d_out_labels32, uint64_t *d_out_labels64, float *d_out_dists,
                              uint32_t *d_out_counts, uint32_t *d_stats, void *stream) {
  if (mode == HS_MODE_SLIM_IDS && !d_out_labels32) return fail(HS_ERR_INVALID, "out_labels32 required");
  if (mode == HS_MODE_PQ && (!d_out_labels64 || !d_out_dists || !d_out_counts)) return fail(HS_ERR_INVALID, "out_labels64/out_dists/out_counts required");
  return search_dev(ix, d_queries, nq, k, mode, d_out_labels32, d_out_labels64, d_out_dists, d_out_counts, d_stats,
                    nullptr, nullptr, (hipStream_t)stream);
}

// H2D of the queries, the search, D2H of the requested outputs: all enqueued on `stream`, no host synchronisation.  The
// staging buffers belong to (index, stream): a second call on the same stream reuses them in stream order.
// The device's address of a host buffer that is page-locked AND mapped into the device's address space (hipHostMalloc /
// hs_host_alloc, hipHostRegister with the mapped flag), or null: such a buffer needs no staging copy -- the kernels read the
// queries from it and write the results into it directly.
static void *mapped_device_pointer(const void *host) {
  if (!host) return nullptr;
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // pageable memory
  if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
  return at.devicePointer;
}
// A batch from page-locked host buffers, stream-ordered.  A SMALL batch (<= kZeroCopyQueryBytes of queries) is served in place:
// every wavefront stages its query once straight from the mapped host buffer and writes its few dozen result bytes straight
// into the caller's -- a small batch's copy-engine round trips (two per batch, each a cross-engine dependency in the stream)
// cost more than they move: 1250-query batches, 16 in flight, PCIe-inclusive 5.22 -> 5.65 M q/s.  Larger batches go through
// the staging copies (10k-query batches: 13.9 vs 13.6 M q/s in favour of staging).
static constexpr size_t kZeroCopyQueryBytes = 2u << 20;
static hs_status search_async(hs_index *ix, const float *queries, size_t nq, size_t k, int mode, uint32_t *l32, uint64_t *l64,
                              float *dd, uint32_t *cnt, uint32_t *stats, hipStream_t st) {
  if (!ix || !queries) return fail(HS_ERR_INVALID, "null argument");
  if (k == 0) return fail(HS_ERR_INVALID, "k must be > 0");
  if (nq == 0) return HS_OK;
  HIP_TRY(hipSetDevice(ix->device));
  const size_t dim = ix->info.dim;
  hs_index::StreamWs *w = ix->stream_ws(st);
  static const bool zero_copy_off = getenv("HS_ZERO_COPY") && !strcmp(getenv("HS_ZERO_COPY"), "0");      // diagnostic A/B knobs
  static const bool zero_copy_in_only = getenv("HS_ZERO_COPY") && !strcmp(getenv("HS_ZERO_COPY"), "in");   // (outputs staged)
  const bool ids = mode == HS_MODE_SLIM_IDS;
  const float *dq = nullptr;
  if (!zero_copy_off && nq * dim * sizeof(float) <= kZeroCopyQueryBytes) dq = static_cast<const float *>(mapped_device_pointer(queries));
  if (!dq) {
    HIP_TRY(w->aq.ensure(nq * dim));
    HIP_TRY(hipMemcpyAsync(w->aq.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, st));
    dq = w->aq.p;
  }
  // each output: the caller's buffer itself when the device can write it (small batches, as for the queries: with 10k-query
  // batches the staged path measured 2 % faster), else a device buffer + a copy back
  const bool small = nq * dim * sizeof(float) <= kZeroCopyQueryBytes;
  auto direct = [&](void *host) -> void * { return (zero_copy_off || zero_copy_in_only || !small) ? nullptr : mapped_device_pointer(host); };
  uint32_t *o32 = static_cast<uint32_t *>(direct(l32));
  uint64_t *o64 = static_cast<uint64_t *>(direct(l64));
  float *odd = static_cast<float *>(direct(dd));
  uint32_t *ocnt = static_cast<uint32_t *>(direct(cnt));
  uint32_t *ost = static_cast<uint32_t *>(direct(stats));
  const bool c32 = !o32 && (l32 || ids), c64 = !o64 && (l64 || !ids), cdd = !odd && (dd || !ids), ccnt = !ocnt, cst = !ost && stats;
  if (c32) { HIP_TRY(w->al32.ensure(nq * k)); o32 = w->al32.p; }
  if (c64) { HIP_TRY(w->al64.ensure(nq * k)); o64 = w->al64.p; }
  if (cdd) { HIP_TRY(w->adist.ensure(nq * k)); odd = w->adist.p; }
  if (ccnt) { HIP_TRY(w->acnt.ensure(nq)); ocnt = w->acnt.p; }
  if (cst) { HIP_TRY(w->astats.ensure(nq * 4)); ost = w->astats.p; }
  hs_status s = search_dev(ix, dq, nq, k, mode, o32, o64, odd, ocnt, ost, nullptr, nullptr, st);
  if (s != HS_OK) return s;
  if (c32 && l32) HIP_TRY(hipMemcpyAsync(l32, w->al32.p, nq * k * 4, hipMemcpyDeviceToHost, st));
  if (c64 && l64) HIP_TRY(hipMemcpyAsync(l64, w->al64.p, nq * k * 8, hipMemcpyDeviceToHost, st));
  if (cdd && dd) HIP_TRY(hipMemcpyAsync(dd, w->adist.p, nq * k * 4, hipMemcpyDeviceToHost, st));
  if (ccnt && cnt) HIP_TRY(hipMemcpyAsync(cnt, w->acnt.p, nq * 4, hipMemcpyDeviceToHost, st));
  if (cst && stats) HIP_TRY(hipMemcpyAsync(stats, w->astats.p, nq * 16, hipMemcpyDeviceToHost, st));
  return HS_OK;
}

static hs_status search_host(hs_index *ix, const float *queries, size_t nq, size_t k, int mode, uint32_t *l32,
                             uint64_t *l64, float *dd, uint32_t *cnt, uint32_t *stats, float *raw_d, uint32_t *raw_i,
                             uint32_t *raw_sz) {
  if (!ix || !queries) return fail(HS_ERR_INVALID, "null argument");
  if (nq == 0) return HS_OK;
  const bool want_raw = raw_d || raw_i || raw_sz;
  if (!want_raw) {   // one stream-ordered sequence of copies and launches, one synchronisation
    hs_status s = search_async(ix, queries, nq, k, mode, l32, l64, dd, cnt, stats, nullptr);
    if (s != HS_OK) return s;
    return hs_search_check(ix, nullptr);
  }
  HIP_TRY(hipSetDevice(ix->device));
  const size_t dim = ix->info.dim;
  const size_t ef = std::max(ix->ef, k);
  HIP_TRY(ix->wq.ensure(nq * dim));
  HIP_TRY(ix->wstats.ensure(nq * 4));
  HIP_TRY(ix->wraw.ensure(nq * ef));
  HIP_TRY(ix->wrawsz.ensure(nq));
  hipStream_t st = nullptr;
  HIP_TRY(hipMemcpyAsync(ix->wq.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(ix->wl32.ensure(nq * k));
  HIP_TRY(ix->wl64.ensure(nq * k));
  HIP_TRY(ix->wdist.ensure(nq * k));
  HIP_TRY(ix->wcnt.ensure(nq));
  hs_status s = search_dev(ix, ix->wq.p, nq, k, mode, ix->wl32.p, ix->wl64.p, ix->wdist.p, ix->wcnt.p, ix->wstats.p,
                           ix->wraw.p, ix->wrawsz.p, st);
  if (s != HS_OK) return s;
  s = hs_search_check(ix, st);
  if (s != HS_OK) return s;
  if (stats) HIP_TRY(hipMemcpy(stats, ix->wstats.p, nq * 16, hipMemcpyDeviceToHost));
  std::vector<Pair> raw(nq * ef);
  std::vector<uint32_t> sz(nq);
  HIP_TRY(hipMemcpy(raw.data(), ix->wraw.p, nq * ef * sizeof(Pair), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sz.data(), ix->wrawsz.p, nq * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nq; i++) {
    if (raw_sz) raw_sz[i] = sz[i];
    for (size_t j = 0; j < ef; j++) {
      const bool v = j < sz[i];
      if (raw_d) raw_d[i * ef + j] = v ? raw[i * ef + j].d : 0.f;
      if (raw_i) raw_i[i * ef + j] = v ? raw[i * ef + j].id : 0u;
    }
  }
  return HS_OK;
}

hs_status hs_search_batch_async(hs_index *ix, const float *queries, size_t nq, size_t k, int mode, uint32_t *out_labels32,
                                uint64_t *out_labels64, float *out_dists, uint32_t *out_counts, uint32_t *stats, void *stream) {
  if (mode == HS_MODE_SLIM_IDS && !out_labels32) return fail(HS_ERR_INVALID, "out_labels32 required");
  if (mode == HS_MODE_PQ && (!out_labels64 || !out_dists || !out_counts)) return fail(HS_ERR_INVALID, "out_labels64/out_dists/out_counts required");
  if (mode != HS_MODE_SLIM_IDS && mode != HS_MODE_PQ) return fail(HS_ERR_INVALID, "bad mode");
  return search_async(ix, queries, nq, k, mode, out_labels32, out_labels64, out_dists, out_counts, stats, (hipStream_t)stream);
}
void *hs_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void hs_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}
void *hs_host_device_pointer(const void *host) { return mapped_device_pointer(host); }

hs_status hs_search_batch(hs_index *ix, const float *queries, size_t nq, size_t k, int mode, uint32_t *out_labels32,
                          uint64_t *out_labels64, float *out_dists, uint32_t *out_counts, uint32_t *stats) {
  if (mode == HS_MODE_SLIM_IDS && !out_labels32) return fail(HS_ERR_INVALID, "out_labels32 required");
  if (mode == HS_MODE_PQ && (!out_labels64 || !out_dists || !out_counts)) return fail(HS_ERR_INVALID, "out_labels64/out_dists/out_counts required");
  return search_host(ix, queries, nq, k, mode, out_labels32, out_labels64, out_dists, out_counts, stats, nullptr,
                     nullptr, nullptr);
}

// ---- HNSW-SlimQ ------------------------------------------------------------------------------------------------
hs_status hs_slimq_set_dataset(hs_index *ix, const float *base, size_t n, size_t dim) {
  if (!ix || !base) return fail(HS_ERR_INVALID, "null argument");
  if (ix->info.kind != HS_KIND_SLIMQ) return fail(HS_ERR_INVALID, "not a SlimQ index");
  if (n != ix->info.n || dim != ix->info.dim) return fail(HS_ERR_INVALID, "dataset shape does not match the index");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(ix->vec.alloc(std::max<size_t>(n * dim, 1)));
  HIP_TRY(hipMemcpy(ix->vec.p, base, n * dim * sizeof(float), hipMemcpyHostToDevice));
  ix->dev.vec = ix->vec.p;
  ix->sq.raw = ix->vec.p;
  ix->has_dataset = true;
  ix->info.device_bytes += n * dim * 4;
  return HS_OK;
}
hs_status hs_slimq_set_tconst(hs_index *ix, double t_const) {
  if (!ix || !(t_const > 0)) return fail(HS_ERR_INVALID, "bad argument");
  if (ix->info.kind != HS_KIND_SLIMQ) return fail(HS_ERR_INVALID, "not a SlimQ index");
  ix->sq.t_const = t_const;
  return HS_OK;
}
double hs_slimq_get_tconst(const hs_index *ix) { return ix && ix->info.kind == HS_KIND_SLIMQ ? ix->sq.t_const : 0.0; }

static constexpr uint32_t kSlimQMaxHash = 16384;  // largest expanded-node set in LDS (64 KiB)
static constexpr uint32_t kSlimQFbHash = 65536, kSlimQFbGrid = 64;  // second pass: the set in global memory, 256 KiB per workgroup

hs_status hs_slimq_search_batch_dev(hs_index *ix, const float *d_queries, size_t nq, size_t k, uint64_t *d_out_labels,
                                    float *d_out_dists, uint32_t *d_out_counts, uint32_t *d_stats, void *stream_) {
  if (!ix || !d_queries || !d_out_labels || !d_out_dists || !d_out_counts) return fail(HS_ERR_INVALID, "null argument");
  if (ix->info.kind != HS_KIND_SLIMQ) return fail(HS_ERR_INVALID, "not a SlimQ index");
  if (!ix->has_dataset) return fail(HS_ERR_INVALID, "hs_slimq_set_dataset() first (setDataset, hnswalg_slimq.h:303)");
  if (k == 0 || k > 1024) return fail(HS_ERR_INVALID, "k must be in 1..1024");
  if (!slimq_supported((uint32_t)ix->ef)) return fail(HS_ERR_UNSUPPORTED, "SlimQ supports 1 <= ef <= 1024");
  if (nq > 0x7FFFFFFFu) return fail(HS_ERR_INVALID, "nq too large");
  if (nq == 0) return HS_OK;
  hipStream_t stream = (hipStream_t)stream_;
  HIP_TRY(hipSetDevice(ix->device));
  hs_index::StreamWs *w = ix->stream_ws(stream);
  HIP_TRY(w->status.ensure(nq));
  if (w->counters.n < 48) {   // sticky until hs_search_check clears them (see search_dev_group); status: written for every query by the first pass
    HIP_TRY(w->counters.ensure(48));
    HIP_TRY(hipMemsetAsync(w->counters.p, 0, 48 * sizeof(uint32_t), stream));
  }
  w->last_nq += nq;
  SlimQArgs a{};
  a.queries = d_queries; a.nq = (uint32_t)nq; a.k = (uint32_t)k; a.pool_cap = (uint32_t)ix->ef;
  // expansions per query stay below ~ef on real graphs; 75 % fill of 4 ef slots leaves 3x headroom, and a query
  // that still outgrows it is redone with the 64 KiB set
  a.hash_slots = ix->user_hash_slots ? next_pow2(ix->user_hash_slots) : next_pow2(std::max<uint32_t>(256, 4 * (uint32_t)ix->ef));
  a.hash_slots = std::min(a.hash_slots, kSlimQMaxHash);
  a.out_labels = d_out_labels; a.out_dists = d_out_dists; a.out_counts = d_out_counts; a.stats = d_stats;
  a.status = w->status.p;
  a.trace = ix->trace_ptr; a.trace_cap = ix->trace_cap;
  const uint32_t pw = slimq_prep_words(ix->sq.ncl, ix->sq.padded);
  HIP_TRY(w->prep.ensure(nq * (size_t)pw));
  ix->last_kernel = "hs::slimq_kernel";
  HIP_TRY(launch_slimq_prep(ix->sq, (uint32_t)ix->info.dim, ix->info.metric, d_queries, (uint32_t)nq, w->prep.p, nullptr, stream));
  a.prep = w->prep.p;
  a.select_mask = 1u << ST_TODO; a.grid = (uint32_t)nq; a.counters = w->counters.p;
  static const int order_env = getenv("HS_ORDER") ? atoi(getenv("HS_ORDER")) : -1;   // diagnostic: 0 = never, 1 = always
  if (!a.trace && (order_env < 0 ? nq >= kOrderMinQueries : order_env != 0)) {
    // descent / order by the entry's estimated distance, farthest first / level-0 search (see search_dev_group)
    HIP_TRY(w->entry.ensure(nq * 4));
    HIP_TRY(w->order.ensure(nq));
    a.entry = reinterpret_cast<uint4 *>(w->entry.p); a.order = w->order.p;
    a.phase = 1;
    HIP_TRY(launch_slimq(ix->dev, ix->sq, a, stream));
    HIP_TRY(launch_order(a.entry, w->order.p, (uint32_t)nq, stream));
    a.phase = 2;
    HIP_TRY(launch_slimq(ix->dev, ix->sq, a, stream));
    a.phase = 0;
  } else {
    HIP_TRY(launch_slimq(ix->dev, ix->sq, a, stream));
  }
  {
    HIP_TRY(w->fb.ensure((size_t)kSlimQFbGrid * kSlimQFbHash));
    a.select_mask = 1u << ST_OVERFLOW; a.grid = (uint32_t)std::min<size_t>(nq, kSlimQFbGrid); a.hash_slots = kSlimQFbHash; a.fb_tab = w->fb.p;
    a.counters = w->counters.p + 8;
    HIP_TRY(launch_slimq(ix->dev, ix->sq, a, stream));
  }
  return HS_OK;
}

// Parity/debug entry: the query preparation as the kernel computed it (rotation, split query, centroid table).
hs_status hs_slimq_prepare_debug(hs_index *ix, const float *queries, size_t nq, float *out) {
  if (!ix || !queries || !out) return fail(HS_ERR_INVALID, "null argument");
  if (ix->info.kind != HS_KIND_SLIMQ) return fail(HS_ERR_INVALID, "not a SlimQ index");
  if (nq == 0) return HS_OK;
  HIP_TRY(hipSetDevice(ix->device));
  const size_t P = ix->sq.padded, ncl = ix->sq.ncl, npl = P / 8, row = P + 3 + ncl + npl;
  const uint32_t pw = slimq_prep_words(ix->sq.ncl, ix->sq.padded);
  DevBuf<float> dq, dy;
  DevBuf<uint32_t> dp;
  HIP_TRY(dq.alloc(nq * ix->info.dim)); HIP_TRY(dy.alloc(nq * P)); HIP_TRY(dp.alloc(nq * (size_t)pw));
  HIP_TRY(hipMemcpy(dq.p, queries, nq * ix->info.dim * 4, hipMemcpyHostToDevice));
  HIP_TRY(launch_slimq_prep(ix->sq, (uint32_t)ix->info.dim, ix->info.metric, dq.p, (uint32_t)nq, dp.p, dy.p, nullptr));
  std::vector<float> y(nq * P);
  std::vector<uint32_t> pr(nq * (size_t)pw);
  HIP_TRY(hipMemcpy(y.data(), dy.p, y.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(pr.data(), dp.p, pr.size() * 4, hipMemcpyDeviceToHost));
  const size_t poff = (4 + ncl + 1) & ~size_t(1);
  for (size_t i = 0; i < nq; i++) {
    float *o = out + i * row;
    const uint32_t *r = &pr[i * pw];
    memcpy(o, &y[i * P], P * 4);
    memcpy(o + P, r, 12);
    memcpy(o + P + 3, r + 4, ncl * 4);
    memcpy(o + P + 3 + ncl, r + poff, npl * 4);
  }
  return HS_OK;
}

// Parity/debug entry: the sequence of SearchBuffer pops of each query (node id, bit 31 = already expanded).
hs_status hs_slimq_trace(hs_index *ix, const float *queries, size_t nq, size_t k, uint32_t *out_trace, size_t trace_cap,
                         uint32_t *stats) {
  if (!ix || !queries || !out_trace || trace_cap == 0) return fail(HS_ERR_INVALID, "null argument");
  if (nq == 0) return HS_OK;
  HIP_TRY(hipSetDevice(ix->device));
  DevBuf<float> dq, dd;
  DevBuf<uint64_t> dl;
  DevBuf<uint32_t> dc, ds, dt;
  HIP_TRY(dq.alloc(nq * ix->info.dim)); HIP_TRY(dl.alloc(nq * k)); HIP_TRY(dd.alloc(nq * k)); HIP_TRY(dc.alloc(nq));
  HIP_TRY(ds.alloc(nq * 4)); HIP_TRY(dt.alloc(nq * trace_cap));
  HIP_TRY(hipMemcpy(dq.p, queries, nq * ix->info.dim * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dt.p, 0xFF, nq * trace_cap * 4));
  ix->trace_ptr = dt.p; ix->trace_cap = (uint32_t)trace_cap;
  hs_status s = hs_slimq_search_batch_dev(ix, dq.p, nq, k, dl.p, dd.p, dc.p, ds.p, nullptr);
  ix->trace_ptr = nullptr; ix->trace_cap = 0;
  if (s != HS_OK) return s;
  s = hs_search_check(ix, nullptr);
  if (s != HS_OK) return s;
  HIP_TRY(hipMemcpy(out_trace, dt.p, nq * trace_cap * 4, hipMemcpyDeviceToHost));
  if (stats) HIP_TRY(hipMemcpy(stats, ds.p, nq * 16, hipMemcpyDeviceToHost));
  return HS_OK;
}

hs_status hs_slimq_search_batch(hs_index *ix, const float *queries, size_t nq, size_t k, uint64_t *out_labels,
                                float *out_dists, uint32_t *out_counts, uint32_t *stats) {
  if (!ix || !queries || !out_labels) return fail(HS_ERR_INVALID, "null argument");
  if (nq == 0) return HS_OK;
  HIP_TRY(hipSetDevice(ix->device));
  const size_t dim = ix->info.dim;
  HIP_TRY(ix->wq.ensure(nq * dim));
  HIP_TRY(ix->wl64.ensure(nq * k));
  HIP_TRY(ix->wdist.ensure(nq * k));
  HIP_TRY(ix->wcnt.ensure(nq));
  HIP_TRY(ix->wstats.ensure(nq * 4));
  hipStream_t st = nullptr;
  HIP_TRY(hipMemcpyAsync(ix->wq.p, queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, st));
  hs_status s = hs_slimq_search_batch_dev(ix, ix->wq.p, nq, k, ix->wl64.p, ix->wdist.p, ix->wcnt.p, ix->wstats.p, st);
  if (s != HS_OK) return s;
  s = hs_search_check(ix, st);
  if (s != HS_OK) return s;
  HIP_TRY(hipMemcpy(out_labels, ix->wl64.p, nq * k * 8, hipMemcpyDeviceToHost));
  if (out_dists) HIP_TRY(hipMemcpy(out_dists, ix->wdist.p, nq * k * 4, hipMemcpyDeviceToHost));
  if (out_counts) HIP_TRY(hipMemcpy(out_counts, ix->wcnt.p, nq * 4, hipMemcpyDeviceToHost));
  if (stats) HIP_TRY(hipMemcpy(stats, ix->wstats.p, nq * 16, hipMemcpyDeviceToHost));
  return HS_OK;
}

hs_status hs_labels(const hs_index *ix, uint64_t *out_labels) {
  if (!ix || !out_labels) return fail(HS_ERR_INVALID, "null argument");
  std::copy(ix->host_labels.begin(), ix->host_labels.end(), out_labels);
  return HS_OK;
}

// The reference tests "!isMarkedDeleted(id) && (*isIdAllowed)(label)" together wherever a filter is consulted
// (hnswalg.h:348-349, 442-444; hnswalg_slim.h:578-580), and a filter forces the !bare_bone branch
// (hnswalg.h:1421, hnswalg_slim.h:1884).  So a filtered search is the ordinary search over an index whose
// delete-mark array is (deleted | !allowed) and whose has_deleted flag is set.
hs_status hs_search_batch_filtered(hs_index *ix, const float *queries, size_t nq, size_t k, const uint8_t *allowed,
                                   uint64_t *out_labels64, float *out_dists, uint32_t *out_counts, uint32_t *stats) {
  if (!ix || !allowed) return fail(HS_ERR_INVALID, "null argument");
  if (!out_labels64 || !out_dists || !out_counts) return fail(HS_ERR_INVALID, "out_labels64/out_dists/out_counts required");
  if (ix->info.kind == HS_KIND_SLIM && ix->info.threshold_level != 0)
    return fail(HS_ERR_UNSUPPORTED, "filtered search on a Slim index with threshold_level > 0 is not supported");
  HIP_TRY(hipSetDevice(ix->device));
  const size_t n = ix->info.n;
  std::vector<uint8_t> excl(std::max<size_t>(n, 1));
  for (size_t i = 0; i < n; i++) excl[i] = (ix->host_deleted[i] || !allowed[i]) ? 1 : 0;
  HIP_TRY(ix->wexcl.ensure(excl.size()));
  HIP_TRY(hipMemcpy(ix->wexcl.p, excl.data(), excl.size(), hipMemcpyHostToDevice));
  const DevIndex saved = ix->dev;
  ix->dev.deleted = ix->wexcl.p;
  ix->dev.has_deleted = 1;
  hs_status s = search_host(ix, queries, nq, k, HS_MODE_PQ, nullptr, out_labels64, out_dists, out_counts, stats, nullptr,
                            nullptr, nullptr);
  ix->dev = saved;
  return s;
}

hs_status hs_search_batch_raw(hs_index *ix, const float *queries, size_t nq, size_t k, int mode, float *raw_dists,
                              uint32_t *raw_ids, uint32_t *raw_sizes, uint32_t *stats) {
  if (!raw_dists || !raw_ids || !raw_sizes) return fail(HS_ERR_INVALID, "raw outputs required");
  return search_host(ix, queries, nq, k, mode, nullptr, nullptr, nullptr, nullptr, stats, raw_dists, raw_ids, raw_sizes);
}

hs_status hs_build_hnsw(const float *base, size_t n, size_t dim, int metric, size_t M, size_t ef_construction,
                        const char *branching_factor, size_t seed, int threads, const char *out_path) {
  return hs_build_hnsw_labeled(base, nullptr, n, dim, metric, M, ef_construction, branching_factor, seed, threads, out_path);
}

hs_status hs_build_hnsw_labeled(const float *base, const uint64_t *labels, size_t n, size_t dim, int metric, size_t M,
                                size_t ef_construction, const char *branching_factor, size_t seed, int threads,
                                const char *out_path) {
  if (!base || !out_path || !branching_factor || n == 0) return fail(HS_ERR_INVALID, "bad argument");
  try {
    VanillaGraph g;
    g.build(base, n, dim, (Metric)metric, M, ef_construction, branching_factor, seed, threads, labels);
    g.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}

hs_status hs_convert_slim(const char *hnsw_path, int metric, size_t dim, int threshold_level, float top_degree_percent0,
                          float top_degree_percent, size_t top_degree_M0, size_t low_degree_m0, size_t top_degree_M,
                          size_t low_degree_m, int threads, const char *out_path) {
  if (!hnsw_path || !out_path) return fail(HS_ERR_INVALID, "bad argument");
  try {
    VanillaGraph g;
    g.load(hnsw_path, (Metric)metric, dim);
    SlimParams p;
    p.threshold_level = threshold_level;
    p.top_pct0 = top_degree_percent0; p.top_pct = top_degree_percent;
    p.top_M0 = top_degree_M0; p.low_m0 = low_degree_m0; p.top_M = top_degree_M; p.low_m = low_degree_m;
    SlimGraph s;
    s.convert(g, p, threads);
    s.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}

// The base graph of HNSW-SlimQ: rabitqlib::hnsw::HierarchicalNSW::construct's edges (RqGraph in host_graph.hpp), stored in the
// vanilla file layout so that the converters read it like any hnswlib index.
hs_status hs_build_rabitq_hnsw(const float *base, size_t n, size_t dim, int metric, size_t M, size_t ef_construction, size_t seed,
                               int threads, const char *out_path) {
  if (!base || !out_path || n == 0 || dim == 0) return fail(HS_ERR_INVALID, "bad argument");
  if (metric != HS_METRIC_L2 && metric != HS_METRIC_IP) return fail(HS_ERR_INVALID, "bad metric");
  if (M < 2) return fail(HS_ERR_INVALID, "M must be >= 2");   // mult = 1 / ln M
  try {
    RqGraph g;
    g.rq_build(base, n, dim, (Metric)metric, M, ef_construction, seed, threads);
    g.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}

// HierarchicalNSWSlimQ::convertFromHNSW's graph passes (hnswalg_slimq.h:1546-1762): Slim's passes with SlimQ's own
// PruneByHeuristic (:1334-1362) and rabitqlib's raw distance.  Output: a Slim-layout file for hs_convert_slimq.
hs_status hs_convert_slimq_graph(const char *hnsw_path, int metric, size_t dim, int threshold_level, float top_degree_percent0,
                                 float top_degree_percent, size_t top_degree_M0, size_t low_degree_m0, size_t top_degree_M,
                                 size_t low_degree_m, int threads, const char *out_path) {
  if (!hnsw_path || !out_path) return fail(HS_ERR_INVALID, "bad argument");
  try {
    VanillaGraph g;
    g.load(hnsw_path, (Metric)metric, dim);
    SlimParams p;
    p.threshold_level = threshold_level;
    p.top_pct0 = top_degree_percent0; p.top_pct = top_degree_percent;
    p.top_M0 = top_degree_M0; p.low_m0 = low_degree_m0; p.top_M = top_degree_M; p.low_m = low_degree_m;
    SlimGraph s;
    s.convert(g, p, threads, true);
    s.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}

// convertFromHNSW with the list-level work on the GPU (convert_gpu.hip); identical output bytes.  Shapes outside the device path
// (degree capacities above 32, a reverse-edge list that outgrows the on-chip buffers) run the CPU conversion instead.
hs_status hs_convert_slim_gpu(const char *hnsw_path, int metric, size_t dim, int threshold_level, float top_degree_percent0,
                              float top_degree_percent, size_t top_degree_M0, size_t low_degree_m0, size_t top_degree_M,
                              size_t low_degree_m, int device, int threads, const char *out_path, int *used_gpu, double *kernel_ms) {
  if (!hnsw_path || !out_path) return fail(HS_ERR_INVALID, "bad argument");
  if (hs_device_count() <= device) return fail(HS_ERR_DEVICE, "no HIP device");
  try {
    VanillaGraph g;
    g.load(hnsw_path, (Metric)metric, dim);
    SlimParams p;
    p.threshold_level = threshold_level;
    p.top_pct0 = top_degree_percent0; p.top_pct = top_degree_percent;
    p.top_M0 = top_degree_M0; p.low_m0 = low_degree_m0; p.top_M = top_degree_M; p.low_m = low_degree_m;
    SlimGraph s;
    std::string err;
    double ms = 0.0;
    const bool ok = s.convert_gpu(g, p, device, threads, &ms, &err);
    if (!ok) {
      if (!err.empty()) return fail(HS_ERR_DEVICE, err);
      s.convert(g, p, threads);
    }
    if (used_gpu) *used_gpu = ok ? 1 : 0;
    if (kernel_ms) *kernel_ms = ok ? ms : 0.0;
    s.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}

// ---- exhaustive k-NN: hnswlib::BruteforceSearch::searchKnn (bruteforce.h:106-135) for a batch ----------------------
hs_status hs_brute_force_dev(const float *d_base, const uint64_t *d_labels, size_t n, size_t dim, int metric,
                             const float *d_queries, size_t nq, size_t k, uint64_t *d_out_labels, float *d_out_dists,
                             uint32_t *d_out_counts, void *stream_) {
  if (!d_base || !d_queries || !d_out_labels || !d_out_dists) return fail(HS_ERR_INVALID, "null argument");
  if (metric != HS_METRIC_L2 && metric != HS_METRIC_IP) return fail(HS_ERR_INVALID, "bad metric");
  if (dim == 0) return fail(HS_ERR_INVALID, "dim must be > 0");
  if (dim > 4096) return fail(HS_ERR_UNSUPPORTED, "brute force supports dim <= 4096");
  if (k == 0 || k > 64) return fail(HS_ERR_UNSUPPORTED, "brute force supports 1 <= k <= 64");
  if (n > 0xFFFFFFF0u || nq > 0x7FFFFFFFu) return fail(HS_ERR_INVALID, "too many rows / queries");
  if (nq == 0) return HS_OK;
  hipStream_t stream = (hipStream_t)stream_;
  hipPointerAttribute_t attr;   // run (and allocate the workspace) on the device that holds the rows
  if (hipPointerGetAttributes(&attr, d_base) == hipSuccess) HIP_TRY(hipSetDevice(attr.device));
  uint32_t gx = 0, rpb = 0;
  const size_t bytes = bf_partial_bytes((uint32_t)n, (uint32_t)nq, (uint32_t)k, &gx, &rpb);
  void *partial = nullptr;
  HIP_TRY(hipMalloc(&partial, std::max<size_t>(bytes, 16)));
  hipError_t e = launch_brute_force(d_base, d_labels, (uint32_t)n, (uint32_t)dim, metric, d_queries, (uint32_t)nq, (uint32_t)k, partial, gx,
                                    rpb, d_out_labels, d_out_dists, d_out_counts, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(partial);
  if (e != hipSuccess) return fail(HS_ERR_DEVICE, std::string("brute force: ") + hipGetErrorString(e));
  return HS_OK;
}

hs_status hs_brute_force(const float *base, size_t n, size_t dim, int metric, const uint64_t *labels, const float *queries,
                         size_t nq, size_t k, int device, uint64_t *out_labels, float *out_dists, uint32_t *out_counts) {
  if (!base || !queries || !out_labels || !out_dists) return fail(HS_ERR_INVALID, "null argument");
  if (hs_device_count() <= device) return fail(HS_ERR_DEVICE, "no HIP device (this library has no CPU search path)");
  if (nq == 0) return HS_OK;
  HIP_TRY(hipSetDevice(device));
  DevBuf<float> db, dq, dd;
  DevBuf<uint64_t> dl, dol;
  DevBuf<uint32_t> dc;
  HIP_TRY(db.alloc(std::max<size_t>(n * dim, 1))); HIP_TRY(dq.alloc(nq * dim)); HIP_TRY(dd.alloc(nq * k)); HIP_TRY(dol.alloc(nq * k));
  HIP_TRY(dc.alloc(nq));
  if (labels) { HIP_TRY(dl.alloc(std::max<size_t>(n, 1))); HIP_TRY(hipMemcpy(dl.p, labels, n * 8, hipMemcpyHostToDevice)); }
  HIP_TRY(hipMemcpy(db.p, base, n * dim * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dq.p, queries, nq * dim * 4, hipMemcpyHostToDevice));
  hs_status s = hs_brute_force_dev(db.p, labels ? dl.p : nullptr, n, dim, metric, dq.p, nq, k, dol.p, dd.p, dc.p, nullptr);
  if (s != HS_OK) return s;
  HIP_TRY(hipMemcpy(out_labels, dol.p, nq * k * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_dists, dd.p, nq * k * 4, hipMemcpyDeviceToHost));
  if (out_counts) HIP_TRY(hipMemcpy(out_counts, dc.p, nq * 4, hipMemcpyDeviceToHost));
  return HS_OK;
}

hs_status hs_convert_slimq(const char *slim_path, int metric, size_t dim, const float *centroids, size_t num_cluster,
                           const uint32_t *cluster_ids, uint64_t flip_seed, int threads, const char *out_path) {
  if (!slim_path || !out_path || !centroids || num_cluster == 0) return fail(HS_ERR_INVALID, "bad argument");
  if (dim < 64 || dim >= 4096) return fail(HS_ERR_UNSUPPORTED, "SlimQ supports 64 <= dim < 4096");
  try {
    SlimGraph s;
    s.load(slim_path, (Metric)metric, dim);
    if (cluster_ids)
      for (size_t i = 0; i < s.count; i++)
        if (cluster_ids[i] >= num_cluster) return fail(HS_ERR_INVALID, "cluster id out of range");
    SlimQGraph q;
    q.from_slim(s, metric, centroids, num_cluster, cluster_ids, flip_seed, threads);
    q.save(out_path);
  } catch (std::bad_alloc &) {
    return fail(HS_ERR_NOMEM, "Not enough memory");
  } catch (std::exception &e) {
    return from_exception(e);
  }
  return HS_OK;
}
double hs_rabitq_default_tconst(size_t padded_dim, uint64_t seed) {
  if (padded_dim == 0 || padded_dim % 64) return 0.0;
  return rq_default_tconst(padded_dim, seed);
}

hs_status hs_rabitq_rotate(size_t dim, const uint8_t *flips, const float *in, size_t n, float *out) {
  if (!flips || !in || !out || dim == 0) return fail(HS_ERR_INVALID, "bad argument");
  Rotator r;
  r.init(dim);
  if (r.trunc < 64 || r.trunc > 2048) return fail(HS_ERR_UNSUPPORTED, "rotator supports 64 <= dim < 4096");
  std::copy(flips, flips + r.flip.size(), r.flip.begin());
  for (size_t i = 0; i < n; i++) r.rotate(in + i * dim, out + i * r.padded);
  return HS_OK;
}
hs_status hs_rabitq_quantize_data(size_t padded, int metric, const float *rotated, size_t n, const float *centroid,
                                  uint64_t *codes, float *factors) {
  if (!rotated || !centroid || !codes || !factors || padded % 64) return fail(HS_ERR_INVALID, "bad argument");
  for (size_t i = 0; i < n; i++) rq_quantize_data(rotated + i * padded, centroid, padded, metric, codes + i * padded / 64, factors + i * 3);
  return HS_OK;
}
hs_status hs_rabitq_prepare_query(size_t padded, double t_const, const float *rotated_q, size_t n, float *out3,
                                  uint64_t *bins) {
  if (!rotated_q || !out3 || !bins || padded % 64) return fail(HS_ERR_INVALID, "bad argument");
  RqQuery q;
  for (size_t i = 0; i < n; i++) {
    rq_prepare_query(rotated_q + i * padded, padded, t_const, q);
    out3[i * 3] = q.delta; out3[i * 3 + 1] = q.vl; out3[i * 3 + 2] = q.k1xsumq;
    std::copy(q.bins.begin(), q.bins.end(), bins + i * padded / 64 * 4);
  }
  return HS_OK;
}
hs_status hs_rabitq_estimate(size_t padded, const uint64_t *codes, const float *factors, size_t nd, const float *q3,
                             const uint64_t *bins, const float *g_add, const float *g_error, size_t nq, float *out) {
  if (!codes || !factors || !q3 || !bins || !g_add || !g_error || !out) return fail(HS_ERR_INVALID, "bad argument");
  const uint32_t nblk = (uint32_t)(padded / 64);
  for (size_t i = 0; i < nq; i++)
    for (size_t j = 0; j < nd; j++) {
      const float ip = rq_ip_x0_qr(codes + j * nblk, bins + i * nblk * 4, nblk, q3[i * 3], q3[i * 3 + 1]);
      const float est = rq_est_dist(factors[j * 3], g_add[i], factors[j * 3 + 1], ip, q3[i * 3 + 2]);
      float *o = out + (i * nd + j) * 3;
      o[0] = ip; o[1] = est; o[2] = est - factors[j * 3 + 2] * g_error[i];
    }
  return HS_OK;
}


// ---- multi-GPU: replicated index, contiguous query shards, one all-gather of the packed results --------
// One process drives n devices; every call below is asynchronous on a per-device stream, so a single host thread
// enqueues the whole step (RCCL's single-process group API: ncclCommInitAll + ncclGroupStart/End).  RCCL is opened with
// dlopen (RTLD_LOCAL): no link-time dependency, and a host process that already carries another RCCL build (PyTorch
// bundles one) keeps its own symbols.  A communicator has kCommSlots independent SLOTS (streams, events, gather buffers per
// device): hs_search_batch_sharded_async(.., slot) only enqueues, hs_comm_check(.., slot) waits for that slot's batch, so a
// caller keeps several batches in flight per device -- a split batch is a small launch per device and lasts as long as its
// longest query, the chip only fills up with several of them.
static constexpr int kCommSlots = 8;
struct hs_comm {
  int n = 0;
  std::vector<int> dev;
  bool loopback = false;   // the same device listed more than once (1-GPU rehearsal): the gather runs as device copies
  void *lib = nullptr;
  std::vector<void *> comms;
  struct Dev {
    DevBuf<float> q, all_dist;
    DevBuf<uint32_t> all_l32, all_cnt;
    DevBuf<uint64_t> all_l64;
  };
  struct Slot {
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    std::vector<std::unique_ptr<Dev>> d;
    std::vector<size_t> rows;   // queries device r searched in the batch in flight (0: nothing to check)
    bool busy = false;
  };
  std::vector<Slot> slots;
  int last_slot = 0;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
enum { kNcclUint8 = 1, kNcclUint32 = 3, kNcclUint64 = 5, kNcclFloat32 = 7 };   // ncclDataType_t (rccl.h)
#if __has_include(<rccl/rccl.h>)
}  // extern "C"
#include <rccl/rccl.h>   // declarations only (nothing here is called through them): the constants above against the header of this ROCm
static_assert((int)ncclUint8 == kNcclUint8 && (int)ncclUint32 == kNcclUint32 && (int)ncclUint64 == kNcclUint64 && (int)ncclFloat32 == kNcclFloat32,
              "ncclDataType_t values differ from the ones hs_search_batch_sharded passes to the dlopen'ed RCCL");
extern "C" {
#endif

void hs_comm_free(hs_comm *c) {
  if (!c) return;
  for (auto &sl : c->slots)
    for (int r = 0; r < (int)c->dev.size(); r++) {
      (void)hipSetDevice(c->dev[r]);
      if (r < (int)sl.d.size()) sl.d[r].reset();
      if (r < (int)sl.done.size() && sl.done[r]) (void)hipEventDestroy(sl.done[r]);
      if (r < (int)sl.streams.size() && sl.streams[r]) (void)hipStreamDestroy(sl.streams[r]);
    }
  for (int r = 0; r < (int)c->dev.size(); r++) {
    (void)hipSetDevice(c->dev[r]);
    if (r < (int)c->comms.size() && c->comms[r] && c->CommDestroy) c->CommDestroy(c->comms[r]);
  }
  if (c->lib) dlclose(c->lib);
  delete c;
}

static hs_status comm_slot_ready(hs_comm *c, int slot) {   // streams / events / buffers of a slot are created on first use
  hs_comm::Slot &sl = c->slots[slot];
  if (!sl.streams.empty()) return HS_OK;
  sl.streams.assign(c->n, nullptr);
  sl.done.assign(c->n, nullptr);
  sl.rows.assign(c->n, 0);
  for (int r = 0; r < c->n; r++) {
    HIP_TRY(hipSetDevice(c->dev[r]));
    HIP_TRY(hipStreamCreateWithFlags(&sl.streams[r], hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&sl.done[r], hipEventDisableTiming));
    sl.d.emplace_back(new hs_comm::Dev());
  }
  return HS_OK;
}

#define NCCL_TRY(c, expr)                                                                                  \
  do {                                                                                                     \
    const int _rc = (expr);                                                                                \
    if (_rc != 0) return fail(HS_ERR_DEVICE, std::string(#expr ": ") + ((c)->GetErrorString ? (c)->GetErrorString(_rc) : "RCCL error")); \
  } while (0)

// The first real exchange of a communicator: 16 words per device through the same in-place all-gather the search uses, checked
// on every device -- a wrong datatype constant, a mismatched library or a broken link shows here, not in a result array.
static hs_status comm_selftest(hs_comm *c) {
  const int n = c->n;
  hs_status st = comm_slot_ready(c, 0);
  if (st != HS_OK) return st;
  hs_comm::Slot &sl = c->slots[0];
  std::vector<uint32_t> h((size_t)n * 16);
  for (int r = 0; r < n; r++) {
    HIP_TRY(hipSetDevice(c->dev[r]));
    HIP_TRY(sl.d[r]->all_l32.ensure((size_t)n * 16));
    for (int i = 0; i < n * 16; i++) h[i] = (i / 16 == r) ? 0xA5000000u + (uint32_t)(r * 16 + i % 16) : 0u;
    HIP_TRY(hipMemcpyAsync(sl.d[r]->all_l32.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, sl.streams[r]));
    HIP_TRY(hipStreamSynchronize(sl.streams[r]));
  }
  int first_err = 0;
  NCCL_TRY(c, c->GroupStart());
  for (int r = 0; r < n; r++) {
    const int rc = c->AllGather(sl.d[r]->all_l32.p + (size_t)r * 16, sl.d[r]->all_l32.p, 16, kNcclUint32, c->comms[r], sl.streams[r]);
    if (rc != 0 && first_err == 0) first_err = rc;
  }
  const int rc_end = c->GroupEnd();
  if (first_err == 0) first_err = rc_end;
  for (int r = 0; r < n; r++) {
    (void)hipSetDevice(c->dev[r]);
    (void)hipStreamSynchronize(sl.streams[r]);
  }
  if (first_err != 0) return fail(HS_ERR_DEVICE, std::string("RCCL all-gather self-test: ") + (c->GetErrorString ? c->GetErrorString(first_err) : "error"));
  for (int r = 0; r < n; r++) {
    HIP_TRY(hipSetDevice(c->dev[r]));
    HIP_TRY(hipMemcpy(h.data(), sl.d[r]->all_l32.p, h.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n * 16; i++)
      if (h[i] != 0xA5000000u + (uint32_t)i)
        return fail(HS_ERR_DEVICE, "RCCL all-gather self-test: device " + std::to_string(c->dev[r]) + " holds a wrong word at " + std::to_string(i));
  }
  return HS_OK;
}

hs_status hs_comm_init(int n_gpus, const int *devices, hs_comm **out) {
  if (!out || n_gpus < 1 || n_gpus > 64) return fail(HS_ERR_INVALID, "bad argument");
  const int have = hs_device_count();
  if (have < 1) return fail(HS_ERR_DEVICE, "no HIP device (this library has no CPU search path)");
  std::unique_ptr<hs_comm, void (*)(hs_comm *)> c(new hs_comm(), hs_comm_free);
  c->n = n_gpus;
  for (int r = 0; r < n_gpus; r++) {
    const int d = devices ? devices[r] : r;
    if (d < 0 || d >= have) return fail(HS_ERR_DEVICE, "device " + std::to_string(d) + " not present (" + std::to_string(have) + " visible)");
    for (int t = 0; t < r; t++) c->loopback = c->loopback || c->dev[t] == d;
    c->dev.push_back(d);
  }
  c->slots.resize(kCommSlots);
  hs_status st = comm_slot_ready(c.get(), 0);
  if (st != HS_OK) return st;
  if (n_gpus > 1 && !c->loopback) {
    c->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!c->lib) c->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!c->lib) return fail(HS_ERR_UNSUPPORTED, std::string("RCCL not found: ") + dlerror());
    c->CommInitAll = (int (*)(void **, int, const int *))dlsym(c->lib, "ncclCommInitAll");
    c->CommDestroy = (int (*)(void *))dlsym(c->lib, "ncclCommDestroy");
    c->AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(c->lib, "ncclAllGather");
    c->GroupStart = (int (*)())dlsym(c->lib, "ncclGroupStart");
    c->GroupEnd = (int (*)())dlsym(c->lib, "ncclGroupEnd");
    c->GetErrorString = (const char *(*)(int))dlsym(c->lib, "ncclGetErrorString");
    if (!c->CommInitAll || !c->CommDestroy || !c->AllGather || !c->GroupStart || !c->GroupEnd)
      return fail(HS_ERR_UNSUPPORTED, "RCCL library lacks the expected entry points");
    c->comms.assign(n_gpus, nullptr);
    const int rc = c->CommInitAll(c->comms.data(), n_gpus, c->dev.data());
    if (rc != 0) return fail(HS_ERR_DEVICE, std::string("ncclCommInitAll: ") + (c->GetErrorString ? c->GetErrorString(rc) : "error"));
    st = comm_selftest(c.get());
    if (st != HS_OK) return st;
  }
  *out = c.release();
  return HS_OK;
}

int hs_comm_size(const hs_comm *c) { return c ? c->n : 0; }
int hs_comm_slots(const hs_comm *c) { return c ? (int)c->slots.size() : 0; }

// Waits for the batch in flight in `slot` and reports its capacity errors (hs_search_check on every device that searched).
hs_status hs_comm_check(hs_comm *c, hs_index *const *ixs, int slot) {
  if (!c || !ixs || slot < 0 || slot >= (int)c->slots.size()) return fail(HS_ERR_INVALID, "bad argument");
  hs_comm::Slot &sl = c->slots[slot];
  if (!sl.busy) return HS_OK;
  sl.busy = false;
  hs_status worst = HS_OK;
  std::string msg;
  for (int r = 0; r < c->n; r++) {
    if (!ixs[r]) return fail(HS_ERR_INVALID, "null index replica");
    hs_status s = HS_OK;
    if (sl.rows[r]) {
      s = hs_search_check(ixs[r], sl.streams[r]);   // synchronises the stream, reads and clears its counters
    } else {   // nothing was searched on this device in this batch: only the exchange and the copies ran on its stream
      if (hipSetDevice(c->dev[r]) != hipSuccess || hipStreamSynchronize(sl.streams[r]) != hipSuccess) s = fail(HS_ERR_DEVICE, "hipStreamSynchronize failed");
    }
    if (s != HS_OK && worst == HS_OK) { worst = s; msg = g_err; }
  }
  if (worst != HS_OK) return fail(worst, msg);
  return HS_OK;
}

// Shard r = rows [r * S, min(nq, (r + 1) * S)), S = ceil(nq / n): device r searches its shard straight into slot r of
// its [n * S x k] gather buffers, the in-place all-gather completes the other slots, device 0's copy goes to the host.
// Enqueue only: queries and outputs (page-locked memory if the copies are to overlap) must stay valid until hs_comm_check(slot).
hs_status hs_search_batch_sharded_async(hs_comm *c, hs_index *const *ixs, const float *queries, size_t nq, size_t k, int mode,
                                        uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists, uint32_t *out_counts, int slot) {
  if (!c || !ixs || !queries) return fail(HS_ERR_INVALID, "null argument");
  if (slot < 0 || slot >= (int)c->slots.size()) return fail(HS_ERR_INVALID, "bad slot");
  if (mode == HS_MODE_SLIM_IDS && !out_labels32) return fail(HS_ERR_INVALID, "out_labels32 required");
  if (mode == HS_MODE_PQ && (!out_labels64 || !out_dists || !out_counts)) return fail(HS_ERR_INVALID, "out_labels64/out_dists/out_counts required");
  if (mode != HS_MODE_SLIM_IDS && mode != HS_MODE_PQ) return fail(HS_ERR_INVALID, "bad mode");
  if (k == 0) return fail(HS_ERR_INVALID, "k must be > 0");
  const int n = c->n;
  for (int r = 0; r < n; r++) {
    if (!ixs[r]) return fail(HS_ERR_INVALID, "null index replica");
    if (ixs[r]->device != c->dev[r]) return fail(HS_ERR_INVALID, "replica " + std::to_string(r) + " is not on the communicator's device");
    if (ixs[r]->info.n != ixs[0]->info.n || ixs[r]->info.dim != ixs[0]->info.dim || ixs[r]->info.kind != ixs[0]->info.kind)
      return fail(HS_ERR_INVALID, "replicas differ");
  }
  if (c->slots[slot].busy) return fail(HS_ERR_INVALID, "slot " + std::to_string(slot) + " still holds a batch: hs_comm_check it first");
  if (nq == 0) return HS_OK;
  hs_status rs = comm_slot_ready(c, slot);
  if (rs != HS_OK) return rs;
  hs_comm::Slot &sl = c->slots[slot];
  c->last_slot = slot;
  const size_t dim = ixs[0]->info.dim, S = (nq + n - 1) / n;
  const bool ids = mode == HS_MODE_SLIM_IDS;
  const bool want_d = out_dists != nullptr, want_c = out_counts != nullptr || !ids;
  for (int r = 0; r < n; r++) {
    HIP_TRY(hipSetDevice(c->dev[r]));
    hs_comm::Dev &d = *sl.d[r];
    HIP_TRY(d.q.ensure(S * dim));
    if (ids) HIP_TRY(d.all_l32.ensure((size_t)n * S * k));
    else HIP_TRY(d.all_l64.ensure((size_t)n * S * k));
    if (want_d || !ids) HIP_TRY(d.all_dist.ensure((size_t)n * S * k));
    HIP_TRY(d.all_cnt.ensure((size_t)n * S));
    const size_t lo = std::min(nq, (size_t)r * S), m = std::min(nq, lo + S) - lo;
    sl.rows[r] = m;
    hipStream_t st = sl.streams[r];
    if (m < S) {   // a short (or empty) last shard: its padding rows must not be garbage in the gathered arrays
      if (ids) HIP_TRY(hipMemsetAsync(d.all_l32.p + (size_t)r * S * k, 0xFF, S * k * 4, st));
      else HIP_TRY(hipMemsetAsync(d.all_l64.p + (size_t)r * S * k, 0xFF, S * k * 8, st));
      if (want_d || !ids) HIP_TRY(hipMemsetAsync(d.all_dist.p + (size_t)r * S * k, 0x7F, S * k * 4, st));   // 0x7F7F7F7F: a large finite float
      HIP_TRY(hipMemsetAsync(d.all_cnt.p + (size_t)r * S, 0, S * 4, st));
    }
    if (m) {
      HIP_TRY(hipMemcpyAsync(d.q.p, queries + lo * dim, m * dim * sizeof(float), hipMemcpyHostToDevice, st));
      hs_status s = search_dev(ixs[r], d.q.p, m, k, mode, ids ? d.all_l32.p + (size_t)r * S * k : nullptr,
                               ids ? nullptr : d.all_l64.p + (size_t)r * S * k, (want_d || !ids) ? d.all_dist.p + (size_t)r * S * k : nullptr,
                               d.all_cnt.p + (size_t)r * S, nullptr, nullptr, nullptr, st);
      if (s != HS_OK) return s;
    }
    if (c->loopback) HIP_TRY(hipEventRecord(sl.done[r], st));
  }
  sl.busy = true;
  if (n > 1 && !c->loopback) {
    // (an error inside the group must not leave it open: every call is made, the first error is reported after GroupEnd)
    int first_err = 0;
    NCCL_TRY(c, c->GroupStart());
    auto ag = [&](const void *src, void *dst, size_t cnt, int ty, int r) {
      const int rc = c->AllGather(src, dst, cnt, ty, c->comms[r], sl.streams[r]);
      if (rc != 0 && first_err == 0) first_err = rc;
    };
    for (int r = 0; r < n; r++) {
      hs_comm::Dev &d = *sl.d[r];
      if (ids) ag(d.all_l32.p + (size_t)r * S * k, d.all_l32.p, S * k, kNcclUint32, r);
      else ag(d.all_l64.p + (size_t)r * S * k, d.all_l64.p, S * k, kNcclUint64, r);
      if (want_d || !ids) ag(d.all_dist.p + (size_t)r * S * k, d.all_dist.p, S * k, kNcclFloat32, r);
      if (want_c) ag(d.all_cnt.p + (size_t)r * S, d.all_cnt.p, S, kNcclUint32, r);
    }
    const int rc_end = c->GroupEnd();
    if (first_err == 0) first_err = rc_end;
    if (first_err != 0) {
      for (int r = 0; r < n; r++) {
        (void)hipSetDevice(c->dev[r]);
        (void)hipStreamSynchronize(sl.streams[r]);
      }
      sl.busy = false;
      return fail(HS_ERR_DEVICE, std::string("ncclAllGather: ") + (c->GetErrorString ? c->GetErrorString(first_err) : "RCCL error"));
    }
  } else if (n > 1) {
    // rehearsal on one device: the same exchange as stream-ordered device copies (slot s of every rank <- slot s of rank s)
    for (int r = 0; r < n; r++) {
      HIP_TRY(hipSetDevice(c->dev[r]));
      hs_comm::Dev &d = *sl.d[r];
      hipStream_t st = sl.streams[r];
      for (int s2 = 0; s2 < n; s2++) {
        if (s2 == r) continue;
        hs_comm::Dev &o = *sl.d[s2];
        HIP_TRY(hipStreamWaitEvent(st, sl.done[s2], 0));
        if (ids) HIP_TRY(hipMemcpyAsync(d.all_l32.p + (size_t)s2 * S * k, o.all_l32.p + (size_t)s2 * S * k, S * k * 4, hipMemcpyDeviceToDevice, st));
        else HIP_TRY(hipMemcpyAsync(d.all_l64.p + (size_t)s2 * S * k, o.all_l64.p + (size_t)s2 * S * k, S * k * 8, hipMemcpyDeviceToDevice, st));
        if (want_d || !ids) HIP_TRY(hipMemcpyAsync(d.all_dist.p + (size_t)s2 * S * k, o.all_dist.p + (size_t)s2 * S * k, S * k * 4, hipMemcpyDeviceToDevice, st));
        if (want_c) HIP_TRY(hipMemcpyAsync(d.all_cnt.p + (size_t)s2 * S, o.all_cnt.p + (size_t)s2 * S, S * 4, hipMemcpyDeviceToDevice, st));
      }
    }
  }
  // the host takes the first nq rows of device 0's gathered arrays
  {
    HIP_TRY(hipSetDevice(c->dev[0]));
    hs_comm::Dev &d = *sl.d[0];
    hipStream_t st = sl.streams[0];
    if (out_labels32) HIP_TRY(hipMemcpyAsync(out_labels32, d.all_l32.p, nq * k * 4, hipMemcpyDeviceToHost, st));
    if (out_labels64) HIP_TRY(hipMemcpyAsync(out_labels64, d.all_l64.p, nq * k * 8, hipMemcpyDeviceToHost, st));
    if (out_dists) HIP_TRY(hipMemcpyAsync(out_dists, d.all_dist.p, nq * k * 4, hipMemcpyDeviceToHost, st));
    if (out_counts) HIP_TRY(hipMemcpyAsync(out_counts, d.all_cnt.p, nq * 4, hipMemcpyDeviceToHost, st));
  }
  return HS_OK;
}
// The synchronous form: slot 0, enqueue + wait.
hs_status hs_search_batch_sharded(hs_comm *c, hs_index *const *ixs, const float *queries, size_t nq, size_t k, int mode,
                                  uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists, uint32_t *out_counts) {
  hs_status s = hs_search_batch_sharded_async(c, ixs, queries, nq, k, mode, out_labels32, out_labels64, out_dists, out_counts, 0);
  if (s != HS_OK) return s;
  return hs_comm_check(c, ixs, 0);
}
// The gathered result arrays as device `rank` holds them after the most recent sharded call ([n * ceil(nq / n) x k], valid until
// that slot's next call): every device has the whole batch's top-k, e.g. for a re-ranking stage that runs on all of them.
hs_status hs_comm_results_dev(hs_comm *c, int rank, const uint32_t **d_labels32, const uint64_t **d_labels64, const float **d_dists,
                              const uint32_t **d_counts) {
  if (!c || rank < 0 || rank >= c->n) return fail(HS_ERR_INVALID, "bad argument");
  if (c->slots[c->last_slot].d.empty()) return fail(HS_ERR_INVALID, "no sharded call yet");
  hs_comm::Dev &d = *c->slots[c->last_slot].d[rank];
  if (d_labels32) *d_labels32 = d.all_l32.p;
  if (d_labels64) *d_labels64 = d.all_l64.p;
  if (d_dists) *d_dists = d.all_dist.p;
  if (d_counts) *d_counts = d.all_cnt.p;
  return HS_OK;
}

}  // extern "C"
