/* hnsw_slim_amd.h -- C ABI of the MI355X-native HNSW / HNSW-Slim batched search engine.
 *
 * This is the drop-in boundary for the reference's searchKnn -> searchBaseLayerST -> distance path
 * (SURVEY.md section 8b).  Plain pointers and sizes only; every entry point names the reference
 * interface it replaces (paths relative to /root/reference/third_party/hnswlib/).  The C++ facade
 * hnsw-slim_amd/hnswlib/hnswlib_amd.h re-creates hnswlib::HierarchicalNSW / HierarchicalNSWSlim /
 * L2Space / InnerProductSpace on top of these calls.
 *
 * Error convention: every call returns an hs_status; hs_last_error() gives the thread-local message,
 * which reuses the reference's exception texts ("Cannot open file", "Index seems to be corrupted or
 * unsupported", ...) so the facade can re-throw std::runtime_error with the same what().
 * There is NO CPU fallback: without a HIP device every search entry point fails with HS_ERR_DEVICE.
 */
#ifndef HNSW_SLIM_AMD_H
#define HNSW_SLIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hs_index hs_index;

typedef enum {
  HS_OK = 0,
  HS_ERR_IO = 1,          /* "Cannot open file"                         hnswalg.h:785, hnswalg_slim.h:757 */
  HS_ERR_CORRUPT = 2,     /* "Index seems to be corrupted or unsupported" hnswalg.h:823-836 */
  HS_ERR_NOMEM = 3,       /* "Not enough memory: ..."                   hnswalg_slim.h:785-788 */
  HS_ERR_INVALID = 4,     /* bad argument                                */
  HS_ERR_UNSUPPORTED = 5, /* e.g. SlimQ with dim < 64, brute force k > 64  */
  HS_ERR_DEVICE = 6,      /* HIP runtime error / no device               */
  HS_ERR_CAPACITY = 7     /* a query outgrew even the fallback on-chip scratch */
} hs_status;

typedef enum { HS_KIND_HNSW = 0, HS_KIND_SLIM = 1, HS_KIND_SLIMQ = 2 } hs_kind;   /* hnswalg.h:18 / hnswalg_slim.h:29 / hnswalg_slimq.h:41 */
typedef enum { HS_METRIC_L2 = 0, HS_METRIC_IP = 1 } hs_metric; /* space_l2.h:208 / space_ip.h:342  */

/* Which searchKnn overload a batch call reproduces. */
typedef enum {
  /* HierarchicalNSWSlim::searchKnn(const void*, size_t k, tableint* result)  hnswalg_slim.h:2030-2131:
   * k uint32 labels per query in the reference's post-nth_element array order. */
  HS_MODE_SLIM_IDS = 0,
  /* priority_queue-returning overloads: HierarchicalNSW::searchKnn hnswalg.h:1378-1440 and
   * HierarchicalNSWSlim::searchKnn(q,k) hnswalg_slim.h:1907-2028: <=k (dist,label) pairs per query. */
  HS_MODE_PQ = 1
} hs_mode;

typedef struct {
  uint64_t n, dim;
  int32_t kind, metric, maxlevel, threshold_level;
  uint32_t enterpoint;
  int32_t has_deleted;
  uint64_t n_edges;      /* entries of the CSR column array (all levels) */
  uint64_t device_bytes; /* HBM held by this index */
  uint64_t max_degree0;
  uint64_t index_size;   /* the reference's indexSize() for this index: hnswalg.h:1533-1547, hnswalg_slim.h:2435-2444,
                            hnswalg_slimq.h:2047-2057 (graph structure of the CPU layout, without the vectors) */
} hs_info;

const char *hs_last_error(void);

/* Number of HIP devices visible (0 when none); does not initialise a device context. */
int hs_device_count(void);

/* loadIndex(path, space, max_elements): hnswalg.h:781-893 (HS_KIND_HNSW), hnswalg_slim.h:753-815
 * (HS_KIND_SLIM), hnswalg_slimq.h:1218-1313 (HS_KIND_SLIMQ).  Parses the reference's serialized index, repacks it
 * (row-major vectors + CSR adjacency + fixed-stride tiles; RaBitQ records for SlimQ) and uploads it to HIP device
 * `device`.  ef starts at 10 as in the reference.  Both metrics take every dim. */
hs_status hs_index_load(const char *path, int kind, int metric, size_t dim, size_t max_elements,
                        int device, hs_index **out);
/* The same, from the serialized bytes already in host memory (what saveIndex would have written: hnswalg.h:748-779,
 * hnswalg_slim.h:717-751, hnswalg_slimq.h:1161-1216) -- for hosts that receive the index over a socket
 * (hnsw_slim_server.cc:59-81) or keep it in a buffer; nothing is written to disk.  The bytes are not retained. */
hs_status hs_index_load_mem(const void *bytes, size_t len, int kind, int metric, size_t dim, size_t max_elements,
                            int device, hs_index **out);
/* Upload an index the host has already parsed (no file involved): what a caller that holds the reference's public members
 * (hnswalg_slim.h:30 onwards, all public) or its own graph hands over.  Node i owns levels[i] + 1 consecutive neighbour lists,
 * level 0 first; list t = list_ids[list_ptr[t] .. list_ptr[t+1]).  labels NULL = row index, deleted NULL = none marked.
 * kind: HS_KIND_HNSW or HS_KIND_SLIM (selects the searchKnn overload semantics); threshold_level applies to Slim. */
hs_status hs_index_from_host_arrays(int kind, int metric, size_t n, size_t dim, const float *vectors, const uint64_t *labels,
                                    const uint8_t *deleted, const int32_t *levels, const uint64_t *list_ptr,
                                    const uint32_t *list_ids, uint32_t enterpoint, int32_t maxlevel, int32_t threshold_level,
                                    int device, hs_index **out);
/* patchFromStream(std::istream&, bool to_add): hnswalg_slim.h:2292-2340, wire format of genPatch :1427-1476 as the reference's
 * server frames it (hnsw_slim_server_patch.cc:280-290, after its `finished` word): u64 cur_element_count, u64 changed_old_cnt,
 * u64 changed_new_cnt, then per changed node  u32 id | 8 B {level, total_neighbor} (old node) or 16 B {level, total, label}
 * (new node) | u32 neighborsSize | blob | the vector (new node, to_add).  Applies to a HS_KIND_SLIM index that was loaded with
 * max_elements > its element count (the reference needs the same room, hnswalg_slim.h:784).  Only the changed nodes' rows of the
 * vectors and level-0 tiles are rewritten in HBM; the small structure arrays are rebuilt.  Like the reference, the stream
 * does not move the enter point.  Not to be called while a search on this index is in flight. */
hs_status hs_index_patch(hs_index *ix, const void *bytes, size_t len, int to_add);
void hs_index_free(hs_index *ix);                        /* ~HierarchicalNSW* / clear(): hnswalg_slim.h:154-167 */
hs_status hs_set_ef(hs_index *ix, size_t ef);            /* setEf: hnswalg.h:184, hnswalg_slim.h:193 */
hs_status hs_index_info(const hs_index *ix, hs_info *out);
/* Name of the device kernel that served pass 0 of the most recent search call on this index ("hs::flat_kernel", ...): the kernel
 * a rocprofv3 kernel trace of that call shows; for measurement scripts, no counterpart in the reference. */
const char *hs_last_kernel(const hs_index *ix);

/* Output-order policy of the result set.  0 (default): the fast kernel answers, each query's entries
 * come out sorted by ascending distance; the k-subset (ids and distances) is exactly the reference's --
 * queries where the reference's choice depends on its heap layout (a distance tie across the k-th
 * boundary) are detected and re-run with the reference's heap mechanics.  1: every query runs the strict
 * kernel and HS_MODE_SLIM_IDS reproduces the reference's post-nth_element array ORDER as well
 * (hnswalg_slim.h:2126-2130 leaves an unordered k-subset). */
hs_status hs_set_exact_order(hs_index *ix, int on);

/* On-chip scratch sizing per query (0 = automatic from ef): candidate-heap capacity and visited-set
 * hash slots (power of two).  Queries that outgrow it are re-run with a whole CU's LDS. */
hs_status hs_set_capacity(hs_index *ix, uint32_t cand_cap, uint32_t hash_slots);

/* Batched searchKnn over nq host-resident queries (nq x dim, row-major fp32).  Outputs (host):
 *   mode HS_MODE_SLIM_IDS: out_labels32[nq*k] (required); out_dists[nq*k] (nullable) in the same order.
 *   mode HS_MODE_PQ      : out_labels64[nq*k] + out_dists[nq*k] (required): the pairs left in
 *                          top_candidates after popping down to k, heap-array order; out_counts[nq].
 *   out_counts[nq] (nullable): number of valid entries per query (min(k, found)); unused slots hold
 *                          0xFFFFFFFF / UINT64_MAX / +inf.
 *   stats (nullable): nq x 4 uint32 {n_dist, n_hops, n_nbr_read, pass}  (SURVEY.md 8d); pass = 0 first
 *                          pass, 1 tie re-run (strict kernel), 2 scratch-overflow re-run.
 * Synchronous: includes H2D of queries and D2H of results. */
hs_status hs_search_batch(hs_index *ix, const float *queries, size_t nq, size_t k, int mode,
                          uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists,
                          uint32_t *out_counts, uint32_t *stats);

/* searchKnn(q, k, BaseFilterFunctor* isIdAllowed): hnswalg.h:1378-1440 with :347-349,441-444, and
 * hnswalg_slim.h:1783-1905 with :462-618.  The functor is a host callback, so the caller evaluates it once
 * per element: allowed[i] != 0 iff (*isIdAllowed)(label of internal id i) (hs_labels() gives the labels).
 * Always the priority_queue result shape (HS_MODE_PQ outputs).  Slim indexes: threshold_level == 0 only. */
hs_status hs_search_batch_filtered(hs_index *ix, const float *queries, size_t nq, size_t k,
                                   const uint8_t *allowed, uint64_t *out_labels64, float *out_dists,
                                   uint32_t *out_counts, uint32_t *stats);
/* External labels by internal id (n entries), to evaluate a filter functor on the host. */
hs_status hs_labels(const hs_index *ix, uint64_t *out_labels);

/* Same search with DEVICE pointers, asynchronous on `stream` (a hipStream_t; NULL = default stream).
 * No host synchronisation happens here; call hs_search_check() after synchronising to learn whether
 * any query exhausted the fallback scratch. */
hs_status hs_search_batch_dev(hs_index *ix, const float *d_queries, size_t nq, size_t k, int mode,
                              uint32_t *d_out_labels32, uint64_t *d_out_labels64, float *d_out_dists,
                              uint32_t *d_out_counts, uint32_t *d_stats, void *stream);
hs_status hs_search_check(hs_index *ix, void *stream);
/* Parity/debug entry: a sequence of candidate_set operations (std::push_heap / std::pop_heap with compare_by_first_rev,
 * hnswalg_slim.h:177-183, 331-332, 353-354, 408-411) through the flat kernel's heap code on the device.  ops: 3 words each
 * {0 = push | 1 = pop, distance bits, id}; wave_pop selects the whole-wave pop; lds_slots: heap slots kept in LDS (the rest
 * in global memory).  out_heap / out_pops: 2 words per entry (n_ops entries of room each); out_n: {final size, pops}. */
/* Parity/debug entry (host only, no device needed): the flat kernel's visited-set plan for an index of n nodes at (ef, queries per
 * call): out5 = {buckets nb, multiplier m, shift s, id-space bits B, ok}; bucket = h mod nb and remainder = h div nb =
 * umulhi(h, m) >> s must be exact for every h < 2^B and the remainders must fit 15 bits (tests/test_host_cpu.py). */
hs_status hs_debug_flat_plan(size_t n, size_t ef, size_t nq, uint32_t *out5);
hs_status hs_debug_heap_ops(const uint32_t *ops, size_t n_ops, int wave_pop, uint32_t lds_slots, uint32_t *out_heap, uint32_t *out_pops,
                            uint32_t *out_n);

/* Host pointers, asynchronous: H2D of the queries, the search and D2H of the requested outputs are enqueued on `stream`
 * and nothing is valid until that stream is synchronised (hs_search_check does it and reports capacity problems).  The
 * serving shape of the reference's query loop (include/strategy/hnsw_slim_strategy.h:107-118: the clock runs around the
 * whole loop, queries in, labels out): batches issued round-robin on a few streams overlap each other's copies and
 * kernels.  queries and outputs should be page-locked (hs_host_alloc, or hipHostMalloc / hipHostRegister of the
 * caller's own buffers): with pageable memory the copies are staged synchronously.  Staging buffers are per
 * (index, stream): do not reuse a stream for a second call on the same index before the first one's outputs are read.
 * Small batches (up to 2 MiB of queries) in device-mapped page-locked buffers are served IN PLACE: no staging copies, the
 * kernels read each query from `queries` and write the results into the output buffers directly -- so `queries` must stay
 * unchanged, and the outputs unread, until the batch has completed on `stream` (as for the copies, only for longer). */
hs_status hs_search_batch_async(hs_index *ix, const float *queries, size_t nq, size_t k, int mode,
                                uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists,
                                uint32_t *out_counts, uint32_t *stats, void *stream);
void *hs_host_alloc(size_t bytes);   /* page-locked host memory (NULL on failure) */
void hs_host_free(void *p);
/* The device's address of a page-locked host buffer that is mapped into the device's address space (hs_host_alloc, hipHostMalloc,
 * mapped hipHostRegister), or NULL (pageable memory).  hs_search_batch_async serves small batches from such buffers in place;
 * a caller of hs_search_batch_dev may pass this address as d_queries for the same effect (the kernels read each query once). */
void *hs_host_device_pointer(const void *host);

/* ---- multi-GPU (no counterpart in the reference, which has no notion of a device) --------------------
 * Queries are independent and the index is read-only during search: the index is REPLICATED (hs_index_load once per
 * device), a batch is split into contiguous shards [r*S, (r+1)*S), S = ceil(nq/n), device r searches shard r, and one
 * RCCL all-gather of the packed [S x k] results over xGMI leaves the whole [nq x k] result on every device and on the
 * host.  One process, one stream per device; parity = the single-device result, bit for bit.
 * hs_comm_init: devices = n_gpus HIP device ordinals (NULL: 0..n-1).  Listing one device more than once is the
 * one-GPU rehearsal mode (the exchange then runs as device copies instead of RCCL, which refuses duplicate devices).
 * With more than one real device the first exchange happens here: a 64-byte all-gather per device, verified on every device
 * (HS_ERR_DEVICE if RCCL, its datatype constants or a link are not what the search will rely on). */
typedef struct hs_comm hs_comm;
hs_status hs_comm_init(int n_gpus, const int *devices, hs_comm **out);
void hs_comm_free(hs_comm *c);
int hs_comm_size(const hs_comm *c);
/* ixs: n_gpus replicas, ixs[r] loaded on the communicator's r-th device; queries and outputs are host pointers (outputs
 * as for hs_search_batch; out_dists / out_counts nullable in HS_MODE_SLIM_IDS).  Synchronous. */
hs_status hs_search_batch_sharded(hs_comm *c, hs_index *const *ixs, const float *queries, size_t nq, size_t k, int mode,
                                  uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists, uint32_t *out_counts);
/* The same in two halves, for callers that keep several batches in flight per device (a split batch is a small launch on every
 * device and lasts as long as its longest query; the chip fills up with several of them): hs_search_batch_sharded_async only
 * ENQUEUES the step (H2D of the shards, search, all-gather, D2H of device 0's copy) on the streams of `slot`
 * (0 <= slot < hs_comm_slots()); hs_comm_check(slot) waits for it and reports capacity errors.  queries and outputs must stay
 * valid (page-locked if the copies are to overlap) until then; a slot holds one batch at a time. */
int hs_comm_slots(const hs_comm *c);
hs_status hs_search_batch_sharded_async(hs_comm *c, hs_index *const *ixs, const float *queries, size_t nq, size_t k, int mode,
                                        uint32_t *out_labels32, uint64_t *out_labels64, float *out_dists, uint32_t *out_counts, int slot);
hs_status hs_comm_check(hs_comm *c, hs_index *const *ixs, int slot);
/* device `rank`'s copy of the gathered arrays of the last hs_search_batch_sharded call ([n_gpus * S x k]; valid until the next call) */
hs_status hs_comm_results_dev(hs_comm *c, int rank, const uint32_t **d_labels32, const uint64_t **d_labels64,
                              const float **d_dists, const uint32_t **d_counts);

/* Parity/debug entry: raw top_candidates arrays after the level-0 beam, exactly as the reference holds
 * them before selection (hnswalg_slim.h:2116-2124): raw_dists/raw_ids are nq x max(ef,k), raw_sizes nq.
 * mark_ep_visited selects the (q,k)/(q,k,filter) overloads' extra visited tag (hnswalg_slim.h:1796,1919). */
hs_status hs_search_batch_raw(hs_index *ix, const float *queries, size_t nq, size_t k, int mode,
                              float *raw_dists, uint32_t *raw_ids, uint32_t *raw_sizes, uint32_t *stats);

/* ---- HNSW-SlimQ: HierarchicalNSWSlimQ (hnswalg_slimq.h), the RaBitQ-quantised variant --------------------------
 * hs_index_load(path, HS_KIND_SLIMQ, ..) parses saveIndex's format (hnswalg_slimq.h:1161-1313).  The search is
 * searchKnn(query, k, tableint *result) (hnswalg_slimq.h:1810-1924): greedy descent and a SearchBuffer beam of
 * capacity ef (hs_set_ef = setEf, :346-349) on ESTIMATED distances, every expanded node re-ranked with its exact
 * distance to the raw row of setDataset() and kept in a k-bounded max-heap.  out_labels/out_dists are nq x k in the
 * order the reference reads them out (the heap ARRAY, :1921-1923), ~0 / +inf beyond out_counts[q] entries (the
 * reference leaves those slots undefined).  stats: nq x 4 {expansions, estimates, buffer inserts, revisits}.
 * Float reductions that the reference runs through Eigen (alignment/width dependent order) are defined as
 * left-to-right fp32 sums; see DESIGN.md "SlimQ". */
hs_status hs_slimq_set_dataset(hs_index *ix, const float *base, size_t n, size_t dim);   /* setDataset, :303-305 */
/* t_const of quant::faster_config (the reference draws it from std::random_device at load time, :1274-1276;
 * default here: hs_rabitq_default_tconst(padded_dim, 1)). */
hs_status hs_slimq_set_tconst(hs_index *ix, double t_const);
double hs_slimq_get_tconst(const hs_index *ix);
hs_status hs_slimq_search_batch(hs_index *ix, const float *queries, size_t nq, size_t k, uint64_t *out_labels,
                                float *out_dists, uint32_t *out_counts, uint32_t *stats);
/* device pointers + HIP stream; asynchronous, hs_search_check(ix, stream) reports capacity problems */
hs_status hs_slimq_search_batch_dev(hs_index *ix, const float *d_queries, size_t nq, size_t k, uint64_t *d_out_labels,
                                    float *d_out_dists, uint32_t *d_out_counts, uint32_t *d_stats, void *stream);

/* Parity/debug entry: the query preparation as the kernel computed it; out is nq x (padded + 3 + num_cluster +
 * padded/8) floats: rotated query, {delta, vl, k1xsumq}, g_add per cluster, the 4 bit planes per 64-dim block as raw
 * u32 pairs (rotator.hpp:370-423, query.hpp:112-156, hnswalg_slimq.h:1822-1848). */
hs_status hs_slimq_prepare_debug(hs_index *ix, const float *queries, size_t nq, float *out);
/* Parity/debug entry: the SearchBuffer events of every query in order, two words each (out_trace nq x trace_cap,
 * 0xFFFFFFFF padding): a pop = {node id (bit 31 set when the node had been expanded before, hnswalg_slimq.h:696-704),
 * buffer size}, an insert = {candidate id | 1<<30, bits of its estimated distance} (:745). */
hs_status hs_slimq_trace(hs_index *ix, const float *queries, size_t nq, size_t k, uint32_t *out_trace, size_t trace_cap,
                         uint32_t *stats);

/* ---- exhaustive k-NN (ground truth): hnswlib::BruteforceSearch::searchKnn, bruteforce.h:106-135, for a batch ------
 * Result per query: the k lexicographically smallest (dist, label) pairs -- what the reference's priority_queue of
 * pairs ends up holding whatever the scan order -- sorted ascending; distances by the same fp32 recipes as the
 * graph search.  labels NULL = row index.  dim <= 4096 (dim % 16 == 0 is the tuned kernel), k <= 64.  out_counts[q] = min(k, n). */
hs_status hs_brute_force(const float *base, size_t n, size_t dim, int metric, const uint64_t *labels, const float *queries,
                         size_t nq, size_t k, int device, uint64_t *out_labels, float *out_dists, uint32_t *out_counts);
/* device pointers; synchronises `stream` before returning */
hs_status hs_brute_force_dev(const float *d_base, const uint64_t *d_labels, size_t n, size_t dim, int metric,
                             const float *d_queries, size_t nq, size_t k, uint64_t *d_out_labels, float *d_out_dists,
                             uint32_t *d_out_counts, void *stream);

/* ---- harness (CPU, not accelerated): produce index files in the reference's formats ------------ */
/* HierarchicalNSW ctor + addPoint loop + saveIndex: hnswalg.h:85-159, 1248-1376, 748-779.
 * labels = row index; threads==1 reproduces the reference's serial build byte for byte. */
hs_status hs_build_hnsw(const float *base, size_t n, size_t dim, int metric, size_t M, size_t ef_construction,
                        const char *branching_factor, size_t seed, int threads, const char *out_path);
/* same with external labels: row i is added as addPoint(base + i*dim, labels[i]) (labels must be distinct) */
hs_status hs_build_hnsw_labeled(const float *base, const uint64_t *labels, size_t n, size_t dim, int metric, size_t M,
                                size_t ef_construction, const char *branching_factor, size_t seed, int threads,
                                const char *out_path);
/* HierarchicalNSWSlim::convertFromHNSW + saveIndex: hnswalg_slim.h:867-1108, 717-751. */
hs_status hs_convert_slim(const char *hnsw_path, int metric, size_t dim, int threshold_level,
                          float top_degree_percent0, float top_degree_percent, size_t top_degree_M0,
                          size_t low_degree_m0, size_t top_degree_M, size_t low_degree_m, int threads,
                          const char *out_path);
/* The same conversion with the per-list work on HIP device `device` (per-node distances, the by-distance
 * std::sort with libstdc++'s tie order, PruneByHeuristic :836-865, the reverse-edge union :988-1012, the re-prune
 * :1038-1062); histograms, hub thresholds and the final assembly run on `threads` host threads.  The file is byte-identical
 * to hs_convert_slim's.  Shapes outside the device path (degree capacities above 32, a reverse-edge list beyond 2048 ids)
 * are converted on the CPU; *used_gpu (nullable) says which path ran, *kernel_ms (nullable) the device time. */
hs_status hs_convert_slim_gpu(const char *hnsw_path, int metric, size_t dim, int threshold_level,
                              float top_degree_percent0, float top_degree_percent, size_t top_degree_M0,
                              size_t low_degree_m0, size_t top_degree_M, size_t low_degree_m, int device, int threads,
                              const char *out_path, int *used_gpu, double *kernel_ms);

/* The graph HNSW-SlimQ is converted FROM: rabitqlib::hnsw::HierarchicalNSW(num_points, dim, total_bits, M, ef_construction,
 * random_seed, metric) + construct() (third_party/rabitqlib/index/hnsw/hnsw.hpp:427-500, 667-1054), as
 * include/strategy/hnsw_slimq_strategy.h:106-121 drives it with M = 32, ef_construction = 128, seed 100.  Edges come from the RAW
 * rows (hnsw.hpp:381-387) in Eigen's inner-product order; heaps order (distance, id) pairs; mult = 1 / ln M.  Only the edges are
 * built here (the RaBitQ records are hs_convert_slimq's job); the file is hnswlib's saveIndex layout (hnswalg.h:748-779), labels ==
 * row index == internal id.  threads == 1 reproduces the reference's serial construct() edge for edge. */
hs_status hs_build_rabitq_hnsw(const float *base, size_t n, size_t dim, int metric, size_t M, size_t ef_construction,
                               size_t seed, int threads, const char *out_path);
/* HierarchicalNSWSlimQ::convertFromHNSW's graph passes (hnswalg_slimq.h:1546-1762): hs_convert_slim's passes with SlimQ's own
 * PruneByHeuristic as written (:1334-1362 -- the occlusion test reads the row whose id is the LOOP INDEX, :1349) and rabitqlib's
 * raw distance (:1623, :1706).  Writes a Slim-layout file; feed it to hs_convert_slimq for the quantised index. */
hs_status hs_convert_slimq_graph(const char *hnsw_path, int metric, size_t dim, int threshold_level,
                                 float top_degree_percent0, float top_degree_percent, size_t top_degree_M0,
                                 size_t low_degree_m0, size_t top_degree_M, size_t low_degree_m, int threads,
                                 const char *out_path);

/* HierarchicalNSWSlimQ::convertFromHNSW's OUTPUT format + saveIndex (hnswalg_slimq.h:1471-1790, 1161-1216): keeps
 * the graph of an existing HierarchicalNSWSlim file and replaces the fp32 rows by RaBitQ records (cluster id,
 * 1-bit code, {f_add, f_rescale, f_error}); the ex-bits area is zero (no function on the search path reads it).
 * centroids: num_cluster x dim raw vectors; cluster_ids: n ids or NULL (= nearest centroid). */
hs_status hs_convert_slimq(const char *slim_path, int metric, size_t dim, const float *centroids, size_t num_cluster,
                           const uint32_t *cluster_ids, uint64_t flip_seed, int threads, const char *out_path);
/* quant::faster_config(padded, 4).t_const (rabitqlib/quantization/rabitq.hpp:27-33) with a seeded generator. */
double hs_rabitq_default_tconst(size_t padded_dim, uint64_t seed);

/* ---- RaBitQ pieces of the HNSW-SlimQ path (CPU; used by the SlimQ harness and query preparation, exposed
 *      so that tests can pin them against the compiled rabitqlib) ------------------------------------- */
/* FhtKacRotator::rotate: rabitqlib/utils/rotator.hpp:370-423.  flips = 4*padded/8 bytes, out = n x padded. */
hs_status hs_rabitq_rotate(size_t dim, const uint8_t *flips, const float *in, size_t n, float *out);
/* one_bit_compact_code<float,uint64_t>: rabitqlib/quantization/rabitq_impl.hpp:75-187.  codes n x padded/64,
 * factors n x {f_add, f_rescale, f_error}. */
hs_status hs_rabitq_quantize_data(size_t padded, int metric, const float *rotated, size_t n, const float *centroid,
                                  uint64_t *codes, float *factors);
/* SplitSingleQuery ctor: rabitqlib/index/query.hpp:112-156.  out3 n x {delta, vl, k1xsumq}; bins n x padded/64*4. */
hs_status hs_rabitq_prepare_query(size_t padded, double t_const, const float *rotated_q, size_t n, float *out3,
                                  uint64_t *bins);
/* split_single_estdist: rabitqlib/index/estimator.hpp:164-188.  out nq x nd x {ip_x0_qr, est_dist, low_dist}. */
hs_status hs_rabitq_estimate(size_t padded, const uint64_t *codes, const float *factors, size_t nd, const float *q3,
                             const uint64_t *bins, const float *g_add, const float *g_error, size_t nq, float *out);

#ifdef __cplusplus
}
#endif
#endif /* HNSW_SLIM_AMD_H */
