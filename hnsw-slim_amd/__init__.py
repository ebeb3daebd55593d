"""hnsw_slim_amd -- thin ctypes binding over the C ABI (include/hnsw_slim_amd.h) of the MI355X-native
HNSW / HNSW-Slim batched search engine.  Used by tests/ and bench.py; the drop-in surface for C++
callers is hnsw-slim_amd/hnswlib/hnswlib_amd.h.

There is no CPU search path here: if the HIP library is missing or no device is visible, index
loading / searching raises (fails loudly) instead of falling back.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HS_LIB", os.path.join(_HERE, "libhnsw_slim_amd.so"))  # HS_LIB: diagnostic builds only

HS_KIND_HNSW, HS_KIND_SLIM, HS_KIND_SLIMQ = 0, 1, 2
HS_METRIC_L2, HS_METRIC_IP = 0, 1
HS_MODE_SLIM_IDS, HS_MODE_PQ = 0, 1
HS_OK, HS_ERR_IO, HS_ERR_CORRUPT, HS_ERR_NOMEM, HS_ERR_INVALID, HS_ERR_UNSUPPORTED, HS_ERR_DEVICE, HS_ERR_CAPACITY = range(8)

EXPORTS = [
    "hs_last_error", "hs_device_count", "hs_index_load", "hs_index_load_mem", "hs_index_free", "hs_set_ef", "hs_index_info",
    "hs_set_capacity", "hs_set_exact_order", "hs_search_batch", "hs_search_batch_dev", "hs_search_check", "hs_search_batch_raw", "hs_search_batch_filtered", "hs_labels",
    "hs_build_hnsw", "hs_build_hnsw_labeled", "hs_convert_slim", "hs_rabitq_rotate", "hs_rabitq_quantize_data", "hs_rabitq_prepare_query",
    "hs_rabitq_estimate", "hs_convert_slimq", "hs_rabitq_default_tconst", "hs_slimq_set_dataset", "hs_slimq_set_tconst", "hs_slimq_get_tconst",
    "hs_slimq_search_batch", "hs_slimq_search_batch_dev", "hs_slimq_trace", "hs_slimq_prepare_debug", "hs_brute_force", "hs_brute_force_dev",
    "hs_search_batch_async", "hs_host_alloc", "hs_host_free", "hs_comm_init", "hs_comm_free", "hs_comm_size", "hs_search_batch_sharded",
    "hs_comm_results_dev", "hs_convert_slim_gpu", "hs_index_patch", "hs_index_from_host_arrays", "hs_build_rabitq_hnsw",
    "hs_convert_slimq_graph", "hs_host_device_pointer",
]


class HsError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class HsInfo(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("dim", ctypes.c_uint64), ("kind", ctypes.c_int32), ("metric", ctypes.c_int32),
                ("maxlevel", ctypes.c_int32), ("threshold_level", ctypes.c_int32), ("enterpoint", ctypes.c_uint32),
                ("has_deleted", ctypes.c_int32), ("n_edges", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64),
                ("max_degree0", ctypes.c_uint64), ("index_size", ctypes.c_uint64)]


def build_library(force=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "libhnsw_slim_amd.so"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HsError(HS_ERR_DEVICE, f"{LIB_PATH} is missing: build it with `make -C hnsw-slim_amd` "
                      "(__graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, ci, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
    L.hs_last_error.restype = ctypes.c_char_p
    L.hs_device_count.restype = ci
    L.hs_index_load.argtypes = [ctypes.c_char_p, ci, ci, sz, sz, ci, ctypes.POINTER(vp)]
    L.hs_index_load_mem.argtypes = [ctypes.c_char_p, sz, ci, ci, sz, sz, ci, ctypes.POINTER(vp)]
    L.hs_index_patch.argtypes = [vp, ctypes.c_char_p, sz, ci]
    L.hs_index_from_host_arrays.argtypes = [ci, ci, sz, sz, vp, vp, vp, vp, vp, vp, u32, ctypes.c_int32, ctypes.c_int32, ci, ctypes.POINTER(vp)]
    L.hs_index_free.argtypes = [vp]
    L.hs_index_free.restype = None
    L.hs_set_ef.argtypes = [vp, sz]
    L.hs_index_info.argtypes = [vp, ctypes.POINTER(HsInfo)]
    L.hs_last_kernel.argtypes = [vp]
    L.hs_last_kernel.restype = ctypes.c_char_p
    L.hs_set_capacity.argtypes = [vp, u32, u32]
    L.hs_set_exact_order.argtypes = [vp, ci]
    L.hs_search_batch.argtypes = [vp, vp, sz, sz, ci, vp, vp, vp, vp, vp]
    L.hs_search_batch_dev.argtypes = [vp, vp, sz, sz, ci, vp, vp, vp, vp, vp, vp]
    L.hs_search_check.argtypes = [vp, vp]
    L.hs_debug_heap_ops.argtypes = [vp, sz, ci, u32, vp, vp, vp]
    L.hs_debug_flat_plan.argtypes = [sz, sz, sz, vp]
    L.hs_search_batch_async.argtypes = [vp, vp, sz, sz, ci, vp, vp, vp, vp, vp, vp]
    L.hs_host_alloc.restype = vp
    L.hs_host_alloc.argtypes = [sz]
    L.hs_host_free.restype = None
    L.hs_host_free.argtypes = [vp]
    L.hs_comm_init.argtypes = [ci, vp, ctypes.POINTER(vp)]
    L.hs_comm_free.restype = None
    L.hs_comm_free.argtypes = [vp]
    L.hs_comm_size.argtypes = [vp]
    L.hs_search_batch_sharded.argtypes = [vp, vp, vp, sz, sz, ci, vp, vp, vp, vp]
    L.hs_search_batch_sharded_async.argtypes = [vp, vp, vp, sz, sz, ci, vp, vp, vp, vp, ci]
    L.hs_comm_check.argtypes = [vp, vp, ci]
    L.hs_comm_slots.argtypes = [vp]
    L.hs_comm_results_dev.argtypes = [vp, ci, vp, vp, vp, vp]
    L.hs_search_batch_raw.argtypes = [vp, vp, sz, sz, ci, vp, vp, vp, vp]
    L.hs_search_batch_filtered.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hs_labels.argtypes = [vp, vp]
    L.hs_build_hnsw.argtypes = [vp, sz, sz, ci, sz, sz, ctypes.c_char_p, sz, ci, ctypes.c_char_p]
    L.hs_convert_slim.argtypes = [ctypes.c_char_p, ci, sz, ci, ctypes.c_float, ctypes.c_float, sz, sz, sz, sz, ci, ctypes.c_char_p]
    L.hs_convert_slimq_graph.argtypes = L.hs_convert_slim.argtypes
    L.hs_host_device_pointer.restype = ctypes.c_void_p
    L.hs_host_device_pointer.argtypes = [vp]
    L.hs_build_rabitq_hnsw.argtypes = [vp, sz, sz, ci, sz, sz, sz, ci, ctypes.c_char_p]
    L.hs_convert_slim_gpu.argtypes = [ctypes.c_char_p, ci, sz, ci, ctypes.c_float, ctypes.c_float, sz, sz, sz, sz, ci, ci, ctypes.c_char_p,
                                      ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_double)]
    L.hs_convert_slimq.argtypes = [ctypes.c_char_p, ci, sz, vp, sz, vp, ctypes.c_uint64, ci, ctypes.c_char_p]
    L.hs_rabitq_default_tconst.restype = ctypes.c_double
    L.hs_rabitq_default_tconst.argtypes = [sz, ctypes.c_uint64]
    L.hs_slimq_set_dataset.argtypes = [vp, vp, sz, sz]
    L.hs_slimq_set_tconst.argtypes = [vp, ctypes.c_double]
    L.hs_slimq_get_tconst.restype = ctypes.c_double
    L.hs_slimq_get_tconst.argtypes = [vp]
    L.hs_slimq_search_batch.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp]
    L.hs_slimq_search_batch_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hs_slimq_trace.argtypes = [vp, vp, sz, sz, vp, sz, vp]
    L.hs_slimq_prepare_debug.argtypes = [vp, vp, sz, vp]
    L.hs_brute_force.argtypes = [vp, sz, sz, ci, vp, vp, sz, sz, ci, vp, vp, vp]
    L.hs_brute_force_dev.argtypes = [vp, vp, sz, sz, ci, vp, sz, sz, vp, vp, vp, vp]
    L.hs_rabitq_rotate.argtypes = [sz, vp, vp, sz, vp]
    L.hs_rabitq_quantize_data.argtypes = [sz, ci, vp, sz, vp, vp, vp]
    L.hs_rabitq_prepare_query.argtypes = [sz, ctypes.c_double, vp, sz, vp, vp]
    L.hs_rabitq_estimate.argtypes = [sz, vp, vp, sz, vp, vp, vp, vp, sz, vp]
    _lib = L
    return L


def _check(rc):
    if rc != HS_OK:
        raise HsError(rc, lib().hs_last_error().decode())


def device_count():
    return lib().hs_device_count()


def build_hnsw(base, out_path, metric=HS_METRIC_L2, M=16, ef_construction=200, branching_factor="4", seed=100, threads=1):
    """HierarchicalNSW ctor + addPoint loop (labels = row index) + saveIndex, on the CPU (harness)."""
    base = np.ascontiguousarray(base, np.float32)
    _check(lib().hs_build_hnsw(base.ctypes.data, base.shape[0], base.shape[1], metric, M, ef_construction,
                               str(branching_factor).encode(), seed, threads, out_path.encode()))


def convert_slim(hnsw_path, out_path, dim, metric=HS_METRIC_L2, threshold_level=0, top_degree_percent0=0.02,
                 top_degree_percent=0.02, top_degree_M0=32, low_degree_m0=8, top_degree_M=16, low_degree_m=4, threads=1):
    """HierarchicalNSWSlim::convertFromHNSW + saveIndex, on the CPU (harness)."""
    _check(lib().hs_convert_slim(hnsw_path.encode(), metric, dim, threshold_level, top_degree_percent0, top_degree_percent,
                                 top_degree_M0, low_degree_m0, top_degree_M, low_degree_m, threads, out_path.encode()))


def build_rabitq_hnsw(base, out_path, metric=HS_METRIC_L2, M=32, ef_construction=128, seed=100, threads=1):
    """rabitqlib::hnsw::HierarchicalNSW::construct's edges (the graph HNSW-SlimQ converts from; defaults =
    hnsw_slimq_strategy.h:106-108), saved in hnswlib's layout.  CPU harness."""
    base = np.ascontiguousarray(base, np.float32)
    _check(lib().hs_build_rabitq_hnsw(base.ctypes.data, base.shape[0], base.shape[1], metric, M, ef_construction, seed, threads,
                                      out_path.encode()))


def convert_slimq_graph(hnsw_path, out_path, dim, metric=HS_METRIC_L2, threshold_level=0, top_degree_percent0=0.02,
                        top_degree_percent=0.02, top_degree_M0=32, low_degree_m0=8, top_degree_M=16, low_degree_m=4, threads=1):
    """HierarchicalNSWSlimQ::convertFromHNSW's graph passes (its own PruneByHeuristic) -> Slim-layout file for convert_slimq."""
    _check(lib().hs_convert_slimq_graph(hnsw_path.encode(), metric, dim, threshold_level, top_degree_percent0, top_degree_percent,
                                        top_degree_M0, low_degree_m0, top_degree_M, low_degree_m, threads, out_path.encode()))


def convert_slim_gpu(hnsw_path, out_path, dim, metric=HS_METRIC_L2, threshold_level=0, top_degree_percent0=0.02,
                     top_degree_percent=0.02, top_degree_M0=32, low_degree_m0=8, top_degree_M=16, low_degree_m=4, device=0, threads=8):
    """convertFromHNSW with the per-list work on the GPU; returns (used_gpu, kernel_ms).  Same bytes as convert_slim."""
    used, ms = ctypes.c_int(0), ctypes.c_double(0.0)
    _check(lib().hs_convert_slim_gpu(hnsw_path.encode(), metric, dim, threshold_level, top_degree_percent0, top_degree_percent,
                                     top_degree_M0, low_degree_m0, top_degree_M, low_degree_m, device, threads, out_path.encode(),
                                     ctypes.byref(used), ctypes.byref(ms)))
    return bool(used.value), ms.value


def debug_flat_plan(n, ef, nq):
    out = np.zeros(5, np.uint32)
    _check(lib().hs_debug_flat_plan(n, ef, nq, out.ctypes.data))
    return dict(nb=int(out[0]), mul=int(out[1]), sh=int(out[2]), bits=int(out[3]), ok=bool(out[4]))


def debug_heap_ops(ops, wave_pop=True, lds_slots=1024):
    """hs_debug_heap_ops: ops = [(0, dist, id) | (1, 0, 0), ...] -> (heap [(dist, id)], pops [(dist, id)]) from the device."""
    n = len(ops)
    arr = np.zeros((max(n, 1), 3), np.uint32)
    for i, (kind, d, idv) in enumerate(ops):
        arr[i] = (kind, np.float32(d).view(np.uint32), idv)
    heap = np.zeros((n + 2, 2), np.uint32)
    pops = np.zeros((n + 2, 2), np.uint32)
    cnt = np.zeros(2, np.uint32)
    _check(lib().hs_debug_heap_ops(arr.ctypes.data, n, 1 if wave_pop else 0, lds_slots, heap.ctypes.data, pops.ctypes.data, cnt.ctypes.data))
    f = lambda a, m: [(float(a[i, 0:1].view(np.float32)[0]), int(a[i, 1])) for i in range(m)]
    return f(heap, int(cnt[0])), f(pops, int(cnt[1]))


def brute_force(base, queries, k, metric=HS_METRIC_L2, labels=None, device=0):
    """hnswlib::BruteforceSearch::searchKnn for a batch: (labels, dists) nq x k, ascending by (dist, label)."""
    b = np.ascontiguousarray(base, np.float32)
    q = np.ascontiguousarray(queries, np.float32)
    lab = None if labels is None else np.ascontiguousarray(labels, np.uint64)
    ol = np.empty((q.shape[0], k), np.uint64)
    od = np.empty((q.shape[0], k), np.float32)
    oc = np.empty(q.shape[0], np.uint32)
    _check(lib().hs_brute_force(b.ctypes.data, b.shape[0], b.shape[1], metric, None if lab is None else lab.ctypes.data, q.ctypes.data,
                                q.shape[0], k, device, ol.ctypes.data, od.ctypes.data, oc.ctypes.data))
    return ol, od, oc


def brute_force_dev(d_base, d_queries, k, d_labels_out, d_dists_out, metric=HS_METRIC_L2, d_counts=None, stream=0):
    """Device tensors (torch): base n x d, queries nq x d, outputs int64 / float32 nq x k."""
    _check(lib().hs_brute_force_dev(d_base.data_ptr(), None, d_base.shape[0], d_base.shape[1], metric, d_queries.data_ptr(),
                                    d_queries.shape[0], k, d_labels_out.data_ptr(), d_dists_out.data_ptr(),
                                    d_counts.data_ptr() if d_counts is not None else None, stream))


def convert_slimq(slim_path, metric, dim, centroids, out_path, cluster_ids=None, flip_seed=1, threads=8):
    c = np.ascontiguousarray(centroids, np.float32).reshape(-1, dim)
    cid = None if cluster_ids is None else np.ascontiguousarray(cluster_ids, np.uint32)
    _check(lib().hs_convert_slimq(slim_path.encode(), metric, dim, c.ctypes.data, c.shape[0], None if cid is None else cid.ctypes.data,
                                  flip_seed, threads, out_path.encode()))


def rabitq_default_tconst(padded_dim, seed=1):
    return lib().hs_rabitq_default_tconst(padded_dim, seed)


def rabitq_rotate(dim, flips, x):
    x = np.ascontiguousarray(x, np.float32)
    padded = (dim + 63) // 64 * 64
    flips = np.ascontiguousarray(flips, np.uint8)
    out = np.empty((x.shape[0], padded), np.float32)
    _check(lib().hs_rabitq_rotate(dim, flips.ctypes.data, x.ctypes.data, x.shape[0], out.ctypes.data))
    return out


def rabitq_quantize_data(rotated, centroid, metric=HS_METRIC_L2):
    r = np.ascontiguousarray(rotated, np.float32)
    c = np.ascontiguousarray(centroid, np.float32)
    n, padded = r.shape
    codes = np.empty((n, padded // 64), np.uint64)
    fac = np.empty((n, 3), np.float32)
    _check(lib().hs_rabitq_quantize_data(padded, metric, r.ctypes.data, n, c.ctypes.data, codes.ctypes.data, fac.ctypes.data))
    return codes, fac


def rabitq_prepare_query(rotated_q, t_const):
    r = np.ascontiguousarray(rotated_q, np.float32)
    n, padded = r.shape
    q3 = np.empty((n, 3), np.float32)
    bins = np.empty((n, padded // 64 * 4), np.uint64)
    _check(lib().hs_rabitq_prepare_query(padded, float(t_const), r.ctypes.data, n, q3.ctypes.data, bins.ctypes.data))
    return q3, bins


def rabitq_estimate(codes, fac, q3, bins, g_add, g_error):
    nd, nblk = codes.shape
    nq = q3.shape[0]
    out = np.empty((nq, nd, 3), np.float32)
    a = [np.ascontiguousarray(x) for x in (codes, fac, q3, bins, np.asarray(g_add, np.float32), np.asarray(g_error, np.float32))]
    _check(lib().hs_rabitq_estimate(nblk * 64, a[0].ctypes.data, a[1].ctypes.data, nd, a[2].ctypes.data, a[3].ctypes.data,
                                    a[4].ctypes.data, a[5].ctypes.data, nq, out.ctypes.data))
    return out


class PinnedArray:
    """A numpy view of page-locked host memory (hs_host_alloc), for the asynchronous host-pointer entry."""

    def __init__(self, shape, dtype):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().hs_host_alloc(self.nbytes)
        if not self.ptr:
            raise HsError(HS_ERR_NOMEM, "hs_host_alloc failed")
        buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def device_view(self, rows=None):
        """An object search_ids_dev accepts as d_queries: the device's address of this buffer (hs_host_device_pointer) -- the kernels
        then read the queries in place instead of from a staged copy.  None when the buffer is not device-mapped."""
        p = lib().hs_host_device_pointer(self.ptr)
        if not p:
            return None
        shape = self.a.shape if rows is None else (rows,) + tuple(self.a.shape[1:])

        class _View:
            def __init__(v):
                v.shape = shape

            def data_ptr(v):
                return p
        return _View()

    def __del__(self):
        try:
            self.a = None
            lib().hs_host_free(self.ptr)
        except Exception:
            pass


class Comm:
    """hs_comm: n devices of one process (the same device listed twice = the one-GPU rehearsal mode)."""

    def __init__(self, devices):
        self._h = ctypes.c_void_p()
        dv = (ctypes.c_int * len(devices))(*devices)
        _check(lib().hs_comm_init(len(devices), dv, ctypes.byref(self._h)))
        self.devices = list(devices)

    def close(self):
        if self._h:
            lib().hs_comm_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_ids(self, replicas, queries, k, want_dists=False):
        """HS_MODE_SLIM_IDS over the replicas (Index objects, one per device of the communicator)."""
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        hs_ = (ctypes.c_void_p * len(replicas))(*[r._h for r in replicas])
        labels = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32) if want_dists else None
        cnt = np.empty(nq, np.uint32)
        _check(lib().hs_search_batch_sharded(self._h, hs_, q.ctypes.data, nq, k, HS_MODE_SLIM_IDS, labels.ctypes.data, None,
                                             dists.ctypes.data if want_dists else None, cnt.ctypes.data))
        return dict(labels=labels, dists=dists, cnt=cnt)

    def slots(self):
        return lib().hs_comm_slots(self._h)

    def search_ids_async(self, replicas, q_pinned, k, labels_pinned, slot):
        """hs_search_batch_sharded_async, HS_MODE_SLIM_IDS: numpy views of page-locked memory; pair with check(replicas, slot)."""
        hs_ = (ctypes.c_void_p * len(replicas))(*[r._h for r in replicas])
        _check(lib().hs_search_batch_sharded_async(self._h, hs_, q_pinned.ctypes.data, q_pinned.shape[0], k, HS_MODE_SLIM_IDS,
                                                   labels_pinned.ctypes.data, None, None, None, slot))

    def check(self, replicas, slot):
        hs_ = (ctypes.c_void_p * len(replicas))(*[r._h for r in replicas])
        _check(lib().hs_comm_check(self._h, hs_, slot))

    def search_pq(self, replicas, queries, k):
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        hs_ = (ctypes.c_void_p * len(replicas))(*[r._h for r in replicas])
        labels = np.empty((nq, k), np.uint64)
        dists = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        _check(lib().hs_search_batch_sharded(self._h, hs_, q.ctypes.data, nq, k, HS_MODE_PQ, None, labels.ctypes.data, dists.ctypes.data,
                                             cnt.ctypes.data))
        return dict(labels=labels, dists=dists, cnt=cnt)


class Index:
    """A device-resident index (mirror of hnswlib::HierarchicalNSW / HierarchicalNSWSlim for search)."""

    def __init__(self, path, kind, dim, metric=HS_METRIC_L2, max_elements=0, device=0):
        self._h = ctypes.c_void_p()
        self.kind, self.dim, self.metric, self.device = kind, dim, metric, device
        self.ef = 10
        if isinstance(path, (bytes, bytearray, memoryview)):   # the serialized index itself (hs_index_load_mem)
            buf = bytes(path)
            _check(lib().hs_index_load_mem(buf, len(buf), kind, metric, dim, max_elements, device, ctypes.byref(self._h)))
        else:
            _check(lib().hs_index_load(path.encode(), kind, metric, dim, max_elements, device, ctypes.byref(self._h)))

    @classmethod
    def from_arrays(cls, kind, metric, vectors, levels, lists, enterpoint, maxlevel, labels=None, deleted=None, threshold_level=0, device=0):
        """hs_index_from_host_arrays: lists[i][l] = neighbour ids of node i at level l (l = 0..levels[i])."""
        v = np.ascontiguousarray(vectors, np.float32)
        n, dim = v.shape
        lv = np.ascontiguousarray(levels, np.int32)
        flat = [np.asarray(l, np.uint32) for node in lists for l in node]
        ptr = np.zeros(len(flat) + 1, np.uint64)
        ptr[1:] = np.cumsum([len(x) for x in flat])
        ids = np.ascontiguousarray(np.concatenate(flat) if flat else np.zeros(0, np.uint32), np.uint32)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint64)
        dl = None if deleted is None else np.ascontiguousarray(deleted, np.uint8)
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.kind, self.dim, self.metric, self.device, self.ef = kind, dim, metric, device, 10
        _check(lib().hs_index_from_host_arrays(kind, metric, n, dim, v.ctypes.data, None if lab is None else lab.ctypes.data,
                                               None if dl is None else dl.ctypes.data, lv.ctypes.data, ptr.ctypes.data, ids.ctypes.data,
                                               int(enterpoint), int(maxlevel), int(threshold_level), device, ctypes.byref(self._h)))
        return self

    @classmethod
    def from_csr(cls, kind, metric, vectors, levels, list_ptr, list_ids, enterpoint, maxlevel, labels=None, threshold_level=0, device=0):
        """hs_index_from_host_arrays with the flat arrays as the C ABI takes them (node i owns levels[i] + 1 consecutive lists)."""
        v = np.ascontiguousarray(vectors, np.float32)
        n, dim = v.shape
        lv, ptr, ids = np.ascontiguousarray(levels, np.int32), np.ascontiguousarray(list_ptr, np.uint64), np.ascontiguousarray(list_ids, np.uint32)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint64)
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        self.kind, self.dim, self.metric, self.device, self.ef = kind, dim, metric, device, 10
        _check(lib().hs_index_from_host_arrays(kind, metric, n, dim, v.ctypes.data, None if lab is None else lab.ctypes.data, None,
                                               lv.ctypes.data, ptr.ctypes.data, ids.ctypes.data, int(enterpoint), int(maxlevel),
                                               int(threshold_level), device, ctypes.byref(self._h)))
        return self

    def close(self):
        if self._h:
            lib().hs_index_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def patch(self, stream_bytes, to_add=False):
        """patchFromStream: apply a genPatch stream to this device-resident Slim index (loaded with max_elements > count)."""
        b = bytes(stream_bytes)
        _check(lib().hs_index_patch(self._h, b, len(b), 1 if to_add else 0))

    def info(self):
        i = HsInfo()
        _check(lib().hs_index_info(self._h, ctypes.byref(i)))
        return {f[0]: getattr(i, f[0]) for f in HsInfo._fields_}

    def set_ef(self, ef):
        self.ef = int(ef)
        _check(lib().hs_set_ef(self._h, int(ef)))

    def set_exact_order(self, on=True):
        """True: strict kernel for every query (reference array order); False: fast kernel, sorted output."""
        _check(lib().hs_set_exact_order(self._h, 1 if on else 0))

    def set_capacity(self, cand_cap=0, hash_slots=0):
        _check(lib().hs_set_capacity(self._h, cand_cap, hash_slots))

    # -- host-pointer API -------------------------------------------------------------------------
    def search_ids(self, queries, k, want_dists=False, want_stats=False):
        """HierarchicalNSWSlim::searchKnn(q, k, tableint*) for every row of `queries`."""
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        labels = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32) if want_dists else None
        cnt = np.empty(nq, np.uint32)
        stats = np.empty((nq, 4), np.uint32) if want_stats else None
        _check(lib().hs_search_batch(self._h, q.ctypes.data, nq, k, HS_MODE_SLIM_IDS, labels.ctypes.data, None,
                                     dists.ctypes.data if want_dists else None, cnt.ctypes.data,
                                     stats.ctypes.data if want_stats else None))
        return dict(labels=labels, dists=dists, cnt=cnt, stats=stats)

    def search_pq(self, queries, k, want_stats=False):
        """priority_queue-returning searchKnn overloads: the <=k (dist,label) pairs per query."""
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        labels = np.empty((nq, k), np.uint64)
        dists = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        stats = np.empty((nq, 4), np.uint32) if want_stats else None
        _check(lib().hs_search_batch(self._h, q.ctypes.data, nq, k, HS_MODE_PQ, None, labels.ctypes.data, dists.ctypes.data,
                                     cnt.ctypes.data, stats.ctypes.data if want_stats else None))
        return dict(labels=labels, dists=dists, cnt=cnt, stats=stats)

    # ---- HNSW-SlimQ (kind == HS_KIND_SLIMQ) ----
    def slimq_set_dataset(self, base):
        """HierarchicalNSWSlimQ::setDataset: raw rows by internal id, used for the exact re-rank."""
        b = np.ascontiguousarray(base, np.float32)
        _check(lib().hs_slimq_set_dataset(self._h, b.ctypes.data, b.shape[0], b.shape[1]))

    def slimq_set_tconst(self, t_const):
        _check(lib().hs_slimq_set_tconst(self._h, float(t_const)))

    def slimq_tconst(self):
        return lib().hs_slimq_get_tconst(self._h)

    def slimq_search(self, queries, k, want_stats=False):
        """searchKnn(q, k, result) of HierarchicalNSWSlimQ: labels/dists in the reference's heap-array order."""
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        labels = np.empty((nq, k), np.uint64)
        dists = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        stats = np.empty((nq, 4), np.uint32) if want_stats else None
        _check(lib().hs_slimq_search_batch(self._h, q.ctypes.data, nq, k, labels.ctypes.data, dists.ctypes.data, cnt.ctypes.data,
                                           stats.ctypes.data if want_stats else None))
        return dict(labels=labels, dists=dists, cnt=cnt, stats=stats)

    def labels(self):
        out = np.empty(self.info()["n"], np.uint64)
        _check(lib().hs_labels(self._h, out.ctypes.data))
        return out

    def search_filtered(self, queries, k, allowed, want_stats=False):
        """searchKnn(q, k, isIdAllowed): allowed[i] != 0 iff the filter accepts the label of internal id i."""
        q = np.ascontiguousarray(queries, np.float32)
        a = np.ascontiguousarray(allowed, np.uint8)
        nq = q.shape[0]
        labels = np.empty((nq, k), np.uint64)
        dists = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        stats = np.empty((nq, 4), np.uint32) if want_stats else None
        _check(lib().hs_search_batch_filtered(self._h, q.ctypes.data, nq, k, a.ctypes.data, labels.ctypes.data, dists.ctypes.data,
                                              cnt.ctypes.data, stats.ctypes.data if want_stats else None))
        return dict(labels=labels, dists=dists, cnt=cnt, stats=stats)

    def search_raw(self, queries, k, mode=HS_MODE_SLIM_IDS):
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        cap = max(self.ef, k)
        rd = np.empty((nq, cap), np.float32)
        ri = np.empty((nq, cap), np.uint32)
        rs = np.empty(nq, np.uint32)
        stats = np.empty((nq, 4), np.uint32)
        _check(lib().hs_search_batch_raw(self._h, q.ctypes.data, nq, k, mode, rd.ctypes.data, ri.ctypes.data, rs.ctypes.data,
                                         stats.ctypes.data))
        return dict(raw_d=rd, raw_i=ri, raw_sz=rs, stats=stats)

    def search_ids_async(self, q_pinned, k, labels_pinned, stream=0, dists_pinned=None, counts_pinned=None):
        """hs_search_batch_async, HS_MODE_SLIM_IDS: numpy views of page-locked memory (PinnedArray.a), a HIP stream handle."""
        _check(lib().hs_search_batch_async(self._h, q_pinned.ctypes.data, q_pinned.shape[0], k, HS_MODE_SLIM_IDS, labels_pinned.ctypes.data, None,
                                           dists_pinned.ctypes.data if dists_pinned is not None else None,
                                           counts_pinned.ctypes.data if counts_pinned is not None else None, None, stream))

    # -- device-pointer API (torch tensors on this index's device; async on `stream`) ----------------
    def search_ids_dev(self, d_queries, k, d_labels, d_dists=None, d_counts=None, d_stats=None, stream=0):
        nq = d_queries.shape[0]
        _check(lib().hs_search_batch_dev(self._h, d_queries.data_ptr(), nq, k, HS_MODE_SLIM_IDS, d_labels.data_ptr(), None,
                                         d_dists.data_ptr() if d_dists is not None else None,
                                         d_counts.data_ptr() if d_counts is not None else None,
                                         d_stats.data_ptr() if d_stats is not None else None, stream))

    def slimq_prepare_debug(self, queries, padded, ncl):
        """dict(rq, q3, g_add, planes) as the kernel computed them."""
        q = np.ascontiguousarray(queries, np.float32)
        row = padded + 3 + ncl + padded // 8
        out = np.empty((q.shape[0], row), np.float32)
        _check(lib().hs_slimq_prepare_debug(self._h, q.ctypes.data, q.shape[0], out.ctypes.data))
        pl = np.ascontiguousarray(out[:, padded + 3 + ncl:]).view(np.uint64)
        return dict(rq=out[:, :padded].copy(), q3=out[:, padded:padded + 3].copy(), g_add=out[:, padded + 3:padded + 3 + ncl].copy(), planes=pl)

    def slimq_trace(self, queries, k, cap=4096):
        q = np.ascontiguousarray(queries, np.float32)
        tr = np.empty((q.shape[0], cap), np.uint32)
        st = np.empty((q.shape[0], 4), np.uint32)
        _check(lib().hs_slimq_trace(self._h, q.ctypes.data, q.shape[0], k, tr.ctypes.data, cap, st.ctypes.data))
        return tr, st

    def slimq_search_dev(self, d_queries, k, d_labels, d_dists, d_counts, d_stats=None, stream=0):
        """Device tensors (labels int64 nq x k, dists f32 nq x k, counts int32 nq) + HIP stream; asynchronous."""
        _check(lib().hs_slimq_search_batch_dev(self._h, d_queries.data_ptr(), d_queries.shape[0], k, d_labels.data_ptr(),
                                               d_dists.data_ptr(), d_counts.data_ptr(),
                                               d_stats.data_ptr() if d_stats is not None else None, stream))

    def last_kernel(self):
        return lib().hs_last_kernel(self._h).decode()

    def check(self, stream=0):
        _check(lib().hs_search_check(self._h, stream))
