"""Shared helpers for tests/bench: fvecs IO, seeded synthetic data, ctypes loaders.

The oracle loader lives here (tests only); the product binding lives in hnsw-slim_amd/__init__.py.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def write_fvecs(path, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    out = np.empty((n, d + 1), dtype=np.float32)
    out[:, 0] = np.frombuffer(np.int32(d).tobytes(), dtype=np.float32)[0]
    out[:, 1:] = x
    out.tofile(path)


def read_fvecs(path):
    raw = np.fromfile(path, dtype=np.float32)
    d = int(raw[:1].view(np.int32)[0])
    return np.ascontiguousarray(raw.reshape(-1, d + 1)[:, 1:])


def mixture(n, d, seed, n_clusters=16, lo=20.0, hi=120.0, sigma=25.0, integer=False, centres_seed=7):
    """Gaussian-mixture rows (SURVEY.md 8d style). integer=True rounds+clips to [0,255] (tie-heavy)."""
    crng = np.random.default_rng(centres_seed)
    centres = crng.uniform(lo, hi, size=(n_clusters, d)).astype(np.float32)
    rng = np.random.default_rng(seed)
    which = rng.integers(0, n_clusters, size=n)
    x = centres[which] + rng.standard_normal((n, d)).astype(np.float32) * np.float32(sigma)
    if integer:
        x = np.clip(np.rint(x), 0, 255)
    return np.ascontiguousarray(x, dtype=np.float32)


def sift_like(n, d, seed, n_clusters=256, rank=12, sigma_sub=40.0, sigma_iso=4.0, centres_seed=7, integer=True,
              centre_lo=40.0, centre_hi=110.0):
    """SIFT-like rows: a mixture of `n_clusters` LOW-RANK Gaussians (intrinsic dimension `rank`, like real
    SIFT descriptors whose local intrinsic dimension is ~10-16) plus small isotropic noise, rounded and
    clipped to integers in [0,255] stored as fp32.  The isotropic mixture of SURVEY.md 8(d) was measured
    to be un-indexable at N=1M (recall@10 0.63 at ef=256), so the headline data uses this."""
    crng = np.random.default_rng(centres_seed)
    centres = crng.uniform(centre_lo, centre_hi, size=(n_clusters, d)).astype(np.float32)
    bases = np.empty((n_clusters, rank, d), np.float32)
    for c in range(n_clusters):
        qm, _ = np.linalg.qr(crng.standard_normal((d, rank)))
        bases[c] = qm.T.astype(np.float32)
    rng = np.random.default_rng(seed)
    which = rng.integers(0, n_clusters, size=n)
    z = rng.standard_normal((n, rank)).astype(np.float32) * np.float32(sigma_sub)
    x = rng.standard_normal((n, d)).astype(np.float32) * np.float32(sigma_iso)
    order = np.argsort(which, kind="stable")
    bounds = np.searchsorted(which[order], np.arange(n_clusters + 1))
    for c in range(n_clusters):
        idx = order[bounds[c]:bounds[c + 1]]
        if len(idx):
            x[idx] += centres[c] + z[idx] @ bases[c]
    if integer:
        x = np.clip(np.rint(x), 0, 255)
    return np.ascontiguousarray(x, dtype=np.float32)


def headline_data(n, d, seed):
    """The bench's SIFT-1M-like distribution (calibrated on the GPU box, tools/calibrate_data.py:
    HNSW-Slim recall@10 at N=1M: 0.82 / 0.945 / 0.992 / 1.0 at ef 32 / 64 / 128 / 256)."""
    return sift_like(n, d, seed, n_clusters=4096, rank=12, sigma_sub=40.0, sigma_iso=4.0)


def load_chal_encode():
    """oracle/chal_encode.py (test infrastructure: the independent Python writer / reader of the Slim file format)."""
    spec = importlib.util.spec_from_file_location("chal_encode", os.path.join(ROOT, "oracle", "chal_encode.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_product():
    """Import the product package (directory name has a hyphen, so load it by path)."""
    name = "hnsw_slim_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "hnsw-slim_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class Oracle:
    """ctypes view of oracle/liboracle.so (CPU restatement; checker only)."""

    def __init__(self, variant=""):
        """variant "": the pinned parity build; "_ofast": the reference's own -Ofast flags (timing only, bench.py)."""
        so = os.path.join(ROOT, "oracle", f"liboracle{variant}.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), f"liboracle{variant}.so"])
        L = ctypes.CDLL(so)
        L.hso_last_error.restype = ctypes.c_char_p
        L.hso_load.restype = ctypes.c_void_p
        L.hso_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        L.hso_free.argtypes = [ctypes.c_void_p]
        L.hso_set_ef.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.hso_set_filter.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hso_set_mark_ep.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.hso_slim_entry.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.hso_pool_run.restype = ctypes.c_size_t
        L.hso_pool_run.argtypes = [ctypes.c_size_t] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 3
        L.hso_count.restype = ctypes.c_size_t
        L.hso_count.argtypes = [ctypes.c_void_p]
        L.hso_maxlevel.argtypes = [ctypes.c_void_p]
        vp = ctypes.c_void_p
        L.hso_slim_search_ids.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_size_t, vp, vp, vp, vp, ctypes.c_int]
        L.hso_search_pq.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp, ctypes.c_int]
        L.hso_dist.argtypes = [ctypes.c_int, vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
        L.hso_brute_force.argtypes = [ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_int]
        sz, dbl, ci = ctypes.c_size_t, ctypes.c_double, ctypes.c_int
        L.hso_slimq_load.restype = vp
        L.hso_slimq_load.argtypes = [ctypes.c_char_p]
        L.hso_slimq_free.argtypes = [vp]
        L.hso_slimq_set.argtypes = [vp, sz, dbl, vp]
        L.hso_slimq_info.argtypes = [vp, vp]
        L.hso_slimq_perturb.argtypes = [vp, vp]
        L.hso_slimq_search.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, ci]
        L.hso_slimq_prepare.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.hso_slimq_trace.restype = sz
        L.hso_slimq_trace.argtypes = [vp, vp, sz, vp, sz]
        L.hso_rq_rotate.argtypes = [sz, vp, vp, sz, vp]
        L.hso_rq_prepare.argtypes = [sz, ci, dbl, vp, sz, vp, sz, vp, vp, vp]
        L.hso_rq_est.argtypes = [sz, ci, vp, vp, sz, vp, vp, vp, sz, vp]
        self.L = L

    def rq_rotate(self, dim, flips, x):
        x = np.ascontiguousarray(x, np.float32)
        flips = np.ascontiguousarray(flips, np.uint8)
        out = np.empty((x.shape[0], (dim + 63) // 64 * 64), np.float32)
        self.L.hso_rq_rotate(dim, flips.ctypes.data, x.ctypes.data, x.shape[0], out.ctypes.data)
        return out

    def rq_prepare(self, rq, metric, t_const, cent):
        rq = np.ascontiguousarray(rq, np.float32)
        cent = np.ascontiguousarray(cent, np.float32).reshape(-1, rq.shape[1])
        n, padded = rq.shape
        ncl = cent.shape[0]
        q3 = np.empty((n, 3), np.float32)
        planes = np.empty((n, padded // 64 * 4), np.uint64)
        q2c = np.empty((n, ncl * (2 if metric == 1 else 1)), np.float32)
        self.L.hso_rq_prepare(padded, metric, float(t_const), rq.ctypes.data, n, cent.ctypes.data, ncl, q3.ctypes.data,
                              planes.ctypes.data, q2c.ctypes.data)
        return q3, planes, q2c

    def rq_est(self, metric, codes, fac, q3, planes, g_add):
        a = [np.ascontiguousarray(x) for x in (codes, fac, q3, planes, np.asarray(g_add, np.float32))]
        nd, nq = codes.shape[0], q3.shape[0]
        out = np.empty((nq, nd), np.float32)
        self.L.hso_rq_est(codes.shape[1] * 64, metric, a[0].ctypes.data, a[1].ctypes.data, nd, a[2].ctypes.data, a[3].ctypes.data,
                          a[4].ctypes.data, nq, out.ctypes.data)
        return out

    def pool_run(self, cap, op, ids, d):
        """SearchBuffer restatement under an op sequence -> (events, final ids, final dists)."""
        op, ids, d = np.ascontiguousarray(op, np.uint8), np.ascontiguousarray(ids, np.uint32), np.ascontiguousarray(d, np.float32)
        ev = np.empty(len(op), np.uint32)
        oi, od = np.empty(cap + 1, np.uint32), np.empty(cap + 1, np.float32)
        sz = self.L.hso_pool_run(cap, op.ctypes.data, ids.ctypes.data, d.ctypes.data, len(op), ev.ctypes.data, oi.ctypes.data, od.ctypes.data)
        return ev, oi[:sz], od[:sz]

    def load_slimq(self, path):
        h = self.L.hso_slimq_load(path.encode())
        if not h:
            raise RuntimeError(self.err())
        return OracleSlimQ(self, h)

    def err(self):
        return self.L.hso_last_error().decode()

    def dist(self, metric, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        out = np.empty(a.shape[0], np.float32)
        rc = self.L.hso_dist(metric, a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data)
        if rc:
            raise RuntimeError(self.err())
        return out

    def brute_force(self, metric, base, q, k, threads=8):
        base = np.ascontiguousarray(base, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty((q.shape[0], k), np.uint32)
        self.L.hso_brute_force(metric, base.ctypes.data, base.shape[0], base.shape[1], q.ctypes.data, q.shape[0], k, out.ctypes.data, threads)
        return out

    def load(self, path, kind, metric, dim):
        h = self.L.hso_load(path.encode(), {"hnsw": 0, "slim": 1}[kind], metric, dim)
        if not h:
            raise RuntimeError(self.err())
        return OracleIndex(self, h, kind, dim)


class OracleSlimQ:
    """HierarchicalNSWSlimQ restatement (oracle/hs_oracle_slimq.hpp)."""

    def __init__(self, o, h):
        self.o, self.h = o, h
        info = np.zeros(8, np.uint64)
        o.L.hso_slimq_info(h, info.ctypes.data)
        self.count, self.dim, self.padded, self.ncl, self.maxlevel, self.enterpoint, self.metric, self.threshold_level = (int(x) for x in info)
        self._raw = None

    def __del__(self):
        try:
            self.o.L.hso_slimq_free(self.h)
        except Exception:
            pass

    def set(self, ef, t_const, raw):
        self._raw = np.ascontiguousarray(raw, np.float32)
        assert self._raw.shape == (self.count, self.dim)
        self.o.L.hso_slimq_set(self.h, ef, float(t_const), self._raw.ctypes.data)

    def perturb(self, ulps):
        """Sensitivity probe: move {delta, vl, k1xsumq, q_to_centroids} by `ulps` units in the last place (4 ints; zeros = exact)."""
        a = np.ascontiguousarray(ulps, np.int32)
        assert a.shape == (4,)
        self.o.L.hso_slimq_perturb(self.h, a.ctypes.data)

    def prepare(self, q):
        q = np.ascontiguousarray(q, np.float32)
        n = q.shape[0]
        rq = np.empty((n, self.padded), np.float32); q3 = np.empty((n, 3), np.float32)
        pl = np.empty((n, self.padded // 64 * 4), np.uint64); ga = np.empty((n, self.ncl), np.float32)
        self.o.L.hso_slimq_prepare(self.h, q.ctypes.data, n, rq.ctypes.data, q3.ctypes.data, pl.ctypes.data, ga.ctypes.data)
        return dict(rq=rq, q3=q3, g_add=ga, planes=pl)

    def trace(self, q1, k, cap=4096):
        q1 = np.ascontiguousarray(q1, np.float32).reshape(-1)
        out = np.full(cap, 0xFFFFFFFF, np.uint32)
        n = self.o.L.hso_slimq_trace(self.h, q1.ctypes.data, k, out.ctypes.data, cap)
        return out[:min(n, cap)]

    def search(self, q, k, threads=1):
        """searchKnn(q, k, result): labels/dists in the reference's heap-array order, count found, counters
        {hops, estimates, pool inserts, revisits}."""
        q = np.ascontiguousarray(q, np.float32)
        nq = q.shape[0]
        lab = np.empty((nq, k), np.uint64)
        d = np.empty((nq, k), np.float32)
        cnt = np.empty(nq, np.uint32)
        ctr = np.zeros((nq, 4), np.uint64)
        rc = self.o.L.hso_slimq_search(self.h, q.ctypes.data, nq, k, lab.ctypes.data, d.ctypes.data, cnt.ctypes.data, ctr.ctypes.data, threads)
        if rc:
            raise RuntimeError(self.o.err())
        return dict(labels=lab, dists=d, counts=cnt, counters=ctr)


class OracleIndex:
    def __init__(self, o, h, kind, dim):
        self.o, self.h, self.kind, self.dim = o, h, kind, dim
        self.ef = 10

    def __del__(self):
        try:
            self.o.L.hso_free(self.h)
        except Exception:
            pass

    @property
    def count(self):
        return self.o.L.hso_count(self.h)

    def set_ef(self, ef):
        self.ef = ef
        self.o.L.hso_set_ef(self.h, ef)

    def set_mark_ep(self, v):
        """-1: as the overload does; 0 / 1: force (cross-pin knob of the pq overloads, see oracle/hs_oracle.hpp)."""
        self.o.L.hso_set_mark_ep(self.h, int(v))

    def entry(self, q):
        """Level-0 entry node of each query (Slim index)."""
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(q.shape[0], np.uint32)
        self.o.L.hso_slim_entry(self.h, q.ctypes.data, q.shape[0], out.ctypes.data)
        return out

    def set_filter(self, allowed):
        """allowed: uint8[n] by internal id, or None."""
        if allowed is None:
            self.o.L.hso_set_filter(self.h, None)
        else:
            a = np.ascontiguousarray(allowed, np.uint8)
            assert a.shape[0] == self.count
            self.o.L.hso_set_filter(self.h, a.ctypes.data)

    def search_ids(self, q, k, threads=1, raw=True):
        """HierarchicalNSWSlim::searchKnn(q,k,tableint*) -> dict(labels, raw_d, raw_i, raw_sz, counters)."""
        q = np.ascontiguousarray(q, np.float32)
        nq = q.shape[0]
        cap = max(self.ef, k)
        out = np.zeros((nq, k), np.uint32)
        rd = np.zeros((nq, cap), np.float32)
        ri = np.zeros((nq, cap), np.uint32)
        rs = np.zeros(nq, np.uint32)
        cn = np.zeros((nq, 5), np.uint32)
        rc = self.o.L.hso_slim_search_ids(self.h, q.ctypes.data, nq, k, out.ctypes.data, cap, rd.ctypes.data, ri.ctypes.data, rs.ctypes.data, cn.ctypes.data, threads)
        if rc:
            raise RuntimeError(self.o.err())
        return dict(labels=out, raw_d=rd, raw_i=ri, raw_sz=rs, counters=cn)

    def search_pq(self, q, k, threads=1):
        """priority_queue-returning searchKnn (vanilla or slim): pop order, farthest first."""
        q = np.ascontiguousarray(q, np.float32)
        nq = q.shape[0]
        cap = max(self.ef, k)
        od = np.zeros((nq, k), np.float32)
        ol = np.zeros((nq, k), np.uint64)
        oc = np.zeros(nq, np.uint32)
        rd = np.zeros((nq, cap), np.float32)
        ri = np.zeros((nq, cap), np.uint32)
        rs = np.zeros(nq, np.uint32)
        cn = np.zeros((nq, 5), np.uint32)
        rc = self.o.L.hso_search_pq(self.h, q.ctypes.data, nq, k, od.ctypes.data, ol.ctypes.data, oc.ctypes.data, cap, rd.ctypes.data, ri.ctypes.data, rs.ctypes.data, cn.ctypes.data, threads)
        if rc:
            raise RuntimeError(self.o.err())
        return dict(dists=od, labels=ol, cnt=oc, raw_d=rd, raw_i=ri, raw_sz=rs, counters=cn)


def read_ref_search(path):
    """Parse the result file written by oracle/ref_driver.cpp `search`."""
    buf = open(path, "rb").read()
    off = 0
    nq, k, nef = np.frombuffer(buf, np.uint32, 3, off)
    off += 12
    out = {}
    for _ in range(nef):
        ef = int(np.frombuffer(buf, np.uint32, 1, off)[0]); off += 4
        dists = np.zeros((nq, k), np.float32)
        labels = np.zeros((nq, k), np.uint64)
        cnt = np.zeros(nq, np.uint32)
        calls = np.zeros(nq, np.uint32)
        for i in range(nq):
            c, nc = np.frombuffer(buf, np.uint32, 2, off); off += 8
            cnt[i], calls[i] = c, nc
            rec = np.frombuffer(buf, np.dtype([("d", "<f4"), ("l", "<u8")]), c, off); off += 12 * int(c)
            dists[i, :c] = rec["d"]
            labels[i, :c] = rec["l"]
        out[ef] = dict(dists=dists, labels=labels, cnt=cnt, calls=calls)
    return out
