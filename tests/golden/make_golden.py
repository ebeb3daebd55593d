#!/usr/bin/env python3
"""Generate tests/golden/ fixtures by RUNNING THE COMPILED REFERENCE (oracle/_ref/ref_hnsw, built by
oracle/Makefile from /root/reference/third_party/hnswlib as-is).  Run in the build container only;
the outputs (data, not source) are committed so the GPU box never needs /root/reference.

    python tests/golden/make_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsutil import GOLDEN, ROOT, mixture, read_ref_search, write_fvecs  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_hnsw")


def run(*args):
    subprocess.check_call([REF, *map(str, args)])


def dist_fixture(tmp):
    out = {}
    rng = np.random.default_rng(20250101)
    for metric, dims in (("l2", [3, 7, 16, 20, 23, 96, 100, 128, 768, 960]), ("ip", [16, 96, 128, 768, 960])):
        for d in dims:
            n = 64 if d <= 128 else 16
            a = (rng.standard_normal((n, d)) * 3).astype(np.float32)
            b = (rng.standard_normal((n, d)) * 3).astype(np.float32)
            if d == 128:  # SIFT-like integer rows too
                a[: n // 2] = np.clip(np.rint(a[: n // 2] * 30 + 100), 0, 255)
                b[: n // 2] = np.clip(np.rint(b[: n // 2] * 30 + 100), 0, 255)
            fa, fb, fo = (os.path.join(tmp, f"{x}.bin") for x in "abo")
            write_fvecs(fa, a)
            write_fvecs(fb, b)
            run("dist", metric, fa, fb, fo)
            out[f"{metric}_{d}_a"] = a
            out[f"{metric}_{d}_b"] = b
            out[f"{metric}_{d}_ref"] = np.fromfile(fo, np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "dist_ref.npz"), **out)


def ip_odd_dims_fixture(tmp):
    """InnerProductSpace off the SIMD16 path (space_ip.h:374-382): scalar (1-3), SIMD4ExtAVX (4, 20, 100), SIMD4 residuals
    (7, 13), SIMD16 residuals (23, 70, 133).  Kept in its own file so that dist_ref.npz stays byte-stable."""
    out = {}
    rng = np.random.default_rng(20250202)
    for d in (1, 3, 4, 7, 13, 20, 23, 70, 100, 133):
        a = (rng.standard_normal((64, d)) * 3).astype(np.float32)
        b = (rng.standard_normal((64, d)) * 3).astype(np.float32)
        fa, fb, fo = (os.path.join(tmp, f"{x}.bin") for x in "abo")
        write_fvecs(fa, a)
        write_fvecs(fb, b)
        run("dist", "ip", fa, fb, fo)
        out[f"ip_{d}_a"] = a
        out[f"ip_{d}_b"] = b
        out[f"ip_{d}_ref"] = np.fromfile(fo, np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "dist_ref_ip_odd.npz"), **out)


def index_fixture(tmp, name, metric, base, queries, M, efC, efs, k=10):
    fb, fq = os.path.join(tmp, "b.fvecs"), os.path.join(tmp, "q.fvecs")
    write_fvecs(fb, base)
    write_fvecs(fq, queries)
    idx = os.path.join(GOLDEN, f"{name}.hnsw.bin")
    run("build", metric, fb, idx, M, efC, "4", 100)
    res = os.path.join(tmp, "res.bin")
    run("search", metric, idx, fq, res, k, *efs)
    parsed = read_ref_search(res)
    out = {"base": base, "queries": queries, "efs": np.array(efs), "k": np.array(k), "M": np.array(M), "efC": np.array(efC)}
    for ef, r in parsed.items():
        for key, v in r.items():
            out[f"ef{ef}_{key}"] = v
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **out)


def deleted_fixture(tmp, src_name, metric, dim, every, efs, k=10):
    """Same graph with delete marks (reference markDelete + saveIndex) and the reference's search results."""
    src = os.path.join(GOLDEN, f"{src_name}.hnsw.bin")
    dst = os.path.join(GOLDEN, f"{src_name}_del.hnsw.bin")
    run("markdel", metric, dim, src, dst, every)
    g = np.load(os.path.join(GOLDEN, f"{src_name}.npz"))
    fq = os.path.join(tmp, "qd.fvecs")
    write_fvecs(fq, g["queries"])
    res = os.path.join(tmp, "resd.bin")
    run("search", metric, dst, fq, res, k, *efs)
    out = {"queries": g["queries"], "efs": np.array(efs), "k": np.array(k), "every": np.array(every)}
    for ef, r in read_ref_search(res).items():
        for key, v in r.items():
            out[f"ef{ef}_{key}"] = v
    np.savez_compressed(os.path.join(GOLDEN, f"{src_name}_del.npz"), **out)


def filter_fixture(tmp, src_name, metric, mod, rem, efs, k=10):
    """searchKnn(q, k, isIdAllowed) of the reference with the filter label % mod != rem."""
    src = os.path.join(GOLDEN, f"{src_name}.hnsw.bin")
    g = np.load(os.path.join(GOLDEN, f"{src_name.replace('_del', '')}.npz"))
    fq = os.path.join(tmp, "qf.fvecs")
    write_fvecs(fq, g["queries"])
    res = os.path.join(tmp, "resf.bin")
    run("searchf", mod, rem, metric, src, fq, res, k, *efs)
    out = {"queries": g["queries"], "efs": np.array(efs), "k": np.array(k), "mod": np.array(mod), "rem": np.array(rem)}
    for ef, r in read_ref_search(res).items():
        for key, v in r.items():
            out[f"ef{ef}_{key}"] = v
    np.savez_compressed(os.path.join(GOLDEN, f"{src_name}_filter.npz"), **out)


def rabitq_fixture(tmp):
    """Outputs of the compiled rabitqlib (oracle/_ref/ref_rabitq) for the separable SlimQ pieces."""
    RQ = os.path.join(ROOT, "oracle", "_ref", "ref_rabitq")
    out = {}
    rng = np.random.default_rng(777)
    for dim, metric in ((128, 0), (96, 0), (768, 1)):
        padded = (dim + 63) // 64 * 64
        nd, nq = 48, 6
        x = (rng.standard_normal((nd, dim)) * 2 + 0.3).astype(np.float32)
        q = (rng.standard_normal((nq, dim)) * 2 + 0.3).astype(np.float32)
        cen = (rng.standard_normal((1, dim)) * 0.5).astype(np.float32)
        f = {k: os.path.join(tmp, f"rq_{k}.bin") for k in ("flip", "x", "q", "c", "rx", "rq", "rc", "codes", "fac", "ga", "ge", "oq", "ob", "oe")}
        if os.path.exists(f["flip"]):
            os.remove(f["flip"])
        x.tofile(f["x"]); q.tofile(f["q"]); cen.tofile(f["c"])
        subprocess.check_call([RQ, "rotate", str(dim), f["flip"], f["x"], str(nd), f["rx"]])
        subprocess.check_call([RQ, "rotate", str(dim), f["flip"], f["q"], str(nq), f["rq"]])
        subprocess.check_call([RQ, "rotate", str(dim), f["flip"], f["c"], "1", f["rc"]])
        subprocess.check_call([RQ, "data", str(padded), str(metric), f["rx"], str(nd), f["rc"], f["codes"], f["fac"]])
        ga = rng.uniform(1, 50, nq).astype(np.float32); ge = rng.uniform(1, 7, nq).astype(np.float32)
        ga.tofile(f["ga"]); ge.tofile(f["ge"])
        t_const = 41.25 if padded == 128 else 90.5
        subprocess.check_call([RQ, "query", str(padded), str(metric), repr(t_const), f["rq"], str(nq), f["codes"], f["fac"], str(nd),
                               f["ga"], f["ge"], f["oq"], f["ob"], f["oe"]])
        p = f"d{dim}_"
        out.update({p + "metric": np.array(metric), p + "t_const": np.array(t_const), p + "x": x, p + "q": q, p + "cen": cen,
                    p + "flip": np.fromfile(f["flip"], np.uint8), p + "rx": np.fromfile(f["rx"], np.float32).reshape(nd, padded),
                    p + "rq": np.fromfile(f["rq"], np.float32).reshape(nq, padded), p + "rc": np.fromfile(f["rc"], np.float32),
                    p + "codes": np.fromfile(f["codes"], np.uint64).reshape(nd, padded // 64),
                    p + "fac": np.fromfile(f["fac"], np.float32).reshape(nd, 3), p + "g_add": ga, p + "g_err": ge,
                    p + "q3": np.fromfile(f["oq"], np.float32).reshape(nq, 3),
                    p + "bins": np.fromfile(f["ob"], np.uint64).reshape(nq, padded // 64 * 4),
                    p + "est": np.fromfile(f["oe"], np.float32).reshape(nq, nd, 3)})
    np.savez_compressed(os.path.join(GOLDEN, "rabitq_ref.npz"), **out)


def rabitq_cent_fixture(tmp):
    """q_to_centroids as the compiled rabitqlib computes it (hnswalg_slimq.h:1823-1848 -> rabitqlib::euclidean_sqr / dot_product,
    Eigen reductions): the rotated queries of rabitq_ref.npz against 16 random rows per shape."""
    RQ = os.path.join(ROOT, "oracle", "_ref", "ref_rabitq")
    ref = np.load(os.path.join(GOLDEN, "rabitq_ref.npz"))
    rng = np.random.default_rng(4242)
    out = {}
    for dim in (128, 96, 768):
        rq = ref[f"d{dim}_rq"]
        nq, padded = rq.shape
        cen = (rng.standard_normal((16, padded)) * 0.7).astype(np.float32)
        fq, fc, fo = (os.path.join(tmp, f"ct_{k}.bin") for k in ("q", "c", "o"))
        rq.tofile(fq); cen.tofile(fc)
        subprocess.check_call([RQ, "cent", str(padded), fq, str(nq), fc, "16", fo])
        o = np.fromfile(fo, np.float32).reshape(nq, 16, 2)
        out.update({f"d{dim}_cen": cen, f"d{dim}_l2sqr": o[:, :, 0].copy(), f"d{dim}_ip": o[:, :, 1].copy()})
    np.savez_compressed(os.path.join(GOLDEN, "rabitq_cent_ref.npz"), **out)


RQ_HNSW_CASES = [   # (name, n, dim, metric, M, efC, integer-valued)
    ("l2_d64", 1500, 64, 0, 8, 40, False), ("l2_d100", 1200, 100, 0, 6, 30, False), ("ip_d128", 1500, 128, 1, 8, 40, False),
    ("l2_int_d64", 1500, 64, 0, 8, 40, True), ("ip_d72", 800, 72, 1, 4, 20, False), ("l2_d112_m32", 900, 112, 0, 32, 128, False),
]


def rq_hnsw_base(n, d, metric, integer):
    if integer:
        return np.ascontiguousarray(mixture(n, d, 3, lo=2, hi=8, sigma=2.0, integer=True), np.float32)
    b = mixture(n, d, 5, lo=-1, hi=1, sigma=0.5)
    if metric:
        b /= np.linalg.norm(b, axis=1, keepdims=True)
    return np.ascontiguousarray(b, np.float32)


def rq_hnsw_fixture(tmp):
    """The graph HNSW-SlimQ converts from, as the compiled rabitqlib builds it serially (hnsw.hpp:667-1054; seed 100):
    [maxlevel, enterpoint, then per node: label, level, per level count + ids].  The rows are regenerated from the seeds."""
    RQ = os.path.join(ROOT, "oracle", "_ref", "ref_rabitq")
    out = {}
    for name, n, d, metric, M, efc, integer in RQ_HNSW_CASES:
        b = rq_hnsw_base(n, d, metric, integer)
        fb, fo = os.path.join(tmp, "rqb.bin"), os.path.join(tmp, "rqo.bin")
        b.tofile(fb)
        subprocess.check_call([RQ, "hnsw", str(n), str(d), str(metric), str(M), str(efc), "100", fb, fo], stdout=subprocess.DEVNULL)
        out[name] = np.fromfile(fo, np.uint32)
        out[name + "_rowsum"] = np.float64(b.astype(np.float64).sum())   # guards the regenerated rows
    np.savez_compressed(os.path.join(GOLDEN, "rabitq_hnsw_ref.npz"), **out)


def bruteforce_fixture(tmp):
    """hnswlib::BruteforceSearch::searchKnn (bruteforce.h:106-135) of the compiled reference: continuous L2, tie-heavy integer
    L2 (ties across the k-th boundary) and inner product."""
    out = {}
    cases = {"l2_cont": ("l2", mixture(3000, 24, 201), mixture(40, 24, 202)),
             "l2_int": ("l2", mixture(3000, 8, 203, lo=2, hi=8, sigma=2.0, integer=True), mixture(40, 8, 204, lo=2, hi=8, sigma=2.0, integer=True)),
             "ip": ("ip", mixture(2000, 48, 205, lo=-1, hi=1, sigma=0.5), mixture(40, 48, 206, lo=-1, hi=1, sigma=0.5))}
    for name, (metric, base, q) in cases.items():
        fb, fq, fo = (os.path.join(tmp, f"bf_{x}.bin") for x in "bqo")
        write_fvecs(fb, base)
        write_fvecs(fq, q)
        for k in (1, 10, 33):
            run("bf", metric, fb, fq, fo, k)
            raw = open(fo, "rb").read()
            nq, kk = np.frombuffer(raw, np.uint32, 2)
            rec = np.frombuffer(raw, np.dtype([("d", "<f4"), ("l", "<u8")]), nq * kk, 8).reshape(nq, kk)
            out[f"{name}_k{k}_dists"] = rec["d"].copy()
            out[f"{name}_k{k}_labels"] = rec["l"].copy()
        out[f"{name}_base"] = base
        out[f"{name}_queries"] = q
    np.savez_compressed(os.path.join(GOLDEN, "bruteforce_ref.npz"), **out)


def searchbuffer_fixture(tmp):
    """Event order of the compiled rabitqlib::buffer::SearchBuffer (rabitqlib/utils/buffer.hpp:16-100) under random
    insert / pop sequences, continuous and tie-heavy distances, several capacities."""
    RQ = os.path.join(ROOT, "oracle", "_ref", "ref_rabitq")
    rng = np.random.default_rng(4242)
    out = {}
    for ci, (cap, n, ties) in enumerate(((8, 400, True), (32, 1500, True), (64, 3000, False), (200, 6000, True), (1, 50, True))):
        op = (rng.random(n) < 0.72).astype(np.uint8)
        op[:3] = 1
        ids = rng.integers(0, 1 << 20, n).astype(np.uint32)
        d = rng.integers(0, 40, n).astype(np.float32) if ties else rng.random(n).astype(np.float32)
        # drift towards smaller distances, like a search that converges
        d = (d * np.linspace(1.5, 0.6, n)).astype(np.float32) if not ties else d
        f = {k: os.path.join(tmp, f"sb_{k}.bin") for k in ("op", "ids", "d", "ev", "fin")}
        op.tofile(f["op"]); ids.tofile(f["ids"]); d.tofile(f["d"])
        subprocess.check_call([RQ, "buffer", str(cap), str(n), f["op"], f["ids"], f["d"], f["ev"], f["fin"]])
        fin = open(f["fin"], "rb").read()
        sz = int(np.frombuffer(fin, np.uint32, 1)[0])
        rec = np.frombuffer(fin, np.dtype([("id", "<u4"), ("d", "<f4")]), sz, 4)
        p = f"c{ci}_"
        out.update({p + "cap": np.array(cap), p + "op": op, p + "ids": ids, p + "d": d, p + "ev": np.fromfile(f["ev"], np.uint32),
                    p + "final_id": rec["id"].copy(), p + "final_d": rec["d"].copy()})
    out["n_cases"] = np.array(5)
    np.savez_compressed(os.path.join(GOLDEN, "searchbuffer_ref.npz"), **out)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    only = set(sys.argv[1:])   # e.g. `make_golden.py bf buffer`: (re)generate just these, leaving the other files byte-stable
    if only:
        with tempfile.TemporaryDirectory() as tmp:
            if "bf" in only:
                bruteforce_fixture(tmp)
            if "buffer" in only:
                searchbuffer_fixture(tmp)
            if "cent" in only:
                rabitq_cent_fixture(tmp)
            if "rqhnsw" in only:
                rq_hnsw_fixture(tmp)
        print("golden fixtures written:", sorted(only))
        return
    with tempfile.TemporaryDirectory() as tmp:
        bruteforce_fixture(tmp)
        searchbuffer_fixture(tmp)
        dist_fixture(tmp)
        # continuous (tie-free) L2, d=32
        index_fixture(tmp, "l2_cont_d32", "l2", mixture(2000, 32, 1), mixture(100, 32, 2), 8, 100, [10, 32, 64])
        # integer-valued (tie-heavy) L2, d=16, tiny value range -> many equal distances
        bi = mixture(2000, 16, 3, lo=2, hi=8, sigma=2.0, integer=True)
        qi = mixture(100, 16, 4, lo=2, hi=8, sigma=2.0, integer=True)
        index_fixture(tmp, "l2_int_d16", "l2", bi, qi, 8, 100, [10, 32, 64])
        # inner product, normalised rows, d=48
        b = mixture(1500, 48, 5, lo=-1, hi=1, sigma=0.5)
        q = mixture(100, 48, 6, lo=-1, hi=1, sigma=0.5)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        index_fixture(tmp, "ip_d48", "ip", b.astype(np.float32), q.astype(np.float32), 8, 100, [10, 48])
        deleted_fixture(tmp, "l2_cont_d32", "l2", 32, 7, [10, 32, 64])
        deleted_fixture(tmp, "l2_int_d16", "l2", 16, 5, [10, 48])
        filter_fixture(tmp, "l2_cont_d32", "l2", 3, 1, [10, 32, 64])
        filter_fixture(tmp, "l2_int_d16_del", "l2", 4, 0, [10, 48])   # delete marks AND a filter
        # dims off the SIMD16 path: L2SqrSIMD4Ext (d=20), SIMD16ExtResiduals (d=21), SIMD4ExtResiduals (d=10)
        for d, seed in ((20, 11), (21, 13), (10, 15)):
            index_fixture(tmp, f"l2_cont_d{d}", "l2", mixture(600, d, seed), mixture(40, d, seed + 1), 8, 60, [10, 32])
        rabitq_fixture(tmp)
        rabitq_cent_fixture(tmp)
        rq_hnsw_fixture(tmp)
        # inner product off the SIMD16 path: SIMD4ExtAVX (d=20), SIMD16ExtResiduals (d=21), SIMD4ExtResiduals (d=10)
        ip_odd_dims_fixture(tmp)
        for d, seed in ((20, 21), (21, 23), (10, 25)):
            b = mixture(600, d, seed, lo=-1, hi=1, sigma=0.5)
            q = mixture(40, d, seed + 1, lo=-1, hi=1, sigma=0.5)
            b /= np.linalg.norm(b, axis=1, keepdims=True)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            index_fixture(tmp, f"ip_d{d}", "ip", b.astype(np.float32), q.astype(np.float32), 8, 60, [10, 32])
    print("golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
