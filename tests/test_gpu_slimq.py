"""HNSW-SlimQ on the GPU against the oracle restatement (oracle/hs_oracle_slimq.hpp), through the C ABI.

Both sides read the same SlimQ index file (written by the product's CPU harness hs_convert_slimq) and use the same
t_const.  Everything is compared exactly: the k-heap array (labels, exact distances, order), the number of entries
and the traversal counters {expansions, estimates, buffer inserts, revisits} -- the counters only agree when every
estimated distance, every SearchBuffer decision and every tie was taken the same way.
"""
import os

import numpy as np
import pytest

from hsutil import Oracle, load_product, sift_like

pytestmark = pytest.mark.gpu


def kmeans(x, k, seed=0, rounds=4):
    rng = np.random.default_rng(seed)
    cen = x[rng.choice(x.shape[0], k, replace=False)].copy()
    for _ in range(rounds):
        d = (x * x).sum(1)[:, None] - 2 * x @ cen.T + (cen * cen).sum(1)[None]
        a = d.argmin(1)
        for c in range(k):
            if (a == c).any():
                cen[c] = x[a == c].mean(0)
    return cen


@pytest.fixture(scope="module")
def env(tmp_path_factory):
    return load_product(), Oracle(), tmp_path_factory.mktemp("slimq")


def build(P, tmp, name, base, metric, ncl, **slim):
    d = base.shape[1]
    h, s, sq = (str(tmp / f"{name}.{e}") for e in ("hnsw", "slim", "slimq"))
    P.build_hnsw(base, h, metric=metric, M=16, ef_construction=100, threads=8)
    P.convert_slim(h, s, d, metric=metric, threads=8, **slim)
    P.convert_slimq(s, metric, d, kmeans(base, ncl), sq, threads=8)
    return sq


def check(P, O, path, base, q, metric, k, efs, t_const=None):
    d = base.shape[1]
    ix = P.Index(path, P.HS_KIND_SLIMQ, d, metric=metric)
    ix.slimq_set_dataset(base)
    if t_const is not None:
        ix.slimq_set_tconst(t_const)
    ox = O.load_slimq(path)
    for ef in efs:
        ix.set_ef(ef)
        ox.set(ef, ix.slimq_tconst(), base)
        if ef == efs[0]:
            gp, op = ix.slimq_prepare_debug(q, ox.padded, ox.ncl), ox.prepare(q)
            for key in ("rq", "q3", "g_add"):
                assert np.array_equal(gp[key].view(np.uint32), op[key].view(np.uint32)), f"query preparation differs: {key}"
            assert np.array_equal(gp["planes"], op["planes"])
        got = ix.slimq_search(q, k, want_stats=True)
        ref = ox.search(q, k, threads=8)
        assert np.array_equal(got["stats"].astype(np.uint64), ref["counters"]), f"traversal differs at ef={ef}"
        assert np.array_equal(got["cnt"], ref["counts"])
        assert np.array_equal(got["labels"], ref["labels"]), f"labels differ at ef={ef}"
        assert np.array_equal(got["dists"].view(np.uint32), ref["dists"].view(np.uint32))
    return ix, ox


def test_slimq_d128_l2(env):
    P, O, tmp = env
    x = sift_like(6000 + 300, 128, seed=11, n_clusters=48)
    base, q = x[:6000], x[6000:]
    path = build(P, tmp, "d128", base, 0, 16)
    ix, ox = check(P, O, path, base, q, 0, 10, (10, 40, 64, 100, 200, 400))
    # sanity: the quantised beam + exact re-rank finds the true neighbours
    gt = O.brute_force(0, base, q, 10)
    ix.set_ef(200)
    got = ix.slimq_search(q, 10)["labels"]
    rec = np.mean([len(set(got[i].tolist()) & set(gt[i].tolist())) / 10 for i in range(q.shape[0])])
    assert rec > 0.9


def test_slimq_ordered_two_launch_pass(env):
    """From 6144 queries the first pass runs as descent / order by entry estimate / level-0 search: same traversal, same answers."""
    P, O, tmp = env
    x = sift_like(4000 + 6500, 128, seed=13, n_clusters=32)
    base, q = x[:4000], x[4000:]
    path = build(P, tmp, "d128_order", base, 0, 8)
    check(P, O, path, base, q, 0, 10, (32, 128))


def test_slimq_integer_ties(env):
    """Small integer coordinates: many equal exact distances (heap ties) and equal estimates."""
    P, O, tmp = env
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, (3000, 64)).astype(np.float32)
    q = rng.integers(0, 4, (200, 64)).astype(np.float32)
    path = build(P, tmp, "int64", base, 0, 4)
    check(P, O, path, base, q, 0, 10, (10, 50, 128))
    check(P, O, path, base, q, 0, 1, (20,))
    check(P, O, path, base, q, 0, 100, (100,))


def test_slimq_d96_padded_rotator(env):
    """dim 96 -> padded 128, truncated Hadamard 64 + Kac walk; runtime-length sign codes."""
    P, O, tmp = env
    x = sift_like(4000 + 200, 96, seed=3, n_clusters=32)
    base, q = x[:4000], x[4000:]
    path = build(P, tmp, "d96", base, 0, 8)
    check(P, O, path, base, q, 0, 10, (30, 100))


def test_slimq_d100_residual_recipes(env):
    """dim 100 (GloVe-like): padded 128, Kac walk, and the exact re-rank through the reference's SIMD4 L2 recipe."""
    P, O, tmp = env
    x = sift_like(3000 + 100, 100, seed=71, n_clusters=16)
    base, q = x[:3000], x[3000:]
    path = build(P, tmp, "d100", base, 0, 8)
    check(P, O, path, base, q, 0, 10, (40, 120))


def test_slimq_d100_ip_residual_recipes(env):
    """Inner product at dim 100: estimator on the padded 128-bit codes, exact re-rank through InnerProductSIMD4ExtAVX."""
    P, O, tmp = env
    rng = np.random.default_rng(19)
    x = rng.standard_normal((3000 + 100, 100)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    base, q = x[:3000], x[3000:]
    path = build(P, tmp, "d100ip", base, 1, 4)
    check(P, O, path, base, q, 1, 10, (40, 120), t_const=31.0)


def test_slimq_d768_ip(env):
    P, O, tmp = env
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2500 + 100, 768)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    base, q = x[:2500], x[2500:]
    path = build(P, tmp, "d768", base, 1, 4)
    check(P, O, path, base, q, 1, 10, (50,), t_const=88.0)


def test_slimq_csr_fallback_layout(env, monkeypatch):
    """The layout wide graphs (level-0 degree > 64) fall back to -- CSR adjacency + separate record array, no fused
    tiles -- forced through the HS_SLIMQ_FUSED=0 knob (the Slim conversion never produces such degrees itself)."""
    P, O, tmp = env
    x = sift_like(3000 + 100, 64, seed=31, n_clusters=8)
    base, q = x[:3000], x[3000:]
    path = build(P, tmp, "csr", base, 0, 4)
    monkeypatch.setenv("HS_SLIMQ_FUSED", "0")
    check(P, O, path, base, q, 0, 10, (40, 150))


def test_slimq_threshold_level_and_errors(env):
    P, O, tmp = env
    x = sift_like(5000 + 100, 128, seed=21, n_clusters=32)
    base, q = x[:5000], x[5000:]
    path = build(P, tmp, "thr", base, 0, 16, threshold_level=1)
    ix, _ = check(P, O, path, base, q, 0, 10, (60,))
    check(P, O, path, base, q[:20], 0, 10, (700, 1024))   # 16 register slots per lane
    ix.set_ef(1100)
    with pytest.raises(P.HsError) as e:
        ix.slimq_search(q, 10)
    assert e.value.status == P.HS_ERR_UNSUPPORTED
    with pytest.raises(P.HsError):
        ix.search_pq(q, 10)
    fresh = P.Index(path, P.HS_KIND_SLIMQ, 128)
    with pytest.raises(P.HsError) as e:
        fresh.slimq_search(q, 10)   # no dataset yet
    assert e.value.status == P.HS_ERR_INVALID


def test_slimq_event_trace_matches_oracle(env):
    """The debug entry: every SearchBuffer pop and insert (with the estimated distance bits) in order."""
    P, O, tmp = env
    x = sift_like(3000 + 6, 128, seed=51, n_clusters=16)
    base, q = x[:3000], x[3000:]
    path = build(P, tmp, "trace", base, 0, 8)
    ix = P.Index(path, P.HS_KIND_SLIMQ, 128)
    ix.slimq_set_dataset(base)
    ox = O.load_slimq(path)
    for ef in (20, 150):
        ix.set_ef(ef); ox.set(ef, ix.slimq_tconst(), base)
        tr, _ = ix.slimq_trace(q, 10, 8192)
        for i in range(q.shape[0]):
            want = ox.trace(q[i], 10, 8192)
            assert np.array_equal(tr[i][:len(want)], want) and np.all(tr[i][len(want):] == 0xFFFFFFFF)


def test_slimq_cpp_facade(env):
    """A caller written against the reference's HierarchicalNSWSlimQ API (tests/facade_smoke.cpp, the call sequence
    of hnsw_slimq_strategy.h): per-query searchKnn(q, K, result) loop and searchKnnBatch."""
    import subprocess
    from hsutil import ROOT
    P, O, tmp = env
    exe = os.path.join(ROOT, "hnsw-slim_amd", "facade_smoke")
    assert os.path.exists(exe)
    x = sift_like(2000 + 30, 64, seed=41, n_clusters=8)
    base, q = np.ascontiguousarray(x[:2000]), np.ascontiguousarray(x[2000:])
    path = build(P, tmp, "facade", base, 0, 4)
    qf, bf, out = str(tmp / "fq.f32"), str(tmp / "fb.f32"), str(tmp / "fo.u32")
    q.tofile(qf)
    base.tofile(bf)
    subprocess.check_call([exe, "slimq", path, "64", qf, "30", "10", "50", out, bf, "2000"])
    got = np.fromfile(out, np.uint32).reshape(2, 30, 10)
    ox = O.load_slimq(path)
    ox.set(50, P.rabitq_default_tconst(64), base)
    want = ox.search(q, 10)
    assert (want["counts"] == 10).all()
    assert np.array_equal(got[0], want["labels"].astype(np.uint32)) and np.array_equal(got[1], want["labels"].astype(np.uint32))


def test_slimq_file_rewritten_by_python(env):
    """A SlimQ file the repository's C++ did not write: the product's file parsed by the independent Python reader
    (oracle/chal_encode.py, to the last byte) and re-serialised from its fields with garbage in every byte the search path
    must not read (the 8 stale pointer bytes of each element, the whole ex-data area) -- product and oracle load it and
    answer exactly as on the original."""
    from hsutil import load_chal_encode
    P, O, tmp = env
    ce = load_chal_encode()
    x = sift_like(3000 + 100, 128, seed=13, n_clusters=24)
    base, q = x[:3000], x[3000:]
    path = build(P, tmp, "rewrite", base, 0, 8)
    raw = open(path, "rb").read()
    g = ce.parse_slimq(raw)
    assert g["ext"][1] == 128 and g["ext"][0] == 8 and len(g["blobs"]) == 3000
    raw2 = ce.write_slimq(g)
    assert len(raw2) == len(raw) and raw2 != raw
    p2 = str(tmp / "rewrite.py.slimq")
    open(p2, "wb").write(raw2)
    a, _ = check(P, O, path, base, q, 0, 10, (40,))
    b, _ = check(P, O, p2, base, q, 0, 10, (40,))
    ra, rb = a.slimq_search(q, 10, want_stats=True), b.slimq_search(q, 10, want_stats=True)
    assert np.array_equal(ra["labels"], rb["labels"]) and ra["dists"].tobytes() == rb["dists"].tobytes() and np.array_equal(ra["stats"], rb["stats"])


def test_slimq_second_pass_with_starved_expanded_set(env):
    """hs_set_capacity(hash_slots = 32): the first pass's expanded-node set overflows on most queries (ST_OVERFLOW) and the second
    pass -- the set per workgroup in global memory, capi.cpp kSlimQFbHash -- must give the oracle's answers."""
    P, O, tmp = env
    x = sift_like(5000 + 200, 128, seed=17, n_clusters=32)
    base, q = x[:5000], x[5000:]
    path = build(P, tmp, "d128_starved", base, 0, 8)
    ix = P.Index(path, P.HS_KIND_SLIMQ, 128)
    ix.slimq_set_dataset(base)
    ox = O.load_slimq(path)
    ix.set_capacity(0, 32)
    for ef in (64, 300):
        ix.set_ef(ef)
        ox.set(ef, ix.slimq_tconst(), base)
        got, ref = ix.slimq_search(q, 10, want_stats=True), ox.search(q, 10, threads=8)
        assert (ref["counters"][:, 0] > 24).mean() > 0.5, "premise: most queries expand more nodes than 75 % of 32 slots"
        assert np.array_equal(got["stats"].astype(np.uint64), ref["counters"]), ef
        assert np.array_equal(got["labels"], ref["labels"]), ef
        assert np.array_equal(got["dists"].view(np.uint32), ref["dists"].view(np.uint32)), ef


def test_slimq_on_the_references_own_graph_pipeline(env):
    """The index built the way hnsw_slimq_strategy.h:106-128 builds it -- rabitqlib-style HNSW (M = 32: level-0 lists of up to 64
    ids) + SlimQ's own PruneByHeuristic -- searched on the GPU and by the oracle: L2 and inner product."""
    P, O, tmp = env
    for metric, d, seed in ((0, 128, 23), (1, 64, 29)):
        x = sift_like(5000 + 200, d, seed=seed, n_clusters=24, integer=False)
        if metric == 1:
            x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
        base, q = np.ascontiguousarray(x[:5000]), np.ascontiguousarray(x[5000:])
        h, s, sq = (str(tmp / f"rq{metric}.{e}") for e in ("hnsw", "slim", "slimq"))
        P.build_rabitq_hnsw(base, h, metric=metric, M=32, ef_construction=128, seed=100, threads=8)
        P.convert_slimq_graph(h, s, d, metric=metric, threads=8)
        P.convert_slimq(s, metric, d, kmeans(base, 8), sq, threads=8)
        ix, _ = check(P, O, sq, base, q, metric, 10, (16, 100, 300))
        assert ix.info()["max_degree0"] > 16, "premise: level-0 lists wider than one 16-id tile"
