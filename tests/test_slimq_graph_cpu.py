"""The graph under HNSW-SlimQ, built the way the reference builds it (CPU harness, SURVEY.md 8 row a12/"next"):

  * hs_build_rabitq_hnsw  == rabitqlib::hnsw::HierarchicalNSW::construct's edges (third_party/rabitqlib/index/hnsw/hnsw.hpp:667-1054),
    pinned edge for edge against the compiled rabitqlib (tests/golden/rabitq_hnsw_ref.npz, made by make_golden.py rqhnsw from
    oracle/_ref/ref_rabitq's `hnsw` command; serial build, seed 100);
  * hs_convert_slimq_graph == HierarchicalNSWSlimQ::convertFromHNSW's graph passes (hnswalg_slimq.h:1546-1762) with its own
    PruneByHeuristic (:1334-1362).  hnswalg_slimq.h needs folly and cannot be compiled here, so this half is checked against the
    plain-Python restatement below on integer-valued rows (every distance exact in any summation order).
"""
import os
import sys

import numpy as np
import pytest

from hsutil import GOLDEN, load_chal_encode, load_product

sys.path.insert(0, GOLDEN)
from make_golden import RQ_HNSW_CASES, rq_hnsw_base  # noqa: E402


@pytest.fixture(scope="module")
def hs():
    return load_product()


def flat_graph(g):
    out = [g["maxlevel"], g["enterpoint"]]
    for i in range(g["count"]):
        out += [int(g["labels"][i]), len(g["lists"][i]) - 1]
        for l in g["lists"][i]:
            out.append(len(l))
            out += [int(x) for x in l]
    return np.array(out, np.uint32)


@pytest.mark.parametrize("case", RQ_HNSW_CASES, ids=[c[0] for c in RQ_HNSW_CASES])
def test_rabitq_builder_matches_compiled_rabitqlib(hs, tmp_path, case):
    name, n, d, metric, M, efc, integer = case
    ref = np.load(os.path.join(GOLDEN, "rabitq_hnsw_ref.npz"))
    base = rq_hnsw_base(n, d, metric, integer)
    assert float(base.astype(np.float64).sum()) == float(ref[name + "_rowsum"]), "regenerated rows differ from the fixture's"
    p = str(tmp_path / "rq.bin")
    hs.build_rabitq_hnsw(base, p, metric=metric, M=M, ef_construction=efc, seed=100, threads=1)
    g = load_chal_encode().parse_vanilla(open(p, "rb").read())
    assert g["M"] == M and g["maxM"] == M and g["maxM0"] == 2 * M and g["efC"] == max(efc, M)
    assert np.array_equal(g["rows"], base)
    got = flat_graph(g)
    want = ref[name]
    assert got.shape == want.shape and np.array_equal(got, want)


def test_rabitq_builder_parallel_keeps_ids_and_is_connected(hs, tmp_path):
    name, n, d, metric, M, efc, integer = RQ_HNSW_CASES[0]
    base = rq_hnsw_base(n, d, metric, integer)
    p = str(tmp_path / "rq.bin")
    hs.build_rabitq_hnsw(base, p, metric=metric, M=M, ef_construction=efc, seed=100, threads=4)
    g = load_chal_encode().parse_vanilla(open(p, "rb").read())
    assert np.array_equal(g["labels"], np.arange(n, dtype=np.uint64)) and np.array_equal(g["rows"], base)
    seen = {int(g["enterpoint"])}
    todo = [int(g["enterpoint"])]
    while todo:   # reachability over all levels from the entry point
        u = todo.pop()
        for l in g["lists"][u]:
            for v in l:
                if int(v) not in seen:
                    seen.add(int(v))
                    todo.append(int(v))
    assert len(seen) > 0.99 * n


def slimq_graph_restated(g, rows, thr_level, pct0, pct, top_M0, low_m0, top_M, low_m):
    """hnswalg_slimq.h:1546-1762 (graph part) in plain Python; rows integer-valued L2 -> exact distances."""
    n, maxlevel, maxM, maxM0 = g["count"], g["maxlevel"], g["maxM"], g["maxM0"]
    lists = g["lists"]
    r64 = rows.astype(np.int64)

    def dist(a, b):
        x = r64[a] - r64[b]
        return int((x * x).sum())
    hist = np.zeros((maxlevel + 1, maxM0 + 2), np.int64)
    level_cnts = np.zeros(maxlevel + 1, np.int64)   # [0] is never counted (:1551-1562)
    for i in range(n):
        for l in range(1, len(lists[i])):
            level_cnts[l] += 1
            hist[l][len(lists[i][l])] += 1
        hist[0][len(lists[i][0])] += 1
    thr = [0] * (maxlevel + 1)
    for l in range(maxlevel + 1):
        top_n = int(level_cnts[l] * np.float32(pct0 if l == 0 else pct) + 0.5)
        acc = 0
        for deg in range(maxM0 + 1, 0, -1):
            acc += hist[l][deg]
            if acc >= top_n:
                thr[l] = deg
                break
    unstable = 0

    def prune(v_sorted, lim):   # :1334-1362 -- `i` below is the LOOP INDEX used as a node id
        out = []
        for i, (dd, nb) in enumerate(v_sorted):
            if len(out) >= lim:
                break
            good = True
            if out and dist(i, nb) < dd:
                good = False
            if good:
                out.append(nb)
        return out

    def by_dist(v, ids):
        nonlocal unstable
        pairs = sorted(((dist(v, int(u)), k, int(u)) for k, u in enumerate(ids)), key=lambda t: (t[0], t[1]))   # stable
        if len(pairs) > 16 and len({p[0] for p in pairs}) != len(pairs):
            unstable += 1   # std::sort's order among equal keys is introsort's beyond 16 elements: outside this restatement
        return [(p[0], p[2]) for p in pairs]
    nn = [[None] * len(lists[v]) for v in range(n)]
    rev = [[[] for _ in lists[v]] for v in range(n)]
    for v in range(n):
        for l, ids in enumerate(lists[v]):
            size = len(ids)
            lim = (top_M0 if size > thr[l] else low_m0) if l == 0 else (top_M if size > thr[l] else low_m)
            nn[v][l] = prune(by_dist(v, ids), lim)
    for v in range(n):
        for l in range(len(lists[v])):
            for u in nn[v][l]:
                rev[u][l].append(v)
    out = []
    for v in range(n):
        node = []
        for l in range(len(lists[v])):
            ids = sorted(set(nn[v][l]) | set(rev[v][l]))
            lim = maxM0 if l == 0 else maxM
            if len(ids) > lim:
                ids = prune(by_dist(v, ids), lim)
            if l != thr_level:   # hierarchical filter (:1724-1736): off the threshold level only neighbours whose top level is l stay
                ids = [u for u in ids if len(lists[u]) - 1 == l]
            node.append(ids)
        out.append(node)
    return out, unstable


def test_slimq_graph_conversion_matches_python_restatement(hs, tmp_path):
    rng = np.random.default_rng(77)
    n, d, M = 700, 64, 8
    centres = rng.integers(-60, 60, (12, d))
    base = (centres[rng.integers(0, 12, n)] + rng.integers(-25, 26, (n, d))).astype(np.float32)
    hp, sp, vp = (str(tmp_path / x) for x in ("h.bin", "s.bin", "v.bin"))
    hs.build_rabitq_hnsw(base, hp, M=M, ef_construction=40, threads=1)
    kw = dict(threshold_level=0, top_degree_percent0=0.02, top_degree_percent=0.02, top_degree_M0=12, low_degree_m0=5, top_degree_M=6,
              low_degree_m=3)
    hs.convert_slimq_graph(hp, sp, d, threads=2, **kw)
    ce = load_chal_encode()
    g = ce.parse_vanilla(open(hp, "rb").read())
    s = ce.parse_slim(open(sp, "rb").read(), d)
    want, unstable = slimq_graph_restated(g, base, kw["threshold_level"], kw["top_degree_percent0"], kw["top_degree_percent"], kw["top_degree_M0"],
                                          kw["low_degree_m0"], kw["top_degree_M"], kw["low_degree_m"])
    assert unstable == 0, "premise: no equal keys in a list of more than 16 (choose another seed)"
    n_reprune = 0
    for v in range(n):
        assert len(s["lists"][v]) == len(want[v])
        for l in range(len(want[v])):
            assert [int(x) for x in s["lists"][v][l]] == want[v][l], (v, l)
            n_reprune += want[v][l] != sorted(want[v][l])
    assert sum(len(w) > 1 and any(w[1:]) for w in want) > 0, "no upper-level list survived the filter"
    assert n_reprune > 0, "the re-prune path (:1690-1716) was not exercised"
    # and it is NOT what Slim's own heuristic (hnswalg_slim.h:836-865) gives on the same input
    hs.convert_slim(hp, vp, d, threads=2, **kw)
    v = ce.parse_slim(open(vp, "rb").read(), d)
    assert any([int(x) for x in v["lists"][i][0]] != want[i][0] for i in range(n))
