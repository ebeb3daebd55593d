"""Pin the CPU restatement (oracle/) against outputs of the COMPILED REFERENCE (tests/golden/*,
made by tests/golden/make_golden.py from oracle/_ref/ref_hnsw).  CPU-only."""
import os

import numpy as np
import pytest

from hsutil import GOLDEN, load_chal_encode

L2, IP = 0, 1


def test_distance_recipes_bit_exact(oracle):
    checked = 0
    for fname in ("dist_ref.npz", "dist_ref_ip_odd.npz"):
        g = np.load(os.path.join(GOLDEN, fname))
        for key in g.files:
            if not key.endswith("_ref"):
                continue
            metric, d = key.split("_")[:2]
            a, b, ref = g[f"{metric}_{d}_a"], g[f"{metric}_{d}_b"], g[key]
            got = oracle.dist(L2 if metric == "l2" else IP, a, b)
            assert got.tobytes() == ref.tobytes(), f"{metric} d={d}: restated recipe differs from the reference"
            checked += 1
    assert checked >= 25


@pytest.mark.parametrize("name,metric,dim", [("l2_cont_d32", L2, 32), ("l2_int_d16", L2, 16), ("ip_d48", IP, 48),
                                             ("l2_cont_d20", L2, 20), ("l2_cont_d21", L2, 21), ("l2_cont_d10", L2, 10),
                                             ("ip_d20", IP, 20), ("ip_d21", IP, 21), ("ip_d10", IP, 10)])
def test_vanilla_search_matches_reference(oracle, name, metric, dim):
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = oracle.load(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "hnsw", metric, dim)
    k = int(g["k"])
    for ef in g["efs"]:
        ix.set_ef(int(ef))
        r = ix.search_pq(g["queries"], k)
        assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
        # labels and fp32 distances bit-exact, in the reference's priority_queue pop order
        assert np.array_equal(r["labels"], g[f"ef{ef}_labels"]), f"{name} ef={ef}"
        assert r["dists"].tobytes() == g[f"ef{ef}_dists"].tobytes()
        # distance-function call count of the reference == the oracle's n_dist counter
        assert np.array_equal(r["counters"][:, 0], g[f"ef{ef}_calls"])


def test_load_errors(oracle, tmp_path):
    with pytest.raises(RuntimeError, match="Cannot open file"):
        oracle.load(str(tmp_path / "nope.bin"), "hnsw", L2, 32)
    bad = tmp_path / "trunc.bin"
    data = open(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), "rb").read()
    bad.write_bytes(data[: len(data) // 2])
    with pytest.raises(RuntimeError, match="corrupted"):
        oracle.load(str(bad), "hnsw", L2, 32)


@pytest.mark.parametrize("name,dim", [("l2_cont_d32_del", 32), ("l2_int_d16_del", 16)])
def test_vanilla_search_with_delete_marks_matches_reference(oracle, name, dim):
    """Index saved by the reference after markDelete: the !bare_bone_search branch (hnswalg.h:347-349,441-444)."""
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = oracle.load(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "hnsw", L2, dim)
    k = int(g["k"])
    every = int(g["every"])
    for ef in g["efs"]:
        ix.set_ef(int(ef))
        r = ix.search_pq(g["queries"], k)
        assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
        assert np.array_equal(r["labels"], g[f"ef{ef}_labels"])
        assert r["dists"].tobytes() == g[f"ef{ef}_dists"].tobytes()
        assert np.array_equal(r["counters"][:, 0], g[f"ef{ef}_calls"])
        valid = np.arange(k)[None, :] < r["cnt"][:, None]
        assert not np.any((r["labels"][valid] % every) == every // 2), "a deleted label was returned"


@pytest.mark.parametrize("name,dim", [("l2_cont_d32", 32), ("l2_int_d16_del", 16)])
def test_vanilla_filtered_search_matches_reference(oracle, name, dim):
    """searchKnn(q, k, isIdAllowed) of the compiled reference (filter: label % mod != rem)."""
    g = np.load(os.path.join(GOLDEN, f"{name}_filter.npz"))
    ix = oracle.load(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "hnsw", L2, dim)
    labels = np.arange(ix.count)  # golden indexes are built with label == internal id
    ix.set_filter((labels % int(g["mod"]) != int(g["rem"])).astype(np.uint8))
    k = int(g["k"])
    for ef in g["efs"]:
        ix.set_ef(int(ef))
        r = ix.search_pq(g["queries"], k)
        assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
        assert np.array_equal(r["labels"], g[f"ef{ef}_labels"])
        assert r["dists"].tobytes() == g[f"ef{ef}_dists"].tobytes()
        assert np.array_equal(r["counters"][:, 0], g[f"ef{ef}_calls"])


# ---- HierarchicalNSWSlim cross-pinned to the compiled vanilla reference --------------------------------------------------
# hnswalg_slim.h itself cannot be compiled here (folly).  A graph the compiled reference built, re-encoded VERBATIM as a Slim
# file (oracle/chal_encode.py: every list kept, reference order, threshold_level 0), makes HierarchicalNSWSlim::searchKnn
# (hnswalg_slim.h:2030-2131) walk exactly what HierarchicalNSW::searchKnn (hnswalg.h:1378-1440) walks: same descent, same
# level-0 beam.  So the Slim restatement -- loader, CHAL slice addressing, beam, nth_element finish -- must reproduce the
# compiled reference's golden k-set, fp32 distances and distance-call count (minus the entry distance that only the vanilla
# searchBaseLayerST recomputes, hnswalg.h:347-351).
PLAIN = [("l2_cont_d32", L2, 32), ("l2_int_d16", L2, 16), ("ip_d48", IP, 48), ("l2_cont_d20", L2, 20), ("l2_cont_d21", L2, 21),
         ("l2_cont_d10", L2, 10), ("ip_d20", IP, 20), ("ip_d21", IP, 21), ("ip_d10", IP, 10)]


def verbatim_slim_file(tmp_path, name, **kw):
    ce = load_chal_encode()
    raw = open(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "rb").read()
    sp = tmp_path / f"{name}.verbatim.slim"
    sp.write_bytes(ce.vanilla_to_slim_verbatim(raw, **kw))
    return str(sp)


def check_ids_against_golden(g, ef, labels, raw_d, raw_i, raw_sz, n_dist, tie_free):
    """labels: nq x k of searchKnn(q,k,tableint*); raw_*: the top_candidates arrays (id -> distance); golden: pq pop order."""
    k = labels.shape[1]
    for i in range(labels.shape[0]):
        assert g[f"ef{ef}_cnt"][i] == k
        d_of = dict(zip(raw_i[i, :raw_sz[i]].tolist(), raw_d[i, :raw_sz[i]].view(np.uint32).tolist()))
        got = sorted((d_of[int(l)], int(l)) for l in labels[i])   # golden indexes: label == internal id
        want = sorted(zip(g[f"ef{ef}_dists"][i].view(np.uint32).tolist(), g[f"ef{ef}_labels"][i].tolist()))
        if tie_free:
            assert got == want, f"ef={ef} query {i}"
        else:   # equal distances across the k-th boundary: the two classes pick by heap layout; the distance multiset is unique
            assert [x[0] for x in got] == [x[0] for x in want], f"ef={ef} query {i}"
    assert np.array_equal(n_dist, g[f"ef{ef}_calls"] - 1), "distance evaluations differ from the compiled reference"


@pytest.mark.parametrize("name,metric,dim", PLAIN)
def test_slim_search_on_verbatim_encoding_matches_reference(oracle, tmp_path, name, metric, dim):
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = oracle.load(verbatim_slim_file(tmp_path, name), "slim", metric, dim)
    k = int(g["k"])
    for ef in g["efs"]:
        ef = int(ef)
        ix.set_ef(ef)
        r = ix.search_ids(g["queries"], k)
        check_ids_against_golden(g, ef, r["labels"], r["raw_d"], r["raw_i"], r["raw_sz"], r["counters"][:, 0], tie_free=name != "l2_int_d16")


def test_slim_filter_branch_on_verbatim_encoding_matches_reference(oracle, tmp_path):
    """searchKnn(q,k,isIdAllowed): the !bare_bone beam with a filter (hnswalg_slim.h:462-618 as reached from :1783-1905)
    against the compiled vanilla reference's filtered search.  The two classes treat the ENTRY differently -- Slim pre-marks
    the global enter point (:1796) and always seeds the result heap with the level-0 entry (:2100), vanilla does neither
    (hnswalg.h:347-362) -- so the comparison runs the restatement without the pre-mark and on the queries whose level-0
    entry passes the filter; there the k-set, distances and distance-call counts must be the reference's."""
    name, dim = "l2_cont_d32", 32
    g = np.load(os.path.join(GOLDEN, f"{name}_filter.npz"))
    ix = oracle.load(verbatim_slim_file(tmp_path, name), "slim", L2, dim)
    allowed = (np.arange(ix.count) % int(g["mod"]) != int(g["rem"])).astype(np.uint8)
    ix.set_filter(allowed)
    ix.set_mark_ep(0)
    ok = allowed[ix.entry(g["queries"])] == 1
    assert ok.sum() >= 0.5 * len(ok)
    k = int(g["k"])
    for ef in g["efs"]:
        ef = int(ef)
        ix.set_ef(ef)
        r = ix.search_pq(g["queries"], k)
        assert np.array_equal(r["cnt"][ok], g[f"ef{ef}_cnt"][ok])
        assert np.array_equal(r["labels"][ok], g[f"ef{ef}_labels"][ok])
        assert r["dists"][ok].tobytes() == g[f"ef{ef}_dists"][ok].tobytes()
        assert np.array_equal(r["counters"][ok, 0], g[f"ef{ef}_calls"][ok] - 1)


def test_python_written_slim_file_round_trips(oracle, tmp_path):
    """The Slim loader restatement on a file this repository's C++ did not write (garbage in the 8 stale pointer bytes of
    every element, hnswalg_slim.h:127-131; blobs present iff blobSize != 0 and total_neighbor != 0, :745-748), and the
    independent Python reader consuming it to the last byte."""
    ce = load_chal_encode()
    raw = open(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), "rb").read()
    v = ce.parse_vanilla(raw)
    a, b = ce.write_slim(v, garbage_seed=1), ce.write_slim(v, garbage_seed=2)
    assert a != b and len(a) == len(b)
    s = ce.parse_slim(a, 32)
    assert s["count"] == v["count"] and all(len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y))
                                            for x, y in zip(s["lists"], v["lists"]))
    outs = []
    for i, blob in enumerate((a, b)):
        f = tmp_path / f"g{i}.slim"
        f.write_bytes(blob)
        ix = oracle.load(str(f), "slim", L2, 32)
        ix.set_ef(40)
        outs.append(ix.search_ids(np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))["queries"], 10)["labels"])
    assert np.array_equal(outs[0], outs[1])


def test_brute_force_matches_compiled_bruteforce(oracle):
    """oracle brute_force against hnswlib::BruteforceSearch::searchKnn (bruteforce.h:106-135) of the compiled reference."""
    g = np.load(os.path.join(GOLDEN, "bruteforce_ref.npz"))
    for name, metric in (("l2_cont", L2), ("l2_int", L2), ("ip", IP)):
        base, q = g[f"{name}_base"], g[f"{name}_queries"]
        for k in (1, 10, 33):
            ids = oracle.brute_force(metric, base, q, k)
            want_l = g[f"{name}_k{k}_labels"][:, ::-1]    # pop order is farthest first
            want_d = g[f"{name}_k{k}_dists"][:, ::-1]
            got_d = np.stack([oracle.dist(metric, np.repeat(q[i:i + 1], k, 0), base[ids[i]]) for i in range(len(q))])
            assert got_d.tobytes() == np.ascontiguousarray(want_d).tobytes(), f"{name} k={k}"
            # the reference keeps the k smallest (dist, label) pairs: with label == row that is the oracle's (dist, id) order
            assert np.array_equal(ids, want_l), f"{name} k={k}"


def test_searchbuffer_matches_compiled_rabitqlib(oracle):
    """SearchBuffer restatement (hnswalg_slimq.h:80-151) against the compiled rabitqlib::buffer::SearchBuffer
    (rabitqlib/utils/buffer.hpp:16-100): which inserts land, every popped id in order, the final array."""
    g = np.load(os.path.join(GOLDEN, "searchbuffer_ref.npz"))
    for c in range(int(g["n_cases"])):
        p = f"c{c}_"
        ev, fi, fd = oracle.pool_run(int(g[p + "cap"]), g[p + "op"], g[p + "ids"], g[p + "d"])
        assert np.array_equal(ev, g[p + "ev"]), f"case {c}: event sequence differs"
        assert np.array_equal(fi, g[p + "final_id"]) and fd.tobytes() == g[p + "final_d"].tobytes()
        assert (g[p + "op"] == 0).sum() > 10 and (ev[g[p + "op"] == 1] == 0).sum() > 0


# ---- RaBitQ pieces of the SlimQ oracle against the compiled rabitqlib ------------------------------------------
RQ = np.load(os.path.join(GOLDEN, "rabitq_ref.npz"))


@pytest.mark.parametrize("dim", (128, 96, 768))
def test_slimq_oracle_pieces_match_rabitqlib(oracle, dim):
    p = f"d{dim}_"
    metric, t_const = int(RQ[p + "metric"]), float(RQ[p + "t_const"])
    for src, dst in (("x", "rx"), ("q", "rq")):
        got = oracle.rq_rotate(dim, RQ[p + "flip"], RQ[p + src])
        assert np.array_equal(got.view(np.uint32), RQ[p + dst].view(np.uint32))
    q3, planes, q2c = oracle.rq_prepare(RQ[p + "rq"], metric, t_const, RQ[p + "rc"])
    scale = np.abs(RQ[p + "q3"]).max(axis=0, keepdims=True)
    assert np.all(np.abs(q3 - RQ[p + "q3"]) <= 1e-4 * scale)
    assert np.array_equal(planes, RQ[p + "bins"])
    est = oracle.rq_est(metric, RQ[p + "codes"], RQ[p + "fac"], RQ[p + "q3"], RQ[p + "bins"], RQ[p + "g_add"])
    assert np.array_equal(est.view(np.uint32), RQ[p + "est"][:, :, 1].view(np.uint32))


CQ = np.load(os.path.join(GOLDEN, "rabitq_cent_ref.npz"))


def _ulps(a, b):
    """distance in units in the last place between two float32 arrays of the same sign"""
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("dim", (128, 96, 768))
def test_q_to_centroids_against_compiled_rabitqlib(oracle, dim):
    """q_to_centroids (hnswalg_slimq.h:1823-1848): sqrt(euclidean_sqr) and dot_product of the compiled rabitqlib (Eigen reductions,
    whose order is alignment- and build-dependent) against the restatement's left-to-right fp32 sums.  Pinned to a few units in
    the last place -- the size of the wobble tools/slimq_sensitivity.py perturbs by."""
    rq, cen = RQ[f"d{dim}_rq"], CQ[f"d{dim}_cen"]
    for metric in (0, 1):
        _, _, q2c = oracle.rq_prepare(rq, metric, 41.25, cen)
        want_l2 = np.sqrt(CQ[f"d{dim}_l2sqr"])
        if metric == 0:
            u = _ulps(np.ascontiguousarray(q2c, np.float32), want_l2.astype(np.float32))
            assert u.max() <= 8, f"L2 norms differ by up to {u.max()} ulp"
        else:
            ncl = cen.shape[0]
            ip, l2 = np.ascontiguousarray(q2c[:, :ncl], np.float32), np.ascontiguousarray(q2c[:, ncl:], np.float32)
            assert _ulps(l2, want_l2.astype(np.float32)).max() <= 8
            # a dot product near zero has no meaningful ulp distance: compare against the magnitude of the summands
            scale = (np.abs(rq[:, None, :] * cen[None, :, :])).sum(-1)
            assert np.all(np.abs(ip - CQ[f"d{dim}_ip"]) <= 4e-7 * scale)
