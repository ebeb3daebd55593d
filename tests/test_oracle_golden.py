"""Pin the CPU restatement (oracle/) against outputs of the COMPILED REFERENCE (tests/golden/*,
made by tests/golden/make_golden.py from oracle/_ref/ref_hnsw).  CPU-only."""
import os

import numpy as np
import pytest

from hsutil import GOLDEN

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
