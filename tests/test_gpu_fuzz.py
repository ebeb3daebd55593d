"""Randomised differential test: many small random configurations (size, dim, M, ef, k, value range, delete marks,
filters, threshold_level, SlimQ cluster counts) through the C ABI against the oracle.  Seeds are fixed, so a failure
names its configuration."""
import numpy as np
import pytest

from hsutil import Oracle, load_product, mixture
from test_gpu_parity import _pq_sorted
from test_gpu_slimq import kmeans

pytestmark = pytest.mark.gpu
L2, IP = 0, 1


@pytest.fixture(scope="module")
def env():
    return load_product(), Oracle()


def _data(rng, n, d, integer, nq=48):
    lo, hi = (0, int(rng.integers(3, 40))) if integer else (0.0, 1.0)
    sigma = (hi - lo) * float(rng.uniform(0.05, 0.3))
    seed = int(rng.integers(1, 1 << 30))
    x = mixture(n + nq, d, seed, n_clusters=int(rng.integers(2, 24)), lo=lo, hi=hi, sigma=sigma, integer=integer)
    return np.ascontiguousarray(x[:n]), np.ascontiguousarray(x[n:])


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_slim(env, tmp_path, seed):
    P, O = env
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([4, 7, 12, 16, 20, 33, 48, 64, 100, 128, 130]))
    n = int(rng.integers(200, 9000))
    M = int(rng.choice([4, 8, 16, 24]))
    integer = bool(rng.integers(0, 2))
    base, q = _data(rng, n, d, integer)
    thr = int(rng.choice([0, 0, 0, 1]))
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    P.build_hnsw(base, hp, M=M, ef_construction=int(rng.integers(20, 120)), threads=4)
    P.convert_slim(hp, sp, d, threads=4, threshold_level=thr, low_degree_m0=int(rng.choice([4, 8, 12])), top_degree_M0=int(rng.choice([16, 32])))
    ix = P.Index(sp, P.HS_KIND_SLIM, d)
    ox = O.load(sp, "slim", L2, d)
    allowed = (rng.random(n) < 0.7).astype(np.uint8) if seed % 3 == 0 else None
    for _ in range(3):
        ef, k = int(rng.integers(1, 520)), int(rng.integers(1, 40))
        cfg = f"seed={seed} n={n} d={d} M={M} int={integer} thr={thr} ef={ef} k={k}"
        if min(ef, k) > n:
            continue
        ix.set_ef(ef); ox.set_ef(ef)
        o = ox.search_ids(q, k, threads=4)
        enough = o["raw_sz"] >= k   # fewer than k reachable: nth_element on top_size < k is UB in the reference (slim.h:2126)
        ix.set_exact_order(True)
        r = ix.search_ids(q, k, want_stats=True)
        assert np.array_equal(r["labels"][enough], o["labels"][enough]), cfg
        assert np.array_equal(r["stats"][:, :3], o["counters"][:, :3]), cfg
        ix.set_exact_order(False)
        f = ix.search_ids(q, k, want_stats=True)
        assert np.array_equal(np.sort(f["labels"][enough], axis=1), np.sort(o["labels"][enough], axis=1)), cfg
        assert np.array_equal(f["stats"][:, :3], o["counters"][:, :3]), cfg
        op, fp = ox.search_pq(q, k, threads=4), ix.search_pq(q, k)
        assert np.array_equal(fp["cnt"], op["cnt"]), cfg
        assert _pq_sorted(fp["dists"], fp["labels"], fp["cnt"]) == _pq_sorted(op["dists"], op["labels"], op["cnt"]), cfg
        if allowed is not None and thr == 0:
            ox.set_filter(allowed)
            of, gf = ox.search_pq(q, k, threads=4), ix.search_filtered(q, k, allowed)
            ox.set_filter(None)
            assert np.array_equal(gf["cnt"], of["cnt"]), cfg
            assert _pq_sorted(gf["dists"], gf["labels"], gf["cnt"]) == _pq_sorted(of["dists"], of["labels"], of["cnt"]), cfg


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_vanilla(env, tmp_path, seed):
    P, O = env
    rng = np.random.default_rng(2000 + seed)
    metric = IP if seed % 4 == 3 else L2
    d = int(rng.choice([16, 32, 48, 64])) if metric == IP else int(rng.choice([3, 8, 16, 21, 40, 96, 128]))
    n = int(rng.integers(100, 2500))
    M = int(rng.choice([4, 8, 16, 40]))
    base, q = _data(rng, n, d, bool(rng.integers(0, 2)) and metric == L2)
    hp = str(tmp_path / "h.bin")
    P.build_hnsw(base, hp, metric=metric, M=M, ef_construction=int(rng.integers(20, 120)), threads=4)
    ix = P.Index(hp, P.HS_KIND_HNSW, d, metric)
    ox = O.load(hp, "hnsw", metric, d)
    for _ in range(3):
        ef, k = int(rng.integers(1, 520)), int(rng.integers(1, 40))
        cfg = f"seed={seed} n={n} d={d} M={M} metric={metric} ef={ef} k={k}"
        ix.set_ef(ef); ox.set_ef(ef)
        op = ox.search_pq(q, k, threads=4)
        for exact in (True, False):
            ix.set_exact_order(exact)
            fp = ix.search_pq(q, k, want_stats=True)
            assert np.array_equal(fp["cnt"], op["cnt"]), cfg
            assert _pq_sorted(fp["dists"], fp["labels"], fp["cnt"]) == _pq_sorted(op["dists"], op["labels"], op["cnt"]), cfg
            assert np.array_equal(fp["stats"][:, :3], op["counters"][:, :3]), cfg


@pytest.mark.parametrize("seed", range(20))
def test_fuzz_slimq(env, tmp_path, seed):
    P, O = env
    rng = np.random.default_rng(3000 + seed)
    metric = IP if seed % 4 == 3 else L2
    d = int(rng.choice([64, 128, 256])) if metric == IP else int(rng.choice([64, 70, 96, 100, 128, 200, 256]))
    n = int(rng.integers(300, 8000))
    integer = bool(rng.integers(0, 2)) and metric == L2
    base, q = _data(rng, n, d, integer)
    if metric == IP:
        base /= np.linalg.norm(base, axis=1, keepdims=True) + 1e-9
        q /= np.linalg.norm(q, axis=1, keepdims=True) + 1e-9
    hp, sp, qp = (str(tmp_path / f) for f in ("h.bin", "s.bin", "q.bin"))
    P.build_hnsw(base, hp, metric=metric, M=int(rng.choice([8, 16])), ef_construction=60, threads=4)
    P.convert_slim(hp, sp, d, metric=metric, threads=4, threshold_level=int(rng.choice([0, 0, 1])))
    P.convert_slimq(sp, metric, d, kmeans(base, int(rng.choice([1, 3, 16]))), qp, flip_seed=int(rng.integers(1, 99)), threads=4)
    ix = P.Index(qp, P.HS_KIND_SLIMQ, d, metric=metric)
    ix.slimq_set_dataset(base)
    ox = O.load_slimq(qp)
    tc = float(rng.uniform(20, 120))
    ix.slimq_set_tconst(tc)
    for _ in range(3):
        ef, k = int(rng.integers(1, 400)), int(rng.integers(1, 80))
        cfg = f"seed={seed} n={n} d={d} metric={metric} int={integer} ef={ef} k={k}"
        ix.set_ef(ef); ox.set(ef, tc, base)
        got, ref = ix.slimq_search(q, k, want_stats=True), ox.search(q, k, threads=4)
        assert np.array_equal(got["stats"].astype(np.uint64), ref["counters"]), cfg
        assert np.array_equal(got["cnt"], ref["counts"]), cfg
        assert np.array_equal(got["labels"], ref["labels"]), cfg
        assert np.array_equal(got["dists"].view(np.uint32), ref["dists"].view(np.uint32)), cfg
