"""The flat kernel's wide shapes (csrc/flat_search.hip): three slots per lane (ef 129..192, exact fit at 192), result sets beyond 256
entries (S = 6 for ef <= 384, S = 8 for ef <= 512)
and rows beyond 1 KB (d = 960 compiled in: two rounds of thirty 8-byte loads per lane; any other dim > 256: rounds of sixteen),
against the oracle (hnswalg_slim.h:321-457 / hnswalg.h:326-479): labels, fp32 distance bits, the three traversal counters, on
tie-heavy integer rows and on continuous ones, L2 and inner product, Slim and vanilla files."""
import numpy as np
import pytest

from hsutil import Oracle, load_product, mixture
from test_gpu_parity import _pq_sorted

pytestmark = pytest.mark.gpu
L2, IP = 0, 1


@pytest.fixture(scope="module")
def env():
    return load_product(), Oracle()


def _rows(n, d, seed, integer, metric):
    if integer:
        return np.ascontiguousarray(mixture(n, d, seed, n_clusters=12, lo=0, hi=6, sigma=1.5, integer=True))
    x = mixture(n, d, seed, n_clusters=12, lo=-1, hi=1, sigma=0.4)
    if metric == IP:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x.astype(np.float32))


@pytest.mark.parametrize("d,metric,integer", [(128, L2, True), (96, L2, False), (960, L2, True), (960, IP, False), (512, L2, False), (320, L2, True),
                                              (64, IP, False)])
def test_flat_kernel_wide_result_sets_and_long_rows(env, tmp_path, d, metric, integer):
    P, O = env
    n = 6000 if d <= 128 else 2500
    base, q = _rows(n, d, 31 + d, integer, metric), _rows(64, d, 77 + d, integer, metric)
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    P.build_hnsw(base, hp, metric=metric, M=12, ef_construction=80, threads=8)
    P.convert_slim(hp, sp, d, metric=metric, threads=8)
    for kind, path, okind in ((P.HS_KIND_SLIM, sp, "slim"), (P.HS_KIND_HNSW, hp, "hnsw")):
        ix = P.Index(path, kind, d, metric=metric)
        ox = O.load(path, okind, metric, d)
        for ef, k in ((129, 10), (160, 33), (192, 64), (257, 10), (320, 40), (384, 10), (385, 64), (500, 10), (512, 64)) if kind == P.HS_KIND_SLIM else ((192, 10), (300, 10), (512, 20)):
            cfg = f"d={d} metric={metric} int={integer} kind={okind} ef={ef} k={k}"
            ix.set_ef(ef); ox.set_ef(ef)
            o, g = ox.search_pq(q, k, threads=8), ix.search_pq(q, k, want_stats=True)
            assert ix.last_kernel() == "hs::flat_kernel", cfg
            assert np.array_equal(g["cnt"], o["cnt"]), cfg
            assert _pq_sorted(g["dists"], g["labels"], g["cnt"]) == _pq_sorted(o["dists"], o["labels"], o["cnt"]), cfg
            if kind == P.HS_KIND_SLIM:
                oi = ox.search_ids(q, k, threads=8)
                ix.set_exact_order(True)
                r = ix.search_ids(q, k, want_stats=True)
                ix.set_exact_order(False)
                assert np.array_equal(r["labels"], oi["labels"]), cfg
                assert np.array_equal(r["stats"][:, :3], oi["counters"][:, :3]), cfg


def test_flat_kernel_long_logs_at_ef_512(env, tmp_path):
    """ef = 512 on a graph small enough that every query visits most of it: the insertion log and the per-hop counts run long
    (capi.cpp log_cap_for / hop_cap_for) and ties at the bound are frequent on integer rows."""
    P, O = env
    d = 48
    base = np.ascontiguousarray(mixture(20000, d, 5, n_clusters=6, lo=0, hi=4, sigma=1.2, integer=True))
    q = np.ascontiguousarray(mixture(128, d, 6, n_clusters=6, lo=0, hi=4, sigma=1.2, integer=True))
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    P.build_hnsw(base, hp, M=16, ef_construction=100, threads=8)
    P.convert_slim(hp, sp, d, threads=8)
    ix, ox = P.Index(sp, P.HS_KIND_SLIM, d), O.load(sp, "slim", L2, d)
    for ef in (400, 512):
        ix.set_ef(ef); ox.set_ef(ef)
        o = ox.search_ids(q, 10, threads=8)
        r = ix.search_ids(q, 10, want_dists=True, want_stats=True)   # (set_exact_order(True) would hand the query to the strict kernel)
        assert ix.last_kernel() == "hs::flat_kernel"
        assert np.array_equal(np.sort(r["labels"], 1), np.sort(o["labels"], 1)), ef
        assert np.array_equal(r["stats"][:, :3], o["counters"][:, :3]), ef
