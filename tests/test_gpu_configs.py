"""The other BASELINE.json shapes (configs[2..4]) on their round-2 calibrated generators (tools/other_configs.py,
tools/slimq_config.py), at the largest size the oracle covers in a few seconds: GPU results identical to the oracle's
(label sets, counters), and the generators do what they were calibrated for (recall@10 high inside the ef sweep)."""
import numpy as np
import pytest

from hsutil import Oracle, load_product, sift_like

pytestmark = pytest.mark.gpu


def _recall(labels, gt):
    return np.mean([len(set(labels[i].tolist()) & set(gt[i].tolist())) for i in range(len(gt))]) / gt.shape[1]


def _fp32_case(tmp_path, base, q, dim, efs, min_recall):
    hs, O = load_product(), Oracle()
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=8)
    hs.convert_slim_gpu(hp, sp, dim, threads=8)
    ix, ox = hs.Index(sp, hs.HS_KIND_SLIM, dim), O.load(sp, "slim", 0, dim)
    gt = hs.brute_force(base, q, 10)[0]
    best = 0.0
    for ef in efs:
        ix.set_ef(ef); ox.set_ef(ef)
        g, o = ix.search_ids(q, 10, want_stats=True), ox.search_ids(q, 10, threads=8, raw=False)
        assert np.array_equal(np.sort(g["labels"], 1), np.sort(o["labels"], 1)), f"ef={ef}"
        assert np.array_equal(g["stats"][:, :3], o["counters"][:, :3]), f"ef={ef}"
        best = max(best, _recall(g["labels"], gt))
    assert best >= min_recall, best


def test_gist_like_d960(tmp_path):
    gen = lambda m, seed: np.clip(sift_like(m, 960, seed, n_clusters=80, rank=24, sigma_sub=40.0, sigma_iso=1.5, integer=False) / 255.0, 0, 1).astype(np.float32)
    _fp32_case(tmp_path, gen(20000, 123), gen(200, 456), 960, (64, 384), 0.9)


def test_deep_like_d96(tmp_path):
    def gen(m, seed):
        # the 10M run uses 32768 components (~300 rows each); the same density at this size
        x = sift_like(m, 96, seed, n_clusters=200, rank=12, sigma_sub=40.0, sigma_iso=4.0, integer=False, centre_lo=-60, centre_hi=60)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    _fp32_case(tmp_path, gen(60000, 123), gen(500, 456), 96, (32, 128, 256), 0.9)


def test_cohere_like_d768_ip_slimq(tmp_path):
    hs, O = load_product(), Oracle()
    def gen(m, seed):
        x = sift_like(m, 768, seed, n_clusters=16, rank=24, sigma_sub=40.0, sigma_iso=1.5, integer=False, centre_lo=-40, centre_hi=40)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    base, q = gen(20000, 123), gen(200, 456)
    hp, sp, qp = (str(tmp_path / f) for f in ("h.bin", "s.bin", "q.bin"))
    hs.build_hnsw(base, hp, metric=1, M=16, ef_construction=200, threads=8)
    hs.convert_slim(hp, sp, 768, metric=1, threads=8)
    rng = np.random.default_rng(0)
    cen = base[rng.choice(len(base), 16, replace=False)].copy()
    for _ in range(4):
        a = (-(base @ cen.T)).argmin(1)
        for c in range(16):
            if (a == c).any():
                cen[c] = base[a == c].mean(0)
    hs.convert_slimq(sp, 1, 768, cen, qp, threads=8)
    ix, ox = hs.Index(qp, hs.HS_KIND_SLIMQ, 768, metric=1), O.load_slimq(qp)
    ix.slimq_set_dataset(base)
    gt = np.argsort(-(q @ base.T), axis=1)[:, :10]
    best = 0.0
    for ef in (64, 512):
        ix.set_ef(ef); ox.set(ef, ix.slimq_tconst(), base)
        g, o = ix.slimq_search(q, 10, want_stats=True), ox.search(q, 10, threads=8)
        assert np.array_equal(g["labels"], o["labels"]) and np.array_equal(g["stats"].astype(np.uint64), o["counters"])
        best = max(best, _recall(g["labels"], gt))
    assert best >= 0.8, best
