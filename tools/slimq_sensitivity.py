#!/usr/bin/env python3
"""Diagnostic (CPU, oracle only): how far does a last-place wobble of the HNSW-SlimQ query factors move the answer?

The reference computes delta / vl / k1xsumq (rabitqlib/index/query.hpp:127-156) and q_to_centroids (hnswalg_slimq.h:1823-1848)
through Eigen reductions whose summation order depends on alignment and build; this repository DEFINES them as left-to-right fp32
sums (agreeing with the compiled rabitqlib to a few units in the last place: tests/test_oracle_golden.py).  This script moves each
quantity by +-1 / +-2 ulp in the oracle (hso_slimq_perturb) and counts the queries whose returned id set changes.
usage: slimq_sensitivity.py [sift|cohere] [n] [nq]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from hsutil import Oracle, headline_data, load_product, sift_like  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sift"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
hs = load_product()
if which == "sift":
    d, metric = 128, 0
    gen = lambda m, seed: headline_data(m, d, seed)
else:
    d, metric = 768, 1

    def gen(m, seed):
        x = sift_like(m, d, seed, n_clusters=256, rank=24, sigma_sub=40.0, sigma_iso=1.5, integer=False, centre_lo=-40.0, centre_hi=40.0)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
base, q = gen(n, 123), gen(nq, 456)
rng = np.random.default_rng(0)
cen = base[rng.choice(n, 16, replace=False)].copy()
tmp = tempfile.mkdtemp()
hp, sp, qp = (os.path.join(tmp, f) for f in ("h.bin", "s.bin", "q.bin"))
t0 = time.time()
hs.build_hnsw(base, hp, metric=metric, M=16, ef_construction=200, threads=8)
hs.convert_slim(hp, sp, d, metric=metric, threads=8)
hs.convert_slimq(sp, metric, d, cen, qp, threads=8)
ox = Oracle().load_slimq(qp)
t_const = hs.rabitq_default_tconst(ox.padded)
print(f"{which}: n={n} d={d} metric={'IP' if metric else 'L2'} nq={nq}, index built in {time.time() - t0:.0f}s, t_const={t_const:.4f}", flush=True)
names = ["delta", "vl", "k1xsumq", "q_to_centroids"]
for ef in (64, 256):
    ox.set(ef, t_const, base)
    ox.perturb([0, 0, 0, 0])
    ref = ox.search(q, 10, threads=8)
    ref_sets = np.sort(ref["labels"], axis=1)
    rows = []
    for i, nm in enumerate(names):
        for u in (+1, -1, +2, -2):
            p = [0, 0, 0, 0]
            p[i] = u
            ox.perturb(p)
            got = np.sort(ox.search(q, 10, threads=8)["labels"], axis=1)
            rows.append((f"{nm} {u:+d} ulp", int((got != ref_sets).any(axis=1).sum())))
    for u in (1, 2):
        worst = 0
        for trial in range(4):
            p = [int(s) * u for s in rng.choice([-1, 1], 4)]
            ox.perturb(p)
            got = np.sort(ox.search(q, 10, threads=8)["labels"], axis=1)
            worst = max(worst, int((got != ref_sets).any(axis=1).sum()))
        rows.append((f"all four, random signs, {u} ulp (worst of 4 draws)", worst))
    ox.perturb([0, 0, 0, 0])
    print(f"ef={ef}: queries (of {nq}) whose returned id set changes: " + "; ".join(f"{k}: {v}" for k, v in rows), flush=True)
