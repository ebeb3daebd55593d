"""Step-by-step SlimQ smoke run with prints (debug aid)."""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from hsutil import load_product, Oracle, sift_like

P = load_product(); O = Oracle()
n, d, nq, k = 3000, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 10
x = sift_like(n + nq, d, seed=5, n_clusters=32)
base, q = x[:n], x[n:]
tmp = tempfile.mkdtemp()
h, s, sq = (os.path.join(tmp, f) for f in ("h.bin", "s.bin", "sq.bin"))
P.build_hnsw(base, h, M=16, ef_construction=100, threads=8); P.convert_slim(h, s, d, threads=8)
P.convert_slimq(s, 0, d, base[:8].copy(), sq, threads=8)
print("files built", flush=True)
ix = P.Index(sq, P.HS_KIND_SLIMQ, d); print("loaded", ix.info(), flush=True)
ix.slimq_set_dataset(base); print("dataset set, t_const", ix.slimq_tconst(), flush=True)
ox = O.load_slimq(sq)
for ef in (10, 100):
    ix.set_ef(ef); ox.set(ef, ix.slimq_tconst(), base)
    print("searching ef", ef, flush=True)
    got = ix.slimq_search(q, k, want_stats=True)
    ref = ox.search(q, k)
    print("stats gpu", got["stats"][:4].tolist(), "ref", ref["counters"][:4].tolist())
    print("labels eq", np.array_equal(got["labels"], ref["labels"]), "dists eq", np.array_equal(got["dists"].view(np.uint32), ref["dists"].view(np.uint32)), "cnt", got["cnt"][:4], ref["counts"][:4])
    if not np.array_equal(got["labels"], ref["labels"]):
        print(got["labels"][0], ref["labels"][0]); print(got["dists"][0], ref["dists"][0])
