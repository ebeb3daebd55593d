#!/usr/bin/env python3
"""CPU (oracle only): where does HNSW-SlimQ's recall on the COHERE-like rows go?  For each way of building the graph
     rq      rabitqlib HNSW (M=32, efC=128) + SlimQ's own PruneByHeuristic as written (hnswalg_slimq.h:1334-1362)  = the reference's pipeline
     rqslim  the same base graph + Slim's heuristic (hnswalg_slim.h:836-865, what :1349 presumably meant)
     hnswlib hnswlib HNSW (M=16, efC=200) + Slim's heuristic                                                        = rounds 1-2
recall@10 of (a) the fp32 search on the base graph, (b) the fp32 Slim search on the pruned graph, (c) the SlimQ search on it.
usage: slimq_graph_recall.py [n] rq,rqslim,hnswlib   (tmp files under $TMPDIR)"""
import os, sys, tempfile, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
TMP = tempfile.mkdtemp()
from hsutil import sift_like, load_product, Oracle
hs = load_product()
n, d, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, 768, 300
def gen(m, seed):
    G = [float(v) for v in os.environ.get("GEN", "256,24,40,1.5,40").split(",")]
    x = sift_like(m, d, seed, n_clusters=int(G[0]), rank=int(G[1]), sigma_sub=G[2], sigma_iso=G[3], integer=False, centre_lo=-G[4], centre_hi=G[4])
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
base = gen(n, 123); q = gen(nq, 456)
o = Oracle()
gt = o.brute_force(1, base, q, 10)
def rec(L): return sum(len(set(map(int, L[i])) & set(map(int, gt[i]))) for i in range(nq)) / (10 * nq)
rng = np.random.default_rng(0); cen = base[rng.choice(n, 16, replace=False)].copy()
for graph in sys.argv[2].split(","):
    hp, sp, qp = (os.path.join(TMP, f"c{graph}_{x}.bin") for x in "hsq")
    t0 = time.time()
    if graph == "rq":
        hs.build_rabitq_hnsw(base, hp, metric=1, M=32, ef_construction=128, threads=8)
        hs.convert_slimq_graph(hp, sp, d, metric=1, threads=8)
    elif graph == "rqslim":   # rabitqlib graph + Slim's (correct) heuristic
        hs.build_rabitq_hnsw(base, hp, metric=1, M=32, ef_construction=128, threads=8)
        hs.convert_slim(hp, sp, d, metric=1, threads=8)
    else:
        hs.build_hnsw(base, hp, metric=1, M=16, ef_construction=200, threads=8)
        hs.convert_slim(hp, sp, d, metric=1, threads=8)
    hs.convert_slimq(sp, 1, d, cen, qp, threads=8)
    print(graph, f"built in {time.time()-t0:.0f}s", flush=True)
    ov = o.load(hp, "hnsw", 1, d)
    os_ = o.load(sp, "slim", 1, d)
    oq = o.load_slimq(qp)
    for ef in (64, 256, 1024):
        ov.set_ef(ef); os_.set_ef(ef); oq.set(ef, hs.rabitq_default_tconst(768), base)
        rv = ov.search_pq(q, 10, threads=8); rs = os_.search_ids(q, 10, threads=8); rq = oq.search(q, 10, threads=8)
        # search_pq labels: farthest first, padded; use sets
        print(graph, "ef", ef, "base-graph fp32", round(rec(rv["labels"]), 4), "| slim fp32", round(rec(rs["labels"]), 4), "| slimq", round(rec(rq["labels"]), 4), flush=True)
