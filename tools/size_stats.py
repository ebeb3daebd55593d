#!/usr/bin/env python3
"""Distribution of per-query scratch needs on the bench workload (oracle counters): peak candidate-heap
size and number of visited ids, per ef.  Used to size the LDS tiers."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hsutil import headline_data, load_product, Oracle
hs = load_product()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
base = headline_data(N, 128, 123)
q = headline_data(10000, 128, 456)
with tempfile.TemporaryDirectory() as tmp:
    hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
    hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=64)
    hs.convert_slim(hp, sp, 128, threads=64)
    ox = Oracle().load(sp, "slim", 0, 128)
    for ef in (32, 64, 96, 128, 192, 256):
        ox.set_ef(ef)
        r = ox.search_ids(q, 10, threads=64)
        c = r["counters"].astype(np.int64)
        pc = lambda x: [int(np.percentile(x, p)) for p in (50, 90, 99, 99.9, 100)]
        print(f"ef={ef}: n_dist p50/90/99/99.9/max={pc(c[:,0])}  max_cand={pc(c[:,4])}  n_accept={pc(c[:,3])} hops={pc(c[:,1])}", flush=True)
