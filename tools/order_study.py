#!/usr/bin/env python3
"""Diagnostic (CPU): which cheap per-query number predicts the length of the level-0 search (hops), i.e. a good longest-first
start order for one launch?  Builds a headline-like index, takes the oracle's hop counts, evaluates predictors available after
the descent (entry distance) or after ONE level-0 expansion, by rank correlation and by the list-scheduling makespan of a launch
(queries started in predictor order on `slots` wave slots, duration = hops).   usage: order_study.py [N] [NQ] [ef] [slots]"""
import heapq
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from hsutil import Oracle, headline_data, load_chal_encode, load_product  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
EF = int(sys.argv[3]) if len(sys.argv) > 3 else 70
SLOTS = int(sys.argv[4]) if len(sys.argv) > 4 else 5120
D, K = 128, 10
hs = load_product()
base = headline_data(N, D, 123)
q = headline_data(NQ, D, 456)
tmp = tempfile.mkdtemp()
hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=8)
hs.convert_slim(hp, sp, D, threads=8)
g = load_chal_encode().parse_slim(open(sp, "rb").read(), D)
adj0 = [l[0] for l in g["lists"]]
ox = Oracle().load(sp, "slim", 0, D)
ox.set_ef(EF)
res = ox.search_ids(q, K, threads=8)
hops = res["counters"][:, 1].astype(np.float64)
entry = ox.entry(q)


def dist(qv, ids):
    x = base[ids] - qv
    return (x * x).sum(1)


feat = {"entry distance": np.zeros(NQ), "best after 1 expansion": np.zeros(NQ), "mean of 4 best after 1 expansion": np.zeros(NQ),
        "4th best after 1 expansion": np.zeros(NQ), "mean of 4 best after 2 expansions": np.zeros(NQ)}
for i in range(NQ):
    e = int(entry[i])
    d0 = float(dist(q[i], np.array([e]))[0])
    nb = adj0[e]
    ds = np.sort(np.concatenate([[d0], dist(q[i], nb)])) if len(nb) else np.array([d0])
    feat["entry distance"][i] = d0
    feat["best after 1 expansion"][i] = ds[0]
    feat["mean of 4 best after 1 expansion"][i] = ds[:4].mean()
    feat["4th best after 1 expansion"][i] = ds[min(3, len(ds) - 1)]
    # second expansion: the nearest neighbour of the entry
    if len(nb):
        dn = dist(q[i], nb)
        b = int(nb[int(np.argmin(dn))])
        nb2 = np.array([x for x in adj0[b] if x != e and x not in set(nb.tolist())], dtype=np.int64)
        ds2 = np.sort(np.concatenate([ds, dist(q[i], nb2)])) if len(nb2) else ds
    else:
        ds2 = ds
    feat["mean of 4 best after 2 expansions"][i] = ds2[:4].mean()


def makespan(order):
    free = [0.0] * SLOTS
    heapq.heapify(free)
    end = 0.0
    for i in order:
        t = heapq.heappop(free) + hops[i]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


def rank(x):
    r = np.empty(len(x))
    r[np.argsort(x)] = np.arange(len(x))
    return r


ideal = max(hops.sum() / SLOTS, hops.max())
print(f"N={N} NQ={NQ} ef={EF} slots={SLOTS}: hops mean {hops.mean():.1f} max {hops.max():.0f}; perfect balance {ideal:.0f}; "
      f"index order {makespan(range(NQ)) / ideal:.3f}x; exact longest-first {makespan(np.argsort(-hops)) / ideal:.3f}x")
for name, f in feat.items():
    rc = np.corrcoef(rank(f), rank(hops))[0, 1]
    print(f"  {name:38s}: rank correlation {rc:+.3f}; makespan farthest-first {makespan(np.argsort(-f)) / ideal:.3f}x, nearest-first {makespan(np.argsort(f)) / ideal:.3f}x")
