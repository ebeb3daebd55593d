#!/usr/bin/env python3
"""Diagnostic (CPU): how often does a heap-free level-0 search of HNSW-Slim (hnswalg_slim.h:321-457) need the reference's
candidate-heap LAYOUT at all?

Simulates the "flat" search (next node = nearest unexpanded entry of the result set) with ORDER-INDEPENDENCE rules:
  * a group of >= 2 unexpanded candidates at the same (minimal) distance d is expanded back to back in any order; the
    outcome is the same set whichever order the reference's heap pops them in, PROVIDED every neighbour evaluated inside
    the window has a key > d and no accept/evict decision inside the window happens at equality with the bound;
  * an entry evicted at exactly the new lowerBound stays expandable (a "ghost") while lowerBound does not move;
  * the final k-selection needs the reference's heap layout only if key[k-1] == key[k]; the insertion log is the
    reference's iff no window was taken.
Counts how many queries fall outside these rules (they need the candidate heap) and checks every other query's id set
and counters against the oracle (oracle/liboracle.so).   usage: tie_study.py [N] [NQ] [ef,ef,...]
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from hsutil import Oracle, headline_data, load_chal_encode, load_product  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
EFS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [70, 128]
K, D = 10, 128

hs = load_product()
base = headline_data(N, D, 123)
q = headline_data(NQ, D, 456)
tmp = tempfile.mkdtemp()
hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
t0 = time.time()
hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=8)
hs.convert_slim(hp, sp, D, threads=8)
print(f"build+convert {time.time()-t0:.1f}s", flush=True)
g = load_chal_encode().parse_slim(open(sp, "rb").read(), D)
adj0 = [l[0] for l in g["lists"]]
ox = Oracle().load(sp, "slim", 0, D)


def dist(qv, ids):
    x = base[ids] - qv
    return (x * x).sum(1).astype(np.float32)   # integer-valued data: exact in any order


def flat_search(qv, entry, ef):
    """returns dict(ids=top-k ids sorted, n_dist, n_hops, n_nbr (level 0 only), need_heap reason or None, stats...)"""
    visited = {int(entry)}
    d0 = float(dist(qv, np.array([entry]))[0])
    T = [[d0, int(entry), False]]          # sorted by key, then insertion order
    ghosts = []                             # unexpanded entries evicted at exactly lowerBound
    lb = d0
    n_dist = n_hops = n_nbr = 0
    windows = 0
    b_events = 0
    ghost_used = 0
    reason = None

    def expand(node, window_d):
        nonlocal lb, n_dist, n_hops, n_nbr, b_events, reason, ghosts
        n_hops += 1
        nb = adj0[node]
        n_nbr += len(nb)
        new = [int(x) for x in nb if int(x) not in visited]
        for x in new:
            visited.add(x)
        if not new:
            return
        ds = dist(qv, np.array(new))
        n_dist += len(new)
        for x, dx in zip(new, ds):
            dx = float(dx)
            if window_d is not None and dx <= window_d:
                reason = reason or "window: evaluated key <= d"
            full = len(T) >= ef
            if full and dx == lb and window_d is not None:
                reason = reason or "window: reject at equality"
            if (not full) or lb > dx:
                # insert after the entries with key <= dx
                pos = len(T)
                for i, e in enumerate(T):
                    if e[0] > dx:
                        pos = i
                        break
                T.insert(pos, [dx, x, False])
                if len(T) > ef:
                    ev = T.pop()
                    new_lb = T[-1][0]
                    if ev[0] == new_lb:
                        b_events += 1
                        if window_d is not None:
                            reason = reason or "window: eviction at equality"
                        if not ev[2]:
                            ghosts.append(ev)
                    if new_lb < lb:
                        ghosts = [e for e in ghosts if e[0] == new_lb]
                    lb = new_lb
                else:
                    lb = T[-1][0]

    while True:
        U = [e for e in T if not e[2]]
        Gh = [e for e in ghosts if e[0] == lb]
        cands = U + Gh
        if not cands:
            break
        d = min(e[0] for e in cands)
        if d > lb:
            break
        group = [e for e in cands if e[0] == d]
        if len(group) == 1:
            e = group[0]
            e[2] = True
            if e in ghosts:
                ghosts.remove(e)
                ghost_used += 1
            expand(e[1], None)
        else:
            windows += 1
            for e in group:
                e[2] = True
                if e in ghosts:
                    ghosts.remove(e)
                    ghost_used += 1
            for e in group:
                expand(e[1], d)
            if reason:
                break
    ktie = len(T) > K and T[K - 1][0] == T[K][0]
    if reason is None and ktie and windows > 0:
        reason = "k-boundary tie after a window"
    return dict(ids=sorted(e[1] for e in T[:K]), n_dist=n_dist, n_hops=n_hops, n_nbr=n_nbr, reason=reason, windows=windows,
                b_events=b_events, ghost_used=ghost_used, ktie=ktie)


for ef in EFS:
    ox.set_ef(ef)
    want = ox.search_ids(q, K, threads=8)
    entry = ox.entry(q)
    # counters of the oracle include the descent: recover the level-0 part by differences against a search from the entry
    reasons = {}
    bad = 0
    n_win = n_b = n_gh = n_kt = 0
    checked = 0
    t0 = time.time()
    lvl0 = []
    for i in range(NQ):
        r = flat_search(q[i], entry[i], ef)
        n_win += r["windows"] > 0
        n_b += r["b_events"] > 0
        n_gh += r["ghost_used"] > 0
        n_kt += r["ktie"]
        if r["reason"]:
            reasons[r["reason"]] = reasons.get(r["reason"], 0) + 1
            continue
        if r["ktie"]:
            continue   # exact log replay decides (not simulated here)
        checked += 1
        ok = r["ids"] == sorted(int(x) for x in want["labels"][i])
        lvl0.append((int(want["counters"][i, 0]) - r["n_dist"], int(want["counters"][i, 1]) - r["n_hops"], int(want["counters"][i, 2]) - r["n_nbr"]))
        bad += not ok
    lv = np.array(lvl0)
    # the descent's share of the counters must be consistent: n_nbr_desc == n_dist_desc - 1
    cons = int(np.sum(lv[:, 2] != lv[:, 0] - 1)) if len(lv) else 0
    print(f"ef={ef}: {NQ} queries in {time.time()-t0:.0f}s: tie windows in {n_win}, evict-at-bound events in {n_b}, ghost expanded in {n_gh}, "
          f"k-boundary tie in {n_kt}; NEED HEAP: {sum(reasons.values())} {reasons}; checked {checked}: id-set mismatches {bad}, "
          f"counter inconsistencies {cons}", flush=True)
