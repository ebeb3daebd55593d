#!/usr/bin/env python3
"""Diagnostic: per-query wall clock and candidate-heap replays of the flat kernel (make -C hnsw-slim_amd flatdiag).
usage: flat_diag.py [index-dir] [ef,ef,...]   (index-dir as written by bench.py --index-dir)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HS_LIB"] = os.path.join(ROOT, "hnsw-slim_amd", "libhnsw_slim_amd_flatdiag.so")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hsutil import headline_data, load_product  # noqa: E402

hs = load_product()
idir = sys.argv[1] if len(sys.argv) > 1 else "/tmp/hsidx"
efs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [48, 68, 128]
NQ, K, D = int(os.environ.get("NQ", "10000")), 10, 128
dev = torch.device("cuda", 0)
if not os.path.exists(os.path.join(idir, "slim.bin")):   # same files as bench.py --index-dir
    os.makedirs(idir, exist_ok=True)
    base = headline_data(int(os.environ.get("N", "1000000")), D, 123)
    hs.build_hnsw(base, os.path.join(idir, "hnsw.bin"), M=16, ef_construction=200, branching_factor="4", seed=100, threads=min(len(os.sched_getaffinity(0)), 64))
    hs.convert_slim(os.path.join(idir, "hnsw.bin"), os.path.join(idir, "slim.bin"), D, threads=min(len(os.sched_getaffinity(0)), 64))
    np.save(os.path.join(idir, "base.npy"), base)
    open(os.path.join(idir, "ready"), "w").write("ok")
ix = hs.Index(os.path.join(idir, "slim.bin"), hs.HS_KIND_SLIM, D)
q_t = torch.from_numpy(headline_data(NQ, D, 456)).to(dev)
lab = torch.empty((NQ, K), dtype=torch.int32, device=dev)
cnt = torch.empty((NQ,), dtype=torch.int32, device=dev)
stats = torch.zeros((5 * NQ, 4), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for ef in efs:
    ix.set_ef(ef)
    for _ in range(2):
        ix.search_ids_dev(q_t, K, lab, None, cnt, stats, st)
        ix.check(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ix.search_ids_dev(q_t, K, lab, None, cnt, stats, st)
    e1.record()
    torch.cuda.synchronize()
    h = stats.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    s4, dg = h[:NQ], h[NQ:].reshape(NQ, 16)
    wall = dg[:, 0] / 100.0   # us
    hops, rep = s4[:, 1], dg[:, 2]
    c_tie, c_ghost, c_end = dg[:, 1] & 0xFF, (dg[:, 1] >> 8) & 0xFF, dg[:, 1] >> 16
    syncs = c_tie + c_ghost + c_end
    ovf, t2, n2 = dg[:, 3] & 0xFFFF, (dg[:, 3] >> 16) & 1, dg[:, 3] >> 17
    print(f"ef={ef}: launch {e0.elapsed_time(e1):.3f} ms; query wall us: mean {wall.mean():.0f} p50 {np.median(wall):.0f} p99 {np.percentile(wall, 99):.0f} max {wall.max():.0f}; "
          f"hops mean {hops.mean():.0f} max {hops.max()}; us/hop (no-sync queries) {wall[syncs == 0].sum() / max(hops[syncs == 0].sum(), 1):.2f}")
    print(f"   queries with a heap sync: {(syncs > 0).mean() * 100:.1f}% (syncs/query among them {syncs[syncs > 0].mean() if (syncs > 0).any() else 0:.1f}, "
          f"replayed hops/query {rep[syncs > 0].mean() if (syncs > 0).any() else 0:.0f} of {hops[syncs > 0].mean() if (syncs > 0).any() else 0:.0f}); "
          f"us/hop among them {wall[syncs > 0].sum() / max(hops[syncs > 0].sum(), 1):.2f}; other passes: {(s4[:, 3] > 1).sum()} replays-at-k: {(s4[:, 3] == 1).sum()}")
    print(f"   sync causes (queries with >= 1): tie among unexpanded entries {(c_tie > 0).mean() * 100:.1f}%, ghost at the bound mid-search {(c_ghost > 0).mean() * 100:.1f}%, ghost at termination {(c_end > 0).mean() * 100:.1f}%")
    print(f"   visited set: overflow list used by {(ovf > 0).mean() * 100:.1f}% (max {ovf.max()}), tier 2 by {t2.sum()} queries (max ids {n2.max()})")
    t0 = dg[:, 4].astype(np.int64)
    t0 = (t0 - t0.min()) & 0xFFFFFFFF
    end = t0 / 100.0 + wall
    sync_us = dg[:, 5] / 100.0
    print(f"   replay: {sync_us[syncs > 0].mean() if (syncs > 0).any() else 0:.0f} us per synced query = {sync_us.sum() / max(rep.sum(), 1):.2f} us per replayed hop; kernel span {end.max():.0f} us, "
          f"last starts at {t0.max() / 100.0:.0f} us; the 5 last to finish: " + "; ".join(f"start {t0[i] / 100.0:.0f} wall {wall[i]:.0f} (replay {sync_us[i]:.0f}) hops {hops[i]}" for i in np.argsort(-end)[:5]))
    ph = dg[:, 8:14].astype(np.float64)
    l0 = np.maximum(hops - s4[:, 1].min() * 0, 1)
    names = ["select", "tile wait", "visited+compaction", "rows+distances", "accept(+pre-select)", "hop end"]
    print("   shader cycles per hop (level-0 hops ~ all hops; the stamps themselves cost ~10 %): " + ", ".join(f"{n} {ph[:, i].sum() / hops.sum():.0f}" for i, n in enumerate(names))
          + f"; sum {ph.sum() / hops.sum():.0f}; hops pre-selected {dg[:, 6].sum() / hops.sum() * 100:.0f}%")
    ns = syncs == 0
    for lo, hi in ((0, 0), (1, 8), (9, 32), (33, 64)):
        m = ns & (ovf >= lo) & (ovf <= hi)
        if m.any():
            print(f"   no-sync queries with {lo}..{hi} ids in the overflow list: {m.sum()} queries, {wall[m].sum() / hops[m].sum():.2f} us/hop, hops mean {hops[m].mean():.0f}, n_dist/hop {s4[m, 0].sum() / hops[m].sum():.2f}")
    print("   last to finish, detail: " + "; ".join(f"[start {t0[i] / 100.0:.0f} wall {wall[i]:.0f} replay {sync_us[i]:.0f} hops {hops[i]} n_dist {s4[i, 0]} ovf {ovf[i]} t2 {n2[i]}]" for i in np.argsort(-end)[:6]))
    top = np.argsort(-wall)[:8]
    print("   slowest: " + "; ".join(f"{wall[i]:.0f}us hops={hops[i]} syncs={syncs[i]} rep={rep[i]} ovf={ovf[i]} t2={n2[i]}" for i in top), flush=True)
