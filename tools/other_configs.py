#!/usr/bin/env python3
"""Parity-at-size + throughput on the other BASELINE.json shapes (not the bench line): GIST-like d=960 (configs[2]) and
DEEP-like d=96 (configs[3]).  Usage: other_configs.py gist|deep [n]     env GEN="n_clusters,rank,sigma_sub,sigma_iso" overrides the
generator (calibration runs), EFS the sweep.  Prints the sweep and, for the smallest ef with recall@10 >= 0.95, the operating point."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("DIAG_EF"):   # phase stamps of the flat kernel (make -C hnsw-slim_amd flatdiag) at that ef instead of the sweep
    os.environ["HS_LIB"] = os.path.join(ROOT, "hnsw-slim_amd", "libhnsw_slim_amd_flatdiag.so")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from hsutil import sift_like, load_product, Oracle
from bench import ground_truth, recall_at_k
hs = load_product()
which = sys.argv[1]
if which == "gist":
    n, d, nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000, 960, int(os.environ.get("NQ", "1000"))   # configs[2]: batches of 1k
    G = [float(x) for x in os.environ.get("GEN", "4096,24,40,1.5").split(",")]   # calibrated in round 2 (profiles/r02_cfg_gist1m_d960.log)
    gen = lambda m, seed: np.clip(sift_like(m, d, seed, n_clusters=int(G[0]), rank=int(G[1]), sigma_sub=G[2], sigma_iso=G[3], integer=False) / 255.0, 0, 1).astype(np.float32)
else:
    n, d, nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000, int(os.environ.get("DIM", "96")), 10000
    G = [float(x) for x in os.environ.get("GEN", "32768,12,40,4").split(",")]   # calibrated in round 2 (profiles/r02_cfg_deep10m_d96.log)
    def gen(m, seed):
        x = sift_like(m, d, seed, n_clusters=int(G[0]), rank=int(G[1]), sigma_sub=G[2], sigma_iso=G[3], integer=False, centre_lo=-60, centre_hi=60)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
# IDX_DIR: keep the data and index files there (a second, profiled run in the same gpurun call then skips the build);
# PROFILE_EF: only the search at that ef, a few launches (what runs under rocprofv3)
idir = os.environ.get("IDX_DIR") or tempfile.mkdtemp()
os.makedirs(idir, exist_ok=True)
hp, sp, bp, qp_ = (os.path.join(idir, f) for f in ("h.bin", "s.bin", "base.npy", "q.npy"))
dev = torch.device("cuda", 0)
if os.path.exists(sp) and os.path.exists(bp):
    base, q = np.load(bp, mmap_mode="r"), np.load(qp_)
else:
    t0 = time.time(); base = gen(n, 123); q = gen(nq, 456); print(f"{which}: generated n={n} d={d} in {time.time()-t0:.0f}s", flush=True)
    thr = min(len(os.sched_getaffinity(0)), 64)
    t0 = time.time(); hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=thr); tb = time.time() - t0
    t0 = time.time(); hs.convert_slim(hp, sp, d, threads=thr); tc = time.time() - t0
    print(f"build {tb:.0f}s convert {tc:.0f}s", flush=True)
    if os.environ.get("IDX_DIR"):
        np.save(bp, base); np.save(qp_, q)
ix = hs.Index(sp, hs.HS_KIND_SLIM, d)
if os.environ.get("PROFILE_EF"):
    ef = int(os.environ["PROFILE_EF"])
    qt = torch.from_numpy(q).to(dev)
    lab = torch.empty((nq, 10), dtype=torch.int32, device=dev); cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ix.set_ef(ef)
    for _ in range(12):
        ix.search_ids_dev(qt, 10, lab, None, cnt, None, s); ix.check(s)
    print(f"profile run: {which} ef={ef}, 12 launches of {nq} queries, kernel {ix.last_kernel()}", flush=True)
    sys.exit(0)
if os.environ.get("DIAG_EF"):
    qt = torch.from_numpy(q).to(dev)
    lab = torch.empty((nq, 10), dtype=torch.int32, device=dev); cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    stats = torch.zeros((5 * nq, 4), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for ef in [int(e) for e in os.environ["DIAG_EF"].split(",")]:
        ix.set_ef(ef)
        for _ in range(3):
            ix.search_ids_dev(qt, 10, lab, None, cnt, stats, s); ix.check(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ix.search_ids_dev(qt, 10, lab, None, cnt, stats, s); e1.record(); torch.cuda.synchronize()
        h = stats.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        s4, dg = h[:nq], h[nq:].reshape(nq, 16)
        wall, hops = dg[:, 0] / 100.0, s4[:, 1]
        ph = dg[:, 8:14].astype(np.float64)
        names = ["select", "tile wait", "visited+compaction", "rows+distances", "accept(+pre-select)", "hop end"]
        print(f"DIAG {which} ef={ef} [{ix.last_kernel()}]: launch {e0.elapsed_time(e1):.3f} ms; query wall us mean {wall.mean():.0f} max {wall.max():.0f}; hops mean {hops.mean():.0f} max {hops.max()}; "
              f"us/hop {wall.sum() / hops.sum():.2f}; n_dist/hop {s4[:, 0].sum() / hops.sum():.2f}; pre-selected {dg[:, 6].sum() / hops.sum() * 100:.0f}%")
        print("   shader cycles per hop: " + ", ".join(f"{nm} {ph[:, i].sum() / hops.sum():.0f}" for i, nm in enumerate(names)) + f"; sum {ph.sum() / hops.sum():.0f}", flush=True)
    sys.exit(0)
ox = Oracle().load(sp, "slim", 0, d)
bt = torch.from_numpy(np.ascontiguousarray(base)).to(dev); qt = torch.from_numpy(q).to(dev)
gt = ground_truth(torch, bt, qt, 10, hs); del bt
lab = torch.empty((nq, 10), dtype=torch.int32, device=dev); cnt = torch.empty((nq,), dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
op = None
for ef in [int(e) for e in os.environ.get('EFS', '32,64,128,192,256,384,512').split(',')]:
    ix.set_ef(ef); ox.set_ef(ef)
    for _ in range(2):
        ix.search_ids_dev(qt, 10, lab, None, cnt, st, s); ix.check(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ix.search_ids_dev(qt, 10, lab, None, cnt, None, s)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    L = lab.cpu().numpy().astype(np.uint32); S = st.cpu().numpy().astype(np.int64)
    want = ox.search_ids(q[:200], 10, threads=32)
    same = bool(np.array_equal(np.sort(L[:200], axis=1), np.sort(want["labels"], axis=1))) and bool(np.array_equal(S[:200, :3], want["counters"][:, :3]))
    by = (S[:, 0] * 4 * d + S[:, 2] * 4 + S[:, 1] * 8)
    print(f"ef={ef} [{ix.last_kernel()}]: recall@10={recall_at_k(L, gt):.4f} qps={nq/ms*1e3:.0f} ms={ms:.3f} n_dist={S[:,0].mean():.0f} hops={S[:,1].mean():.0f} alg_GB/s={by.sum()/ms/1e6:.0f} "
          f"frac={by.sum()/ms/1e6/8000:.3f} passes(tie/overflow)={(S[:,3]==1).sum()}/{(S[:,3]==2).sum()} oracle_match_first200={same}", flush=True)
    if op is None and recall_at_k(L, gt) >= 0.95:
        op = (ef, recall_at_k(L, gt), nq / ms * 1e3, by.sum() / ms / 1e6)
if op:
    # the same operating point with 16 launches in flight on 16 streams (device-resident queries; what a serving loop sees: a
    # launch smaller than the wave slots lasts as long as its longest query, several of them fill the chip)
    ix.set_ef(op[0])
    NS = 16
    streams = [torch.cuda.Stream() for _ in range(NS)]
    outs = [(torch.empty_like(lab), torch.empty_like(cnt)) for _ in range(NS)]
    for st_, o_ in zip(streams, outs):
        ix.search_ids_dev(qt, 10, o_[0], None, o_[1], None, st_.cuda_stream)
    torch.cuda.synchronize(); t0 = time.time()
    NB = 8 * NS
    for r_ in range(NB):
        st_, o_ = streams[r_ % NS], outs[r_ % NS]
        ix.search_ids_dev(qt, 10, o_[0], None, o_[1], None, st_.cuda_stream)
    torch.cuda.synchronize(); msp = (time.time() - t0) * 1e3 / NB
    for st_ in streams: ix.check(st_.cuda_stream)
    ix.search_ids_dev(qt, 10, lab, None, cnt, None, s); ix.check(s)
    same_p = all(bool(torch.equal(o_[0], lab)) for o_ in outs)
    print(f"PIPELINED {which} ef={op[0]}: {NS} launches of {nq} queries in flight: {nq/msp*1e3:.0f} q/s = {op[3] * ((nq/msp*1e3)/op[2]):.0f} alg GB/s = frac {op[3] * ((nq/msp*1e3)/op[2]) / 8000:.3f}; labels of all {NS} streams equal the single launch's: {same_p}", flush=True)
    open(os.path.join(idir, "operating_ef"), "w").write(str(op[0]))
    print(f"OPERATING POINT {which} n={n} d={d} nq={nq}: ef={op[0]} recall@10={op[1]:.4f} single-launch qps={op[2]:.0f} alg GB/s={op[3]:.0f} frac={op[3]/8000:.3f}", flush=True)
else:
    print(f"OPERATING POINT {which}: recall@10 >= 0.95 not reached inside the sweep", flush=True)
