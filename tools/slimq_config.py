#!/usr/bin/env python3
"""HNSW-SlimQ (RaBitQ) at size: parity against the oracle on a query sample + throughput / recall sweep.
Usage: slimq_config.py sift|cohere [n]
  sift   : the bench's SIFT-1M-like data, d=128 L2  (same graph as the fp32 bench line, quantised)
  cohere : COHERE-1M-like, d=768 inner product on unit vectors (BASELINE.json configs[4])
env GRAPH=rq : build the base graph the way the reference's SlimQ strategy does (rabitqlib HNSW M=32/efC=128 + SlimQ's prune)
               instead of the fp32 bench line's hnswlib graph"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("DIAG_EF"):   # phase stamps of the SlimQ kernel (make -C hnsw-slim_amd slimqdiag) at those ef instead of the sweep
    os.environ["HS_LIB"] = os.path.join(ROOT, "hnsw-slim_amd", "libhnsw_slim_amd_slimqdiag.so")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from hsutil import sift_like, headline_data, load_product, Oracle
from bench import recall_at_k
hs = load_product()
which = sys.argv[1]
if which == "sift":
    n, d, nq, metric = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000, 128, 10000, 0
    gen = lambda m, seed: headline_data(m, d, seed)
else:
    n, d, nq, metric = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000, 768, 10000, 1
    def gen(m, seed):
        # round-2 calibration (profiles/r02_calib_cohere_local.log): 256 components of rank 24 keep both the pruned graph and the
        # 1-bit estimates usable; GEN = n_clusters,rank,sigma_sub,sigma_iso,centre_half_width
        G = [float(v) for v in os.environ.get("GEN", "256,24,40,1.5,40").split(",")]
        x = sift_like(m, d, seed, n_clusters=int(G[0]), rank=int(G[1]), sigma_sub=G[2], sigma_iso=G[3], integer=False, centre_lo=-G[4], centre_hi=G[4])
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
# IDX_DIR: keep data + index files (a profiled second run skips generation and build); PROFILE_EF: only the search at that ef
idir = os.environ.get("IDX_DIR") or tempfile.mkdtemp()
os.makedirs(idir, exist_ok=True)
hp, sp, qp, bp, qf = (os.path.join(idir, f) for f in ("h.bin", "s.bin", "q.bin", "base.npy", "queries.npy"))
dev = torch.device("cuda", 0)
profile_ef = int(os.environ.get("PROFILE_EF", "0"))
if os.path.exists(qp) and os.path.exists(bp):
    base, q = np.load(bp, mmap_mode="r"), np.load(qf)
    qt = torch.from_numpy(q).to(dev)
    gt = np.load(os.path.join(idir, "gt.npy")) if os.path.exists(os.path.join(idir, "gt.npy")) else None
else:
    t0 = time.time(); base = gen(n, 123); q = gen(nq, 456); print(f"{which}: generated n={n} d={d} in {time.time()-t0:.0f}s", flush=True)
    bt = torch.from_numpy(base).to(dev); qt = torch.from_numpy(q).to(dev)
    # exact ground truth on the GPU (plumbing): L2 or inner product
    out = []
    bn = (bt * bt).sum(1)
    for s0 in range(0, nq, 512):
        sc = qt[s0:s0 + 512] @ bt.T
        dist = bn[None, :] - 2.0 * sc if metric == 0 else -sc
        out.append(torch.topk(dist, 10, dim=1, largest=False).indices)
    gt = torch.cat(out).cpu().numpy(); del out, bn
    # 16 centroids by a few Lloyd rounds on a sample (the reference takes them from a k-means script: hnsw_slimq_strategy.h:101-106)
    rng = np.random.default_rng(0)
    samp = bt[torch.from_numpy(rng.choice(n, min(n, 200_000), replace=False)).to(dev)]
    cen = samp[:16].clone()
    for _ in range(8):
        a = torch.cdist(samp, cen).argmin(1)
        for c in range(16):
            m = a == c
            if m.any(): cen[c] = samp[m].mean(0)
    cen = cen.cpu().numpy(); del samp, bt
    thr = min(len(os.sched_getaffinity(0)), 64)
    if os.environ.get("GRAPH", "hnswlib") == "rq":
        # the reference's own SlimQ pipeline (hnsw_slimq_strategy.h:106-128): rabitqlib's HNSW (M=32, efC=128, seed 100) + SlimQ's own
        # PruneByHeuristic in convertFromHNSW (hnswalg_slimq.h:1334-1362)
        t0 = time.time(); hs.build_rabitq_hnsw(base, hp, metric=metric, M=32, ef_construction=128, seed=100, threads=thr); tb = time.time() - t0
        t0 = time.time(); hs.convert_slimq_graph(hp, sp, d, metric=metric, threads=thr); tc = time.time() - t0
        print("graph: rabitqlib-style HNSW M=32 efC=128 + SlimQ prune", flush=True)
    else:
        t0 = time.time(); hs.build_hnsw(base, hp, metric=metric, M=16, ef_construction=200, threads=thr); tb = time.time() - t0
        t0 = time.time(); hs.convert_slim(hp, sp, d, metric=metric, threads=thr); tc = time.time() - t0
        print("graph: hnswlib HNSW M=16 efC=200 + Slim prune", flush=True)
    t0 = time.time(); hs.convert_slimq(sp, metric, d, cen, qp, threads=thr); tq = time.time() - t0
    print(f"build {tb:.0f}s convert {tc:.0f}s quantise {tq:.0f}s  slim {os.path.getsize(sp)/1e6:.0f} MB slimq {os.path.getsize(qp)/1e6:.0f} MB", flush=True)
    if os.environ.get("IDX_DIR"):
        np.save(bp, base); np.save(qf, q); np.save(os.path.join(idir, "gt.npy"), gt)
ix = hs.Index(qp, hs.HS_KIND_SLIMQ, d, metric=metric)
if profile_ef:
    ix.slimq_set_dataset(np.ascontiguousarray(base))
    ix.set_ef(profile_ef)
    lab = torch.empty((nq, 10), dtype=torch.int64, device=dev); dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(12):
        ix.slimq_search_dev(qt, 10, lab, dd, cnt, None, s); ix.check(s)
    print(f"profile run: {which} ef={profile_ef}, 12 launches of {nq} queries, kernel {ix.last_kernel()}", flush=True)
    sys.exit(0)
if os.environ.get("DIAG_EF"):
    ix.slimq_set_dataset(np.ascontiguousarray(base))
    lab = torch.empty((nq, 10), dtype=torch.int64, device=dev); dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    cnt = torch.empty((nq,), dtype=torch.int32, device=dev); stats = torch.zeros((5 * nq, 4), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    names = ["pop+expanded-set", "tile wait", "estimates", "pre-test (set lookups)", "insertions", "re-rank flush"]
    for ef in [int(e) for e in os.environ["DIAG_EF"].split(",")]:
        ix.set_ef(ef)
        for _ in range(3):
            ix.slimq_search_dev(qt, 10, lab, dd, cnt, stats, s); ix.check(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ix.slimq_search_dev(qt, 10, lab, dd, cnt, stats, s); e1.record(); torch.cuda.synchronize()
        h = stats.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        s4, dg = h[:nq], h[nq:].reshape(nq, 16)
        wall, pops = dg[:, 0] / 100.0, s4[:, 0] + s4[:, 3]
        ph = dg[:, 8:14].astype(np.float64)
        print(f"DIAG {which} ef={ef}: launch {e0.elapsed_time(e1):.3f} ms; query wall us mean {wall.mean():.0f} max {wall.max():.0f}; pops (hops+revisits) mean {pops.mean():.0f} max {pops.max()}; "
              f"us/pop {wall.sum() / pops.sum():.2f}; est/hop {s4[:, 1].sum() / s4[:, 0].sum():.1f} ins/hop {s4[:, 2].sum() / s4[:, 0].sum():.2f}")
        print("   shader cycles per pop: " + ", ".join(f"{nm} {ph[:, i].sum() / pops.sum():.0f}" for i, nm in enumerate(names)) + f"; sum {ph.sum() / pops.sum():.0f}", flush=True)
    sys.exit(0)
ox = Oracle().load_slimq(qp)
base = np.ascontiguousarray(base)
ix.slimq_set_dataset(base)
ox.set(64, ix.slimq_tconst(), base)
gp, op = ix.slimq_prepare_debug(q, ox.padded, ox.ncl), ox.prepare(q)
for key in ("rq", "q3", "g_add"):
    neq = (gp[key].view(np.uint32) != op[key].view(np.uint32))
    print(f"prep {key}: {int(neq.any(axis=1).sum())} of {nq} queries differ, {int(neq.sum())} elements", flush=True)
    if neq.any():
        i, j = np.argwhere(neq)[0]; print("   first:", i, j, gp[key][i, j], op[key][i, j], hex(gp[key].view(np.uint32)[i, j]), hex(op[key].view(np.uint32)[i, j]))
print("prep planes equal:", bool(np.array_equal(gp["planes"], op["planes"])), flush=True)
print("device bytes", ix.info()["device_bytes"], "t_const", ix.slimq_tconst(), flush=True)
lab = torch.empty((nq, 10), dtype=torch.int64, device=dev); dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
cnt = torch.empty((nq,), dtype=torch.int32, device=dev); st = torch.empty((nq, 4), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
streams = [torch.cuda.Stream() for _ in range(4)]
outs = [(torch.empty_like(lab), torch.empty_like(dd), torch.empty_like(cnt)) for _ in range(4)]
rec_bytes = 16 + (d + 63) // 64 * 8
tile = max(16, (int(ix.info()["max_degree0"]) + 15) // 16 * 16)   # ids per adjacency tile (capi.cpp tile_stride_for)
print("adjacency tile", tile, "ids", flush=True)
for ef in [int(e) for e in os.environ.get("EFS", "64,128,256,512,1024").split(",")]:
    ix.set_ef(ef); ox.set(ef, ix.slimq_tconst(), base)
    for _ in range(2):
        ix.slimq_search_dev(qt, 10, lab, dd, cnt, st, s); ix.check(s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ix.slimq_search_dev(qt, 10, lab, dd, cnt, None, s)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    # pipelined over 4 HIP streams (as bench.py does for the fp32 path): hides the last-wave tail of each launch
    for st_, o_ in zip(streams, outs):
        ix.slimq_search_dev(qt, 10, o_[0], o_[1], o_[2], None, st_.cuda_stream)
    torch.cuda.synchronize(); t0 = time.time()
    for r_ in range(16):
        st_, o_ = streams[r_ % 4], outs[r_ % 4]
        ix.slimq_search_dev(qt, 10, o_[0], o_[1], o_[2], None, st_.cuda_stream)
    torch.cuda.synchronize(); ms4 = (time.time() - t0) * 1e3 / 16
    for st_ in streams: ix.check(st_.cuda_stream)
    L = lab.cpu().numpy().astype(np.uint64); S = st.cpu().numpy().astype(np.int64)
    t0 = time.time(); want = ox.search(q[:200], 10, threads=32); tcpu = time.time() - t0
    same = bool(np.array_equal(L[:200], want["labels"])) and bool(np.array_equal(S[:200], want["counters"].astype(np.int64)))
    if not same:
        bad = [i for i in range(200) if not (np.array_equal(L[i], want["labels"][i]) and np.array_equal(S[i], want["counters"][i].astype(np.int64)))]
        i = bad[0]
        print(f"  MISMATCH in {len(bad)} of 200 queries; first {i}: gpu stats {S[i].tolist()} oracle {want['counters'][i].tolist()}")
        print("   gpu labels", L[i].tolist(), "\n   ora labels", want["labels"][i].tolist())
        tg, sg = ix.slimq_trace(q[i:i + 1], 10, 16384)
        tg2, _ = ix.slimq_trace(q[i:i + 1], 10, 16384)
        to = ox.trace(q[i], 10, 16384); to2 = ox.trace(q[i], 10, 16384)
        tg = tg[0][:len(to) + 4]
        print("   gpu runs equal", bool(np.array_equal(tg, tg2[0][:len(tg)])), "oracle runs equal", bool(np.array_equal(to, to2)), "single-query gpu stats", sg[0].tolist())
        m = min(len(tg), len(to)); dv = np.nonzero(tg[:m] != to[:m])[0]
        print("   pops oracle", len(to), "first divergence at pop", (int(dv[0]) if len(dv) else None))
        if len(dv):
            j = int(dv[0]); print("   gpu", [hex(int(v)) for v in tg[max(0, j - 3):j + 5]], "\n   ora", [hex(int(v)) for v in to[max(0, j - 3):j + 5]])
            j &= ~1
            print("   gpu", [hex(int(v)) for v in tg[max(0, j - 6):j + 8]], "\n   ora", [hex(int(v)) for v in to[max(0, j - 6):j + 8]])
            cand = int(tg[j]) & 0x3FFFFFFF
            prev = [(k2, hex(int(to[k2 + 1]))) for k2 in range(0, j, 2) if int(to[k2]) == (cand | 0x40000000)]
            print("   earlier inserts of it (event, d bits) oracle:", prev, " gpu d bits now:", hex(int(tg[j + 1])))
            pops_before = [int(v) & 0x3FFFFFFF for v in to[:j] if not (int(v) & 0x40000000)]
            print("   extra candidate", cand, "popped before:", cand in pops_before, "times inserted before:", sum(1 for v in to[:j] if int(v) == (cand | 0x40000000)))
            import pickle; pickle.dump(dict(tg=tg, to=to, q=q[i]), open(os.path.join(ROOT, "gpurun_out", "slimq_diverge.pkl"), "wb"))
        print("   gpu dists", dd.cpu().numpy()[i].tolist(), "\n   ora dists", want["dists"][i].tolist(), flush=True)
    by = S[:, 1] * rec_bytes + S[:, 0] * (4 * d + 4 * tile)   # estimates x record + expansions x (raw row + adjacency tile)
    print(f"ef={ef}: recall@10={recall_at_k(L, gt):.4f} qps={nq/ms*1e3:.0f} ms={ms:.3f} qps_4streams={nq/ms4*1e3:.0f} hops={S[:,0].mean():.0f} est={S[:,1].mean():.0f} ins={S[:,2].mean():.0f} "
          f"revisit={S[:,3].mean():.0f} alg_GB/s={by.sum()/ms/1e6:.0f} oracle_match_first200={same} oracle_32thr_qps={200/tcpu:.0f}", flush=True)
