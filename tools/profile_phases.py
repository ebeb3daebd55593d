#!/usr/bin/env python3
"""Diagnostic: where does a query spend its cycles?  Runs the HS_PROFILE build (make -C hnsw-slim_amd prof)
on the bench workload and prints the per-phase share of shader cycles (averaged over queries).
Never quote this build's run time (stamps serialise the wave); read the shares."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HS_LIB"] = os.path.join(ROOT, "hnsw-slim_amd", "libhnsw_slim_amd_prof.so")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hsutil import headline_data, load_product  # noqa: E402

hs = load_product()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
efs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128]
NQ, K, D = int(os.environ.get("NQ", "10000")), 10, 128
dev = torch.device("cuda", 0)
base = headline_data(N, D, 123)
q = headline_data(NQ, D, 456)
with tempfile.TemporaryDirectory() as tmp:
    hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
    t0 = time.time()
    hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=64)
    hs.convert_slim(hp, sp, D, threads=64)
    print(f"build+convert {time.time()-t0:.1f}s", flush=True)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, D)
q_t = torch.from_numpy(q).to(dev)
lab = torch.empty((NQ, K), dtype=torch.int32, device=dev)
cnt = torch.empty((NQ,), dtype=torch.int32, device=dev)
stats = torch.zeros((NQ * 4 + NQ * 16,), dtype=torch.int32, device=dev)  # nq x 4 u32 + nq x 8 u64
names = ["pop", "adjacency", "visited", "distances", "accept", "upper", "wall100MHz", "final"]
for ef in efs:
    ix.set_ef(ef)
    for _ in range(2):
        ix.search_ids_dev(q_t, K, lab, None, cnt, stats, torch.cuda.current_stream().cuda_stream)
        ix.check(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    h = stats.cpu().numpy()
    st = h[: NQ * 4].reshape(NQ, 4)
    ph = h[NQ * 4:].view(np.uint64).reshape(NQ, 8).astype(np.float64)
    wall = ph[:, 6].copy()
    start = ph[:, 7].copy()
    ph[:, 6] = 0
    ph[:, 7] = 0
    tot = ph.sum(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ix.search_ids_dev(q_t, K, lab, None, cnt, stats, torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    print(f"   batch wall {e0.elapsed_time(e1):.3f} ms; per-query wall (100MHz ticks) mean={wall.mean():.0f} -> {wall.mean()/100:.1f} us; "
          f"implied shader clock {tot.mean()/max(wall.mean(),1)*100:.0f} MHz; sum(query wall)/batch wall = concurrency {wall.sum()/100/1e3/e0.elapsed_time(e1):.0f}")
    ok = start > 0
    start, wall_ok = start[ok], wall[ok]
    t0 = start.min()
    end = start + wall_ok
    T = end.max() - t0
    samples = [int(((start <= t0 + f * T) & (end > t0 + f * T)).sum()) for f in (0.02, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.98)]
    print(f"   kernel span {T/100:.0f} us; resident queries at 2/10/20/.../90/98% of span: {samples}; passes: fast={int((st[:,3]==0).sum())} tie-rerun={int((st[:,3]==1).sum())} overflow-rerun={int((st[:,3]==2).sum())}; "
          f"query wall us p50={np.median(wall)/100:.0f} p99={np.percentile(wall,99)/100:.0f} max={wall.max()/100:.0f}")
    print(f"ef={ef}: n_dist={st[:,0].mean():.0f} hops={st[:,1].mean():.0f} nbr={st[:,2].mean():.0f} fallback={st[:,3].sum()} "
          f"cycles/query mean={tot.mean():.0f} p50={np.median(tot):.0f} max={tot.max():.0f}")
    print("   " + "  ".join(f"{n}={ph[:,i].mean():.0f} ({100*ph[:,i].sum()/tot.sum():.1f}%)" for i, n in enumerate(names)))
    print(f"   per hop: " + "  ".join(f"{n}={ph[:,i].sum()/st[:,1].sum():.0f}" for i, n in enumerate(names[:5])), flush=True)
