#!/usr/bin/env python3
"""Quick kernel bench for development: one index (built/cached under --index-dir), fixed ef list, device-resident queries,
single-stream launch time via HIP events.  Not the judged bench (bench.py is)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hsutil import headline_data, load_product

ap = argparse.ArgumentParser()
ap.add_argument("--index-dir", default="/tmp/idx")
ap.add_argument("--efs", default="68")
ap.add_argument("--nq", type=int, default=10000)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--hash-slots", type=int, default=0)
ap.add_argument("--cand-cap", type=int, default=0)
ap.add_argument("--check", action="store_true", help="compare labels/counters of the first 2000 queries with the oracle")
args = ap.parse_args()
import torch
hs = load_product()
D, K = 128, 10
os.makedirs(args.index_dir, exist_ok=True)
hp, sp = os.path.join(args.index_dir, "hnsw.bin"), os.path.join(args.index_dir, "slim.bin")
if not os.path.exists(sp):
    base = headline_data(args.n, D, 123)
    t0 = time.time()
    hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=min(os.cpu_count(), 64))
    hs.convert_slim(hp, sp, D, threads=min(os.cpu_count(), 64))
    print(f"built in {time.time() - t0:.1f}s", file=sys.stderr)
ix = hs.Index(sp, hs.HS_KIND_SLIM, D)
if args.hash_slots or args.cand_cap:
    ix.set_capacity(args.cand_cap, args.hash_slots)
q = headline_data(args.nq, D, 456)
dev = torch.device("cuda", 0)
q_t = torch.from_numpy(q).to(dev)
lab = torch.empty((args.nq, K), dtype=torch.int32, device=dev)
cnt = torch.empty((args.nq,), dtype=torch.int32, device=dev)
st = torch.empty((args.nq, 4), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for ef in [int(x) for x in args.efs.split(",")]:
    ix.set_ef(ef)
    ix.search_ids_dev(q_t, K, lab, None, cnt, st, stream); ix.check(stream)
    s = st.cpu().numpy().astype(np.int64)
    byts = float((s[:, 0] * 4 * D + s[:, 2] * 4 + s[:, 1] * 8).sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ix.search_ids_dev(q_t, K, lab, None, cnt, None, stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps):
        e0.record(); ix.search_ids_dev(q_t, K, lab, None, cnt, None, stream); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ix.check(stream)
    m = float(np.median(ms))
    print(f"ef={ef}: {m:.3f} ms/launch (min {min(ms):.3f}) -> {args.nq / m / 1e3:.2f} M q/s, {byts / m / 1e6:.0f} GB/s = {byts / m / 1e6 / 8000:.3f} of peak; "
          f"n_dist {s[:, 0].mean():.0f} hops {s[:, 1].mean():.0f} replay {int((s[:, 3] == 1).sum())} other-pass {int((s[:, 3] > 1).sum())}", flush=True)
    if args.check:
        from hsutil import Oracle
        ox = Oracle().load(sp, "slim", 0, D)
        ox.set_ef(ef)
        n = min(2000, args.nq)
        o = ox.search_ids(q[:n], K, threads=16)
        g = lab.cpu().numpy().astype(np.uint32)[:n]
        print("   oracle: label sets equal", bool(np.array_equal(np.sort(g, 1), np.sort(o["labels"], 1))), "counters equal", bool(np.array_equal(s[:n, :3], o["counters"][:, :3])))
