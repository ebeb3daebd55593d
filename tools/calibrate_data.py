#!/usr/bin/env python3
"""Calibrate the synthetic SIFT-like generator: recall@10 of vanilla HNSW and HNSW-Slim (GPU search)
vs ef for a few generator settings at full size.  Run on the GPU box; prints one line per setting."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from hsutil import load_product, sift_like  # noqa: E402
from bench import ground_truth, recall_at_k  # noqa: E402

import torch  # noqa: E402

hs = load_product()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
# (n_clusters, rank, sigma_sub, sigma_iso, centre_lo, centre_hi)
settings = [(1, 16, 40.0, 4.0, 80.0, 80.0), (256, 16, 40.0, 4.0, 70.0, 90.0), (4096, 12, 40.0, 4.0, 40.0, 110.0), (256, 32, 30.0, 3.0, 75.0, 85.0)]
dev = torch.device("cuda", 0)
for ncl, rank, ssub, siso, clo, chi in settings:
    t0 = time.time()
    base = sift_like(N, 128, 123, n_clusters=ncl, rank=rank, sigma_sub=ssub, sigma_iso=siso, centre_lo=clo, centre_hi=chi)
    q = sift_like(10000, 128, 456, n_clusters=ncl, rank=rank, sigma_sub=ssub, sigma_iso=siso, centre_lo=clo, centre_hi=chi)
    tg = time.time() - t0
    with tempfile.TemporaryDirectory() as tmp:
        hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
        t0 = time.time()
        hs.build_hnsw(base, hp, M=16, ef_construction=200, threads=64)
        tb = time.time() - t0
        hs.convert_slim(hp, sp, 128, threads=64)
        bt = torch.from_numpy(base).to(dev)
        qt = torch.from_numpy(q).to(dev)
        gt = ground_truth(torch, bt, qt, 10)
        del bt
        line = f"ncl={ncl} rank={rank} sub={ssub} iso={siso} c=[{clo},{chi}] gen={tg:.1f}s build={tb:.1f}s |"
        for kind, path in ((hs.HS_KIND_SLIM, sp), (hs.HS_KIND_HNSW, hp)):
            ix = hs.Index(path, kind, 128)
            line += " slim:" if kind == hs.HS_KIND_SLIM else " hnsw:"
            for ef in (32, 64, 128, 256):
                ix.set_ef(ef)
                if kind == hs.HS_KIND_SLIM:
                    r = ix.search_ids(q, 10, want_stats=True)
                    lab = r["labels"]
                else:
                    r = ix.search_pq(q, 10, want_stats=True)
                    lab = r["labels"].astype(np.uint32)
                line += f" ef{ef}={recall_at_k(lab, gt):.3f}/nd{r['stats'][:,0].mean():.0f}"
            ix.close()
        print(line, flush=True)
