#!/usr/bin/env python3
"""bench.py -- QPS @ recall@10 >= 0.95 of the batched HNSW-Slim search on SIFT-1M-like data (BASELINE.json).

One "step" = one pass of the hot path (hs_search_batch_dev: HierarchicalNSWSlim::searchKnn(q,k,tableint*)
for every query) over one 10k-query batch that is already resident in HBM.  Workload = BASELINE.json
configs[1]: SIFT-1M-like d=128 L2, M=16 efC=200, Slim defaults, k=10; ef_search = the smallest value of the
sweep {32,48,64,72,80,96,128,192,256} whose recall@10 >= 0.95 on this data (all sweep points are reported in `config`).

Multi-GPU (torchrun, one rank per GPU): the index is replicated, every rank searches its own 10k-query
batch (weak scaling) and the per-rank top-k labels are joined by one RCCL all-gather inside the step.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hsutil import headline_data, load_product  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def ground_truth(torch, base_t, q_t, k, hs=None):
    """Exact L2 k-NN by brute force on the GPU.  With the product module: its exhaustive-scan kernel (the reference's
    BruteforceSearch semantics, exact recipe distances, ties broken by label); otherwise a torch GEMM (fp32;
    exact on integer-valued data)."""
    if hs is not None and base_t.shape[1] % 16 == 0 and k <= 64:
        lab = torch.empty((q_t.shape[0], k), dtype=torch.int64, device=q_t.device)
        dd = torch.empty((q_t.shape[0], k), dtype=torch.float32, device=q_t.device)
        hs.brute_force_dev(base_t, q_t, k, lab, dd)
        return lab.cpu().numpy()
    bn = (base_t * base_t).sum(1)
    out = []
    for s in range(0, q_t.shape[0], 1024):
        q = q_t[s:s + 1024]
        d = bn[None, :] - 2.0 * (q @ base_t.T)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def recall_at_k(labels, gt):
    hits = 0
    for i in range(labels.shape[0]):
        hits += len(set(labels[i].tolist()) & set(gt[i].tolist()))
    return hits / gt.size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--base-size", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", dest="nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=0, help="fixed ef_search (0 = smallest sweep value with recall>=0.95)")
    ap.add_argument("--build-threads", type=int, default=0)
    ap.add_argument("--index-dir", default="", help="reuse/build index files here instead of a temp dir")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cand-cap", type=int, default=0)
    ap.add_argument("--hash-slots", type=int, default=0)
    ap.add_argument("--streams", type=int, default=8,
                    help="HIP streams the timed steps are issued on round-robin (batches in flight); 1 = strictly serial")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (1-GPU box): HS_BENCH_ONE_DEVICE=1 maps every rank to cuda:0, HS_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device); the driver's multi-GPU run uses neither.
    if os.environ.get("HS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("HS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    hs = load_product()
    import importlib.util
    _sp = importlib.util.spec_from_file_location('hs_sharded', os.path.join(ROOT, 'hnsw-slim_amd', 'sharded.py'))
    sharded = importlib.util.module_from_spec(_sp)
    _sp.loader.exec_module(sharded)
    threads = args.build_threads or min(os.cpu_count() or 8, 64)
    N, D, NQ, K = args.n, args.dim, args.nq, args.k

    # ---- data + index (rank 0 builds once, everyone loads the same files) ---------------------------
    tmp = None
    if args.index_dir:
        idir = args.index_dir
        os.makedirs(idir, exist_ok=True)
    elif world > 1:
        idir = os.path.join(tempfile.gettempdir(), f"hs_bench_{os.environ.get('MASTER_PORT', '0')}_{N}_{D}")
        os.makedirs(idir, exist_ok=True)
    else:
        tmp = tempfile.TemporaryDirectory()
        idir = tmp.name
    hpath, spath, bpath = (os.path.join(idir, f) for f in ("hnsw.bin", "slim.bin", "base.npy"))
    t_build = t_conv = 0.0
    if rank == 0 and not (os.path.exists(spath) and os.path.exists(bpath)):
        t0 = time.time()
        base = headline_data(N, D, 123)  # SIFT-like: low-rank integer mixture (hsutil.sift_like)
        log(f"generated base {base.shape} in {time.time() - t0:.1f}s; building HNSW M=16 efC=200 with {threads} threads")
        t0 = time.time()
        hs.build_hnsw(base, hpath, M=16, ef_construction=200, branching_factor="4", seed=100, threads=threads)
        t_build = time.time() - t0
        t0 = time.time()
        hs.convert_slim(hpath, spath, D, threads=threads)
        t_conv = time.time() - t0
        np.save(bpath, base)
        log(f"build {t_build:.1f}s, convertFromHNSW {t_conv:.1f}s")
    if world > 1:
        dist.barrier()
    base = np.load(bpath, mmap_mode="r")
    queries = headline_data(NQ, D, 456 + rank)

    ix = hs.Index(spath, hs.HS_KIND_SLIM, D, hs.HS_METRIC_L2, device=local_rank)
    info = ix.info()
    if args.cand_cap or args.hash_slots:
        ix.set_capacity(args.cand_cap, args.hash_slots)
    base_t = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
    q_t = torch.from_numpy(queries).to(dev)
    gt = ground_truth(torch, base_t, q_t, K, hs)
    del base_t
    torch.cuda.empty_cache()

    d_labels = torch.empty((NQ, K), dtype=torch.int32, device=dev)
    d_counts = torch.empty((NQ,), dtype=torch.int32, device=dev)
    d_stats = torch.empty((NQ, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(ef, stats=False):
        ix.set_ef(ef)
        ix.search_ids_dev(q_t, K, d_labels, None, d_counts, d_stats if stats else None, stream)
        ix.check(stream)

    # ---- ef sweep: recall + counters at every point; pick the operating point -----------------------
    sweep = {}
    efs = [args.ef] if args.ef else [32, 48, 64, 68, 72, 80, 96, 128, 192, 256]
    chosen = None
    for ef in efs:
        run(ef, stats=True)
        torch.cuda.synchronize()
        lab = d_labels.cpu().numpy().astype(np.uint32)
        st = d_stats.cpu().numpy().astype(np.int64)
        rec = recall_at_k(lab, gt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(ef)
        e0.record()
        for _ in range(3):
            run(ef)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        bytes_q = st[:, 0] * 4 * D + st[:, 2] * 4 + st[:, 1] * 8
        sweep[ef] = dict(recall=round(rec, 4), qps=round(NQ / ms * 1e3), n_dist=round(float(st[:, 0].mean()), 1),
                         n_hops=round(float(st[:, 1].mean()), 1), bytes_per_query=round(float(bytes_q.mean())),
                         fallback=int(st[:, 3].sum()), alg_GBs=round(float(bytes_q.sum()) / ms / 1e6, 1))
        if rank == 0:
            log(f"ef={ef}: {sweep[ef]}")
        if chosen is None and rec >= 0.95:
            chosen = ef
            if not args.ef and os.environ.get("HS_BENCH_FULL_SWEEP", "1") == "0":
                break
    if chosen is None:
        chosen = efs[-1]
    if world > 1:  # all ranks must time the same ef
        c = torch.tensor([chosen], device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        chosen = int(c.item())
    ix.set_ef(chosen)
    run(chosen, stats=True)
    torch.cuda.synchronize()
    st = d_stats.cpu().numpy().astype(np.int64)
    alg_bytes_step = float((st[:, 0] * 4 * D + st[:, 2] * 4 + st[:, 1] * 8).sum())
    recall = recall_at_k(d_labels.cpu().numpy().astype(np.uint32), gt)


    # Steps are issued round-robin on S HIP streams with per-stream outputs, so up to S batches are in
    # flight: the tail of one batch (its few longest queries) overlaps the bulk of the next, as in a serving
    # loop.  Every step still runs the complete search of its 10k queries; nothing is shared between steps.
    S = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    outs = [(torch.empty((NQ, K), dtype=torch.int32, device=dev), torch.empty((NQ,), dtype=torch.int32, device=dev)) for _ in range(S)]

    def step(i):
        st = streams[i % S]
        lab, cnt = outs[i % S]
        with torch.cuda.stream(st):
            ix.search_ids_dev(q_t, K, lab, None, cnt, None, st.cuda_stream)
            if world > 1:
                sharded.all_gather_rows(lab, world * NQ, world, rank)  # RCCL over xGMI: every rank holds all top-k

    def finish():
        for st in streams:
            ix.check(st.cuda_stream)  # synchronises the stream and reports capacity errors

    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    assert np.array_equal(np.sort(outs[0][0].cpu().numpy(), axis=1), np.sort(d_labels.cpu().numpy(), axis=1))
    # one launch at a time (what a single rocprofv3 kernel duration corresponds to): HIP events on the launch stream
    ke0, ke1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kern_ms = 0.0
    for _ in range(args.steps):
        ke0.record()
        ix.search_ids_dev(q_t, K, d_labels, None, d_counts, None, stream)
        ke1.record()
        torch.cuda.synchronize()
        kern_ms += ke0.elapsed_time(ke1)
    ix.check(stream)
    kern_ms /= args.steps
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = world * NQ * args.steps / elapsed

    # PCIe-inclusive rate of the host-pointer entry (hs_search_batch: H2D of the queries, search, D2H of the labels);
    # reported for DESIGN.md, never `value`
    host_api_qps = None
    if rank == 0 and world == 1:
        ix.search_ids(queries, K)
        t0 = time.perf_counter()
        for _ in range(5):
            ix.search_ids(queries, K)
        host_api_qps = round(5 * NQ / (time.perf_counter() - t0), 1)

    # ---- CPU baseline: the oracle (port) on this box's host cores, rank 0, N=1 only ------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from hsutil import Oracle
        ox = Oracle().load(spath, "slim", 0, D)
        ox.set_ef(chosen)
        cores = min(os.cpu_count() or 1, 64)
        ns = min(NQ, 2000)
        t0 = time.perf_counter()
        r1 = ox.search_ids(queries[:ns], K, threads=1)
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        rN = ox.search_ids(queries, K, threads=cores)
        tN = time.perf_counter() - t0
        same = bool(np.array_equal(np.sort(rN["labels"], axis=1), np.sort(d_labels.cpu().numpy().astype(np.uint32), axis=1)))
        cpu = dict(value=round(NQ / tN, 1), unit="queries/s", cores=cores, kind="port",
                   sample=f"all {NQ} queries, OpenMP dynamic over {cores} threads, ef={chosen}; serial (as shipped, 1 core) on first {ns}: {ns / t1:.0f} queries/s",
                   serial_qps=round(ns / t1, 1), gpu_label_sets_identical=same)

    if rank == 0:
        # HBM traffic from the PMC counters cannot be collected from inside this process; the committed
        # profile (tools/traffic_cmd.sh -> profiles/r01_traffic.json) is quoted when it is for this workload.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and N == 1_000_000 and NQ == 10_000 and D == 128 and chosen == json.load(open(tpath)).get("ef"):
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        step_ms = elapsed / args.steps * 1e3
        achieved = alg_bytes_step / (step_ms * 1e-3) / 1e9          # timed region: S launches in flight
        achieved_single = alg_bytes_step / (kern_ms * 1e-3) / 1e9   # one launch alone on the GPU
        out = {
            "metric": "QPS @ recall@10>=0.95 (SIFT-1M d=128, k=10)", "value": round(qps, 1), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SIFT-1M-like d={D} L2 (4096-component rank-12 integer mixture), N={N}, batch={NQ} queries/GPU, "
                                   f"HNSW-Slim M=16 efC=200 (Slim defaults), k={K}, ef_search={chosen}",
                       "ef_search": chosen, "recall_at_10": round(recall, 4), "sweep": sweep, "index": info,
                       "build_s": round(t_build, 1), "convert_s": round(t_conv, 1), "build_threads": threads,
                       "pipelining": f"steps issued round-robin on {S} HIP streams (up to {S} batches in flight)",
                       "host_pointer_api_qps_pcie_inclusive": host_api_qps},
            # achieved/frac: algorithmic bytes of one launch / the HIP-event duration of that launch alone on the GPU
            # (what a rocprofv3 kernel-trace average for fast_kernel measures, profiles/r01_kernel_stats_1stream.csv);
            # pipelined_*: the same bytes / the effective per-step time of the timed region, S launches in flight.
            "roofline": {"bound": "hbm", "achieved": round(achieved_single, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_single / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "hs::fast_kernel", "launch_ms": round(kern_ms, 4),
                         "pipelined_achieved": round(achieved, 1), "pipelined_frac": round(achieved / HBM_PEAK_GBS, 4),
                         "pipelined_ms_per_step": round(step_ms, 4), "launches_in_flight": S,
                         "algorithmic_bytes_per_launch": alg_bytes_step},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
