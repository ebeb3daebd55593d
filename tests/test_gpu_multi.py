"""Multi-GPU and asynchronous entries of the C ABI on the one-GPU box.

hs_search_batch_sharded: replicated index, contiguous shards, one gather -- parity = the single-device result bit for bit
(SURVEY.md 8e).  RCCL refuses two ranks on one device, so the exchange is rehearsed in the communicator's loopback mode
(device 0 listed n times: same sharding, same buffers, device copies instead of ncclAllGather); the RCCL leg itself runs in
the driver's multi-GPU bench.  Two-rank process test: one process per rank (gloo rendezvous, both on cuda:0), each driving
hs_search_batch_dev on its shard, all-gather, union == single-rank result.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from hsutil import GOLDEN, ROOT, load_product, mixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hs():
    m = load_product()
    assert m.device_count() > 0, "no HIP device visible"
    return m


@pytest.fixture(scope="module")
def slim_file(hs, tmp_path_factory):
    sp = str(tmp_path_factory.mktemp("multi") / "s.bin")
    hs.convert_slim(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), sp, 32)
    return sp


@pytest.mark.parametrize("n_dev", [1, 2, 3])
def test_sharded_equals_single_device(hs, slim_file, n_dev):
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    reps = [hs.Index(slim_file, hs.HS_KIND_SLIM, 32) for _ in range(n_dev)]
    comm = hs.Comm([0] * n_dev)
    for ef in (16, 48):
        for r in reps:
            r.set_ef(ef)
        for nq in (100, 37, 2, 1):   # even, ragged, fewer queries than devices
            q = g["queries"][:nq]
            want = reps[0].search_ids(q, 10, want_dists=True)
            got = comm.search_ids(reps, q, 10, want_dists=True)
            assert np.array_equal(got["labels"], want["labels"]), (n_dev, ef, nq)
            assert got["dists"].tobytes() == want["dists"].tobytes()
            assert np.array_equal(got["cnt"], want["cnt"])
    # priority_queue overload on a vanilla index
    hp = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    vreps = [hs.Index(hp, hs.HS_KIND_HNSW, 32) for _ in range(n_dev)]
    for r in vreps:
        r.set_ef(32)
    want, got = vreps[0].search_pq(g["queries"], 10), comm.search_pq(vreps, g["queries"], 10)
    assert np.array_equal(got["labels"], want["labels"]) and got["dists"].tobytes() == want["dists"].tobytes()
    assert np.array_equal(got["cnt"], g["ef32_cnt"])
    # a replica on the wrong device / a missing replica is refused
    with pytest.raises(hs.HsError):
        hs.Comm([0, 99])


def test_sharded_async_slots_in_flight(hs, slim_file):
    """hs_search_batch_sharded_async / hs_comm_check: three "devices" (loopback), six distinct batches in flight in six slots,
    twice over (a slot is reused after its check); every batch equals the single-device call bit for bit; a slot that still
    holds a batch refuses a second one."""
    base_q = mixture(6 * 257, 32, 91)
    reps = [hs.Index(slim_file, hs.HS_KIND_SLIM, 32) for _ in range(3)]
    for r in reps:
        r.set_ef(40)
    comm = hs.Comm([0, 0, 0])
    assert comm.slots() >= 6
    qs = [hs.PinnedArray((257, 32), np.float32) for _ in range(6)]
    outs = [hs.PinnedArray((257, 10), np.uint32) for _ in range(6)]
    for rnd in range(2):
        for b in range(6):
            qs[b].a[:] = np.roll(base_q, rnd * 13, axis=0)[b * 257:(b + 1) * 257]
            outs[b].a[:] = 0
            comm.search_ids_async(reps, qs[b].a, 10, outs[b].a, slot=b)
        with pytest.raises(hs.HsError):
            comm.search_ids_async(reps, qs[0].a, 10, outs[0].a, slot=0)
        for b in range(6):
            comm.check(reps, b)
            want = reps[0].search_ids(qs[b].a.copy(), 10)
            assert np.array_equal(outs[b].a, want["labels"]), (rnd, b)


def test_async_host_entry_overlapped_streams(hs, slim_file):
    """hs_search_batch_async: distinct batches in page-locked memory issued round-robin on three HIP streams; every batch's
    labels equal the synchronous call's."""
    import torch
    base_q = mixture(3000, 32, 77)
    ix = hs.Index(slim_file, hs.HS_KIND_SLIM, 32)
    ix.set_ef(40)
    nb, bs, S = 7, 400, 3
    streams = [torch.cuda.Stream() for _ in range(S)]
    qs = [hs.PinnedArray((bs, 32), np.float32) for _ in range(nb)]
    outs = [hs.PinnedArray((bs, 10), np.uint32) for _ in range(nb)]
    for b in range(nb):
        qs[b].a[:] = base_q[b * bs:(b + 1) * bs]
        outs[b].a[:] = 0
    for b in range(nb):
        if b >= S:
            ix.check(streams[b % S].cuda_stream)   # the stream's staging buffers are free again once it has drained
        ix.search_ids_async(qs[b].a, 10, outs[b].a, streams[b % S].cuda_stream)
    for s in streams:
        ix.check(s.cuda_stream)
    for b in range(nb):
        want = ix.search_ids(base_q[b * bs:(b + 1) * bs], 10)["labels"]
        assert np.array_equal(outs[b].a, want), b


def test_cpp_facade_sharded_batch(hs, slim_file, tmp_path):
    exe = os.path.join(ROOT, "hnsw-slim_amd", "facade_smoke")
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    q = np.ascontiguousarray(g["queries"][:41])
    qf, out = str(tmp_path / "q.f32"), str(tmp_path / "o.bin")
    q.tofile(qf)
    subprocess.check_call([exe, "slim2", slim_file, "32", qf, "41", "10", "48", out])
    got = np.fromfile(out, np.uint32).reshape(41, 10)
    ix = hs.Index(slim_file, hs.HS_KIND_SLIM, 32)
    ix.set_ef(48)
    ix.set_exact_order(True)
    assert np.array_equal(got, ix.search_ids(q, 10)["labels"])


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from hsutil import load_product
import importlib.util
spec = importlib.util.spec_from_file_location("hs_sharded", os.path.join(sys.argv[1], "hnsw-slim_amd", "sharded.py"))
sh = importlib.util.module_from_spec(spec); spec.loader.exec_module(sh)
hs = load_product()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ix = hs.Index(sys.argv[2], hs.HS_KIND_SLIM, 32)
ix.set_ef(48)
q = torch.from_numpy(np.load(sys.argv[3])).to(dev)
nq, k = q.shape[0], 10
stream = torch.cuda.current_stream().cuda_stream

def search_fn(block):
    lab = torch.empty((block.shape[0], k), dtype=torch.int32, device=dev)
    if block.shape[0]:
        ix.search_ids_dev(block.contiguous(), k, lab, None, None, None, stream)
        ix.check(stream)
    return lab.cpu()          # gloo gathers host tensors

full = sh.search_sharded(search_fn, q, k)
if rank == 0:
    want = torch.empty((nq, k), dtype=torch.int32, device=dev)
    ix.search_ids_dev(q, k, want, None, None, None, stream); ix.check(stream)
    ok = bool(torch.equal(full, want.cpu()))
    open(sys.argv[4], "w").write("OK" if ok else "MISMATCH")
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("nq", [100, 37])
def test_two_rank_processes_drive_the_device_entry(hs, slim_file, tmp_path, nq):
    """One process per rank (as bench.py runs under torchrun; here 2 ranks share cuda:0 and rendezvous over gloo): each rank
    searches its contiguous shard with hs_search_batch_dev, one all-gather joins them; rank 0 checks the union against its
    own single-rank search of the whole batch, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    qf, res, wf = str(tmp_path / "q.npy"), str(tmp_path / "res.txt"), str(tmp_path / "worker.py")
    np.save(qf, g["queries"][:nq])
    open(wf, "w").write(_WORKER)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, wf, ROOT, slim_file, qf, res], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert open(res).read() == "OK"


def test_kernel_variants_parity_under_env(hs, tmp_path):
    """The flat kernel (flat_search.hip: the default on every index it supports -- lazily replayed candidate heap, bucketed visited
    set) and the lean kernel (lean_search.hip, forced for every ef here): same labels, distances and counters as the fast kernel
    on tie-heavy integer data, with starved scratch (overflow list / tier-2 visited set / candidate heap) as well; likewise the
    ordered three-launch pass and the 32-bit form of the visited set, each forced by its environment knob."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from hsutil import GOLDEN, load_product, mixture
hs = load_product()
out = {}
q = mixture(300, 128, 22, n_clusters=64, integer=True)
ix = hs.Index(sys.argv[2] + "/s.bin", hs.HS_KIND_SLIM, 128)
for ef, cap in ((10, (0, 0)), (32, (0, 0)), (64, (0, 0)), (64, (40, 128)), (200, (0, 0))):
    ix.set_ef(ef); ix.set_capacity(*cap)
    r = ix.search_ids(q, 10, want_dists=True, want_stats=True)
    out[f"l{ef}_{cap[0]}"] = np.sort(r["labels"], 1); out[f"d{ef}_{cap[0]}"] = np.sort(r["dists"], 1); out[f"s{ef}_{cap[0]}"] = r["stats"][:, :3]
    p = ix.search_pq(q, 10)
    out[f"pl{ef}_{cap[0]}"] = np.sort(p["labels"], 1)
np.savez(sys.argv[3], **out)
'''
    wf = str(tmp_path / "w.py")
    open(wf, "w").write(code)
    base = mixture(20000, 128, 21, n_clusters=64, integer=True)
    hs.build_hnsw(base, str(tmp_path / "h.bin"), M=16, ef_construction=100, threads=8)
    hs.convert_slim(str(tmp_path / "h.bin"), str(tmp_path / "s.bin"), 128, threads=8)
    res = {}
    # (the last two force the descent / order / level-0 launches of large batches, csrc/capi.cpp, onto this 300-query batch)
    for tag, env in (("fast", {"HS_LEAN_MIN_EF": "100000", "HS_ORDER": "0"}), ("flat", {"HS_KERNEL": "flat", "HS_ORDER": "0"}), ("flat_ordered", {"HS_KERNEL": "flat", "HS_ORDER": "1"}),
                     ("lean", {"HS_LEAN_MIN_EF": "1", "HS_ORDER": "0"}),
                     ("fast_ordered", {"HS_LEAN_MIN_EF": "100000", "HS_ORDER": "1"}), ("lean_ordered", {"HS_LEAN_MIN_EF": "1", "HS_ORDER": "1"}),
                     # the visited set's 32-bit form (what an index beyond 2^(log2(buckets)+16) nodes gets) instead of the 16-bit one
                     ("fast_vis32", {"HS_LEAN_MIN_EF": "100000", "HS_VIS16": "0"}), ("lean_vis32", {"HS_LEAN_MIN_EF": "1", "HS_VIS16": "0"}),
                     # the candidate heap from the first expansion instead of the flat start + materialisation on a tie
                     ("fast_heap", {"HS_LEAN_MIN_EF": "100000", "HS_FLAT": "0"})):
        of = str(tmp_path / f"{tag}.npz")
        subprocess.check_call([sys.executable, wf, ROOT, str(tmp_path), of], env=dict(os.environ, **env))
        res[tag] = np.load(of)
    for key in res["fast"].files:
        assert np.array_equal(res["fast"][key], res["flat"][key]), key
        assert np.array_equal(res["fast"][key], res["flat_ordered"][key]), key
        assert np.array_equal(res["fast"][key], res["lean"][key]), key
        assert np.array_equal(res["fast"][key], res["fast_ordered"][key]), key
        assert np.array_equal(res["fast"][key], res["lean_ordered"][key]), key
        assert np.array_equal(res["fast"][key], res["fast_vis32"][key]), key
        assert np.array_equal(res["fast"][key], res["lean_vis32"][key]), key
        assert np.array_equal(res["fast"][key], res["fast_heap"][key]), key


def test_index_from_host_arrays_equals_index_from_file(hs, slim_file):
    """hs_index_from_host_arrays: the graph handed over as plain arrays (parsed here by the independent Python reader) gives the
    same index as loading the file -- Slim and vanilla kinds, labels, counters."""
    from hsutil import load_chal_encode
    ce = load_chal_encode()
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    s = ce.parse_slim(open(slim_file, "rb").read(), 32)
    a = hs.Index(slim_file, hs.HS_KIND_SLIM, 32)
    b = hs.Index.from_arrays(hs.HS_KIND_SLIM, hs.HS_METRIC_L2, s["rows"], s["level"], s["lists"], s["enterpoint"], s["maxlevel"], labels=s["labels"])
    v = ce.parse_vanilla(open(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), "rb").read())
    c = hs.Index(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), hs.HS_KIND_HNSW, 32)
    d = hs.Index.from_arrays(hs.HS_KIND_HNSW, hs.HS_METRIC_L2, v["rows"], [len(x) - 1 for x in v["lists"]], v["lists"], v["enterpoint"],
                             v["maxlevel"], labels=v["labels"])
    for ef in (10, 48):
        for x in (a, b, c, d):
            x.set_ef(ef)
        for exact in (True, False):
            a.set_exact_order(exact); b.set_exact_order(exact)
            ra, rb = a.search_ids(g["queries"], 10, want_dists=True, want_stats=True), b.search_ids(g["queries"], 10, want_dists=True, want_stats=True)
            assert np.array_equal(ra["labels"], rb["labels"]) and ra["dists"].tobytes() == rb["dists"].tobytes() and np.array_equal(ra["stats"], rb["stats"])
        rc, rd = c.search_pq(g["queries"], 10, want_stats=True), d.search_pq(g["queries"], 10, want_stats=True)
        assert np.array_equal(rc["labels"], rd["labels"]) and rc["dists"].tobytes() == rd["dists"].tobytes() and np.array_equal(rc["stats"], rd["stats"])
    with pytest.raises(hs.HsError):
        hs.Index.from_arrays(hs.HS_KIND_SLIM, hs.HS_METRIC_L2, s["rows"], s["level"], s["lists"], 10 ** 9, s["maxlevel"])


def test_lean_kernel_with_more_than_64k_of_lds(hs, tmp_path):
    """A user candidate-heap capacity that takes the lean kernel's LDS request beyond 64 KiB (lean_search.hip sets
    hipFuncAttributeMaxDynamicSharedMemorySize for it): the launch must succeed and answer like the default kernel."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from hsutil import load_product, mixture
hs = load_product()
q = mixture(100, 128, 32, n_clusters=32, integer=True)
ix = hs.Index(sys.argv[2] + "/s.bin", hs.HS_KIND_SLIM, 128)
ix.set_ef(64); ix.set_capacity(int(sys.argv[4]), 0)
r = ix.search_ids(q, 10, want_dists=True, want_stats=True)
np.savez(sys.argv[3], l=np.sort(r["labels"], 1), d=np.sort(r["dists"], 1), s=r["stats"][:, :3], k=np.array([ix.last_kernel()]))
'''
    wf = str(tmp_path / "w.py")
    open(wf, "w").write(code)
    base = mixture(8000, 128, 31, n_clusters=32, integer=True)
    hs.build_hnsw(base, str(tmp_path / "h.bin"), M=16, ef_construction=80, threads=8)
    hs.convert_slim(str(tmp_path / "h.bin"), str(tmp_path / "s.bin"), 128, threads=8)
    res = {}
    for tag, env, cap in (("default", {}, "0"), ("lean_big", {"HS_KERNEL": "lean", "HS_LEAN_MIN_EF": "1", "HS_ORDER": "0"}, "9000")):
        of = str(tmp_path / f"{tag}.npz")
        subprocess.check_call([sys.executable, wf, ROOT, str(tmp_path), of, cap], env=dict(os.environ, **env))
        res[tag] = np.load(of)
    assert str(res["lean_big"]["k"][0]) == "hs::lean_kernel" and str(res["default"]["k"][0]) == "hs::flat_kernel"
    for key in ("l", "d", "s"):
        assert np.array_equal(res["default"][key], res["lean_big"][key]), key


def test_async_entry_reads_and_writes_mapped_host_buffers_in_place(hs, slim_file):
    """hs_search_batch_async with page-locked (device-mapped) buffers: results land in the caller's buffer without a copy back,
    small batches are read in place (<= 2 MiB of queries), larger ones through the staging copy; views at an offset inside the
    allocation and pageable (unmapped) buffers take whichever path applies -- all equal to the synchronous call."""
    import torch
    ix = hs.Index(slim_file, hs.HS_KIND_SLIM, 32)
    ix.set_ef(40)
    st = torch.cuda.Stream()
    for nq, off in ((300, 0), (300, 7), (20000, 0), (20000, 3)):   # 20000 x 32 floats = 2.4 MiB: staged both ways
        base_q = mixture(nq + off, 32, 91 + nq + off)
        qp, op, dp, cp = hs.PinnedArray((nq + off, 32), np.float32), hs.PinnedArray((nq + off, 10), np.uint32), hs.PinnedArray((nq + off, 10), np.float32), hs.PinnedArray((nq + off,), np.uint32)
        qp.a[:] = base_q
        op.a[:] = 0xFFFFFFFF
        dp.a[:] = -1.0
        cp.a[:] = 77
        ix.search_ids_async(qp.a[off:], 10, op.a[off:], st.cuda_stream, dists_pinned=dp.a[off:], counts_pinned=cp.a[off:])
        ix.check(st.cuda_stream)
        want = ix.search_ids(base_q[off:], 10, want_dists=True)
        assert np.array_equal(op.a[off:], want["labels"]), (nq, off)
        assert np.array_equal(dp.a[off:].view(np.uint32), want["dists"].view(np.uint32)), (nq, off)
        assert np.all(cp.a[off:] == 10) and np.all(op.a[:off] == 0xFFFFFFFF) and np.all(cp.a[:off] == 77), (nq, off)
        # pageable output buffer with a mapped query buffer, and the other way round
        lab = np.zeros((nq, 10), np.uint32)
        ix.search_ids_async(qp.a[off:], 10, lab, st.cuda_stream)
        ix.check(st.cuda_stream)
        assert np.array_equal(lab, want["labels"]), (nq, off)
        op.a[:] = 0
        ix.search_ids_async(np.ascontiguousarray(base_q[off:]), 10, op.a[off:], st.cuda_stream)
        ix.check(st.cuda_stream)
        assert np.array_equal(op.a[off:], want["labels"]), (nq, off)
