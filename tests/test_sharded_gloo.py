"""world_size-2 `gloo` test (CPU) of the query-sharded path: the union of the shard results must equal the
single-process result bit for bit.  The per-rank searcher here is the CPU oracle -- the test
covers the sharding / all-gather logic that bench.py and the GPU path use, not the kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hsutil import GOLDEN, ROOT, Oracle


def _worker(rank, world, port, nq, ret):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("hs_sharded", os.path.join(ROOT, "hnsw-slim_amd", "sharded.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    ox = Oracle().load(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), "hnsw", 0, 32)
    ox.set_ef(32)
    q = torch.from_numpy(g["queries"][:nq])

    def search_fn(block):
        r = ox.search_pq(block.numpy(), 10)
        return torch.from_numpy(r["labels"].astype(np.int64))

    full = sh.search_sharded(search_fn, q, 10)
    want = torch.from_numpy(ox.search_pq(q.numpy(), 10)["labels"].astype(np.int64))
    ok = bool(torch.equal(full, want))
    lo, hi = sh.shard_range(nq, rank, world)
    ret[rank] = (ok, lo, hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [100, 37])  # even and ragged split
def test_two_rank_shard_union_equals_single(nq):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nq, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == 0 and ret[0][2] == ret[1][1] and ret[1][2] == nq
