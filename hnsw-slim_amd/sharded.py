"""Query-sharded multi-GPU search (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" on CPU for tests).  Queries are independent units and the index is replicated on every GPU
(SURVEY.md 8e): rank r owns the contiguous block [r*nq/G, (r+1)*nq/G) of a global batch, runs the same
kernel, and ONE all-gather of the per-rank top-k joins the results so every rank holds the full [nq x k].
There is no reduction and no other exchange on the data path.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous block of rank `rank` out of `world` over n items (sizes differ by at most one)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def all_gather_rows(local, n_total, world=None, rank=None):
    """All-gather row blocks produced by shard_range back into one [n_total, ...] tensor on every rank."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == width for lo, hi in sizes):
        out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # ragged split (n_total % world != 0): pad every block to the widest, gather, drop the padding
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad)
    return torch.cat([buf[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


def search_sharded(search_fn, queries, k):
    """Strong-scaling entry: `queries` is the same [nq x d] tensor on every rank; search_fn(q_block) returns
    this rank's [rows x k] labels tensor; the result is the full [nq x k] on every rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(queries.shape[0], rank, world)
    local = search_fn(queries[lo:hi])
    assert local.shape[0] == hi - lo and local.shape[1] == k
    return all_gather_rows(local, queries.shape[0], world, rank)
