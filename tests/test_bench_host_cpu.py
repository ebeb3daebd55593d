"""Host-side logic of bench.py that runs before any GPU work: the label checksum every timed batch is checked with, and the
hardware-queue hint for runs whose per-rank shards are small (profiles/r03_small_batch_queues.log)."""
import os
import sys

import numpy as np

from hsutil import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_checksum_is_order_independent_within_a_query_and_sensitive_to_a_changed_label():
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 1_000_000, (500, 10)).astype(np.uint32)
    perm = np.stack([rng.permutation(row) for row in lab])
    assert bench.checksum(lab) == bench.checksum(perm)
    bad = lab.copy()
    bad[17, 3] += 1
    assert bench.checksum(lab) != bench.checksum(bad)


def test_hardware_queue_hint(monkeypatch):
    def hint(nq, world, scaling, preset=None, one_device=False):
        monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
        monkeypatch.delenv("HS_BENCH_ONE_DEVICE", raising=False)
        if preset:
            monkeypatch.setenv("GPU_MAX_HW_QUEUES", preset)
        if one_device:
            monkeypatch.setenv("HS_BENCH_ONE_DEVICE", "1")
        bench.hip_queue_hint(nq, world, scaling)
        return os.environ.get("GPU_MAX_HW_QUEUES")
    assert hint(10000, 1, "auto") is None              # the default command: runtime default
    assert hint(10000, 2, "auto") is None              # 5000 queries per rank
    assert hint(10000, 4, "auto") == "8"               # 2500 per rank, one process per GPU
    assert hint(10000, 8, "auto") == "8"
    assert hint(10000, 8, "weak") is None              # every rank its own 10k batches
    assert hint(1250, 1, "auto") == "16"               # a single process issuing small batches
    assert hint(10000, 8, "auto", preset="4") == "4"   # the caller's setting wins
    assert hint(2000, 2, "auto", one_device=True) is None   # ranks rehearsing on one GPU: never
