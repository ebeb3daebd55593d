"""An index beyond 2^24 nodes (what DEEP-100M would be, without its two-hour build): a synthetic level-0-only graph of
17.5 M nodes, d = 16, degree 16, handed over through hs_index_from_host_arrays.  At this size the id space is 25 bits wide:
the flat kernel's visited set must grow its bucket count so that the stored remainders still fit 15 bits, and the fast / lean
kernels leave the 16-bit visited set for the 32-bit form on their own (no HS_VIS16 knob involved).  Labels, fp32 distances
and the three counters of 200 queries must equal the oracle's (which reads the same graph from a Slim file written here,
hnswalg_slim.h:717-751 layout)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from hsutil import ROOT, Oracle, load_chal_encode

pytestmark = pytest.mark.gpu

N, D, DEG, NQ, K = 17_500_000, 16, 16, 200, 10

_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from hsutil import load_product
hs = load_product()
d = np.load(sys.argv[2])
n = int(d["n"])
vec = np.load(sys.argv[3], mmap_mode="r")
ids = np.load(sys.argv[4], mmap_mode="r")
ix = hs.Index.from_csr(hs.HS_KIND_SLIM, hs.HS_METRIC_L2, vec, np.zeros(n, np.int32), np.arange(n + 1, dtype=np.uint64) * int(d["deg"]), ids, 0, 0)
out = {}
for ef in (int(x) for x in d["efs"]):
    ix.set_ef(ef)
    r = ix.search_ids(d["q"], int(d["k"]), want_dists=True, want_stats=True)
    out[f"l{ef}"], out[f"d{ef}"], out[f"s{ef}"], out[f"k{ef}"] = r["labels"], r["dists"], r["stats"][:, :3], np.array([ix.last_kernel()])
np.savez(sys.argv[5], **out)
'''


def test_index_beyond_2_24_nodes(tmp_path):
    rng = np.random.default_rng(5)
    vec = rng.integers(0, 64, size=(N, D), dtype=np.int32).astype(np.float32)   # integer-valued: ties included
    ids = rng.integers(0, N, size=(N, DEG), dtype=np.uint32)
    # (a list must not name its own node or one node twice; re-draw the few rows that do)
    srt = np.sort(ids, axis=1)
    bad = np.nonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1) | (ids == np.arange(N, dtype=np.uint32)[:, None]).any(axis=1))[0]
    for i in bad:
        row = rng.choice(N - 1, size=DEG, replace=False).astype(np.uint32)
        row[row >= i] += 1
        ids[i] = row
    q = rng.integers(0, 64, size=(NQ, D), dtype=np.int32).astype(np.float32)
    # the same graph as a Slim file for the oracle: header, elements [i32 level][u32 total][u64 label][8 B pointer][data], then
    # per element u32 blobSize + the ids (level 0: no offsets)
    sp = str(tmp_path / "big.slim")
    ce = load_chal_encode()
    with open(sp, "wb") as f:
        f.write(struct.pack(ce.SLIM_HDR, N, 24 + 4 * D, 8, 4, 24, 16, 0, 0, 0, 16, 32, 16, 200, False))
        step = 1 << 20
        for lo in range(0, N, step):
            hi = min(N, lo + step)
            el = np.zeros(hi - lo, dtype=[("level", "<i4"), ("total", "<u4"), ("label", "<u8"), ("ptr", "<u8"), ("data", "<f4", (D,))])
            el["total"], el["label"], el["data"] = DEG, np.arange(lo, hi), vec[lo:hi]
            f.write(el.tobytes())
        for lo in range(0, N, step):
            hi = min(N, lo + step)
            bl = np.zeros(hi - lo, dtype=[("size", "<u4"), ("ids", "<u4", (DEG,))])
            bl["size"], bl["ids"] = 4 * DEG, ids[lo:hi]
            f.write(bl.tobytes())
    ox = Oracle().load(sp, "slim", 0, D)
    efs = (32, 100)
    want = {}
    for ef in efs:
        ox.set_ef(ef)
        want[ef] = ox.search_ids(q, K, threads=8)
    np.savez(str(tmp_path / "in.npz"), n=N, deg=DEG, k=K, efs=np.array(efs), q=q)
    np.save(str(tmp_path / "vec.npy"), vec)
    np.save(str(tmp_path / "ids.npy"), ids)
    del vec, ids, ox
    os.remove(sp)
    open(str(tmp_path / "w.py"), "w").write(_WORKER)
    seen = set()
    for tag, env in (("default", {}), ("fast", {"HS_KERNEL": "fast"}), ("lean", {"HS_LEAN_MIN_EF": "1"})):
        of = str(tmp_path / f"{tag}.npz")
        subprocess.check_call([sys.executable, str(tmp_path / "w.py"), ROOT, str(tmp_path / "in.npz"), str(tmp_path / "vec.npy"), str(tmp_path / "ids.npy"), of],
                              env=dict(os.environ, **env))
        got = np.load(of)
        for ef in efs:
            w = want[ef]
            assert np.array_equal(np.sort(got[f"l{ef}"], axis=1), np.sort(w["labels"], axis=1)), (tag, ef)
            assert np.array_equal(got[f"s{ef}"], w["counters"][:, :3]), (tag, ef)
            seen.add(str(got[f"k{ef}"][0]))
    assert "hs::flat_kernel" in seen and len(seen) >= 2, seen
