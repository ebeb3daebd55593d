"""CPU-only tests of the product's host side: C-ABI exports, heap/selection emulation vs libstdc++,
the HNSW builder vs the compiled reference's index file, Slim conversion + file round trip, errors."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from hsutil import GOLDEN, ROOT, load_product, mixture

L2, IP = 0, 1


@pytest.fixture(scope="module")
def hs():
    m = load_product()
    m.build_library()
    return m


def test_c_abi_exports_every_declared_symbol(hs):
    hdr = open(os.path.join(ROOT, "include", "hnsw_slim_amd.h")).read()
    declared = set(re.findall(r"\b(hs_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"hs_status"}
    L = hs.lib()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/hnsw_slim_amd.h but not exported"
    assert set(hs.EXPORTS) <= declared


def test_heap_emulation_matches_libstdcxx(hs):
    exe = os.path.join(ROOT, "hnsw-slim_amd", "selftest")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "selftest"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("name,metric", [("l2_cont_d32", L2), ("l2_int_d16", L2), ("ip_d48", IP), ("l2_cont_d20", L2),
                                         ("l2_cont_d21", L2), ("l2_cont_d10", L2), ("ip_d20", IP), ("ip_d21", IP),
                                         ("ip_d10", IP)])
def test_serial_builder_writes_reference_bytes(hs, tmp_path, name, metric):
    """threads=1 build == the reference's serial addPoint loop + saveIndex, byte for byte."""
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    out = str(tmp_path / "mine.bin")
    hs.build_hnsw(g["base"], out, metric=metric, M=int(g["M"]), ef_construction=int(g["efC"]), branching_factor="4", seed=100, threads=1)
    ref = open(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "rb").read()
    assert open(out, "rb").read() == ref


def test_parallel_builder_is_searchable(hs, oracle, tmp_path):
    base = mixture(3000, 32, 11)
    q = mixture(50, 32, 12)
    out = str(tmp_path / "par.bin")
    hs.build_hnsw(base, out, M=8, ef_construction=100, threads=4)
    ix = oracle.load(out, "hnsw", L2, 32)
    ix.set_ef(64)
    r = ix.search_pq(q, 10)
    gt = oracle.brute_force(L2, base, q, 10)
    # pop order is farthest-first; compare as sets
    hits = sum(len(set(map(int, r["labels"][i])) & set(map(int, gt[i]))) for i in range(len(q)))
    assert hits / (10 * len(q)) > 0.9


def test_slim_convert_roundtrip(hs, oracle, tmp_path):
    """convertFromHNSW -> saveIndex -> (oracle) loadIndex: structure invariants of the CHAL blobs."""
    src = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    out = str(tmp_path / "slim.bin")
    hs.convert_slim(src, out, 32)
    data = open(out, "rb").read()
    n = int(np.frombuffer(data, np.uint64, 1, 0)[0])
    spe = int(np.frombuffer(data, np.uint64, 1, 8)[0])
    assert n == 2000 and spe == 24 + 4 * 32
    # independent parse of the documented layout ('<6Q2iI4Q?' = 93-byte header)
    off = 93
    el = np.frombuffer(data, np.uint8, n * spe, off).reshape(n, spe)
    level = el[:, 0:4].copy().view(np.int32)[:, 0]
    total = el[:, 4:8].copy().view(np.uint32)[:, 0]
    off += n * spe
    for i in range(n):
        sz = int(np.frombuffer(data, np.uint32, 1, off)[0]); off += 4
        assert sz == 2 * level[i] + 4 * total[i]
        if sz and total[i]:
            offs = np.frombuffer(data, np.uint16, level[i], off)
            ids = np.frombuffer(data, np.uint32, total[i], off + 2 * level[i])
            assert np.all(np.diff(np.concatenate([[0], offs, [total[i]]]).astype(np.int64)) >= 0)
            assert ids.max() < n
            lvl0 = ids[: (offs[0] if level[i] else total[i])]
            assert len(lvl0) <= 16 and len(set(lvl0.tolist())) == len(lvl0)  # unique; id-sorted unless re-pruned (slim.h:1005-1010, 1038-1062)
            off += sz
    assert off == len(data)
    # vectors and labels carried over unchanged
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    assert np.array_equal(el[:, 24:].copy().view(np.float32), g["base"])
    # and the oracle can search it
    ix = oracle.load(out, "slim", L2, 32)
    ix.set_ef(64)
    r = ix.search_ids(g["queries"], 10)
    gt = oracle.brute_force(L2, g["base"], g["queries"], 10)
    hits = sum(len(set(map(int, r["labels"][i])) & set(map(int, gt[i]))) for i in range(len(gt)))
    assert hits / gt.size > 0.8


def test_loaders_read_a_memory_image_like_the_file(hs, tmp_path):
    """BinSource / MemBuf (behind hs_index_load_mem): same graphs from bytes in memory as from the file; truncation rejected."""
    exe = os.path.join(ROOT, "hnsw-slim_amd", "selftest")
    hp = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(hp, sp, 32)
    out = subprocess.run([exe, "loadmem", hp, sp, "32"], capture_output=True, text=True)
    assert out.returncode == 0, f"rc={out.returncode} {out.stdout} {out.stderr}"


def test_error_conventions(hs, tmp_path):
    with pytest.raises(hs.HsError, match="dim must be > 0"):
        hs.Index(str(tmp_path / "x"), hs.HS_KIND_SLIM, 0, hs.HS_METRIC_IP)
    if hs.device_count() == 0:
        with pytest.raises(hs.HsError) as e:
            hs.Index(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), hs.HS_KIND_HNSW, 32)
        assert e.value.status == hs.HS_ERR_DEVICE  # no CPU fallback: fails loudly
    with pytest.raises(hs.HsError, match="Cannot open file"):
        hs.convert_slim(str(tmp_path / "nope.bin"), str(tmp_path / "o.bin"), 32)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(open(os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin"), "rb").read()[:5000])
    with pytest.raises(hs.HsError, match="corrupted"):
        hs.convert_slim(str(bad), str(tmp_path / "o.bin"), 32)
    # the SlimQ graph harness follows the same conventions
    with pytest.raises(hs.HsError, match="Cannot open file"):
        hs.convert_slimq_graph(str(tmp_path / "nope.bin"), str(tmp_path / "o.bin"), 32)
    with pytest.raises(hs.HsError, match="corrupted"):
        hs.convert_slimq_graph(str(bad), str(tmp_path / "o.bin"), 32)
    with pytest.raises(hs.HsError, match="M must be >= 2"):
        hs.build_rabitq_hnsw(np.zeros((4, 64), np.float32), str(tmp_path / "o.bin"), M=1)
    with pytest.raises(hs.HsError, match="bad metric"):
        hs.build_rabitq_hnsw(np.zeros((4, 64), np.float32), str(tmp_path / "o.bin"), metric=7)


def test_facade_error_conventions(hs):
    """A caller compiled against the hnswlib-compatible facade sees the reference's exception texts."""
    exe = os.path.join(ROOT, "hnsw-slim_amd", "facade_smoke")
    subprocess.check_call(["make", "-C", os.path.dirname(exe), "facade_smoke"])
    out = subprocess.run([exe, "errors"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_hot_kernels_keep_their_wave_budget():
    """Register pressure is part of the design: the hot kernels must stay at their resident-wave budgets without
    scratch (measured: one wave less per SIMD costs 10-25 % of throughput).  Read from the compiler's report."""
    import re
    path = os.path.join(ROOT, "hnsw-slim_amd", "resource_usage.txt")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.dirname(path), "-B", "libhnsw_slim_amd.so"])
    txt = open(path).read()
    kern = {}
    for m in re.finditer(r"Function Name: (\S+)\s+VGPRs: (\d+)\s+ScratchSize \[bytes/lane\]: (\d+)\s+Occupancy \[waves/SIMD\]: (\d+)", txt):
        kern[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
    want = {
        "_ZN2hs11fast_kernelILi0ELi1ELi8ELb0ELb1EEEvNS_8DevIndexENS_10SearchArgsE": 4,   # d=128 L2, ef <= 64
        "_ZN2hs11fast_kernelILi0ELi2ELi8ELb0ELb1EEEvNS_8DevIndexENS_10SearchArgsE": 4,   # d=128 L2, ef <= 128 (the bench point)
        "_ZN2hs12slimq_kernelILi0ELi2ELi2ELb0EEEvNS_8DevIndexENS_8DevSlimQENS_9SlimQArgsE": 7,
        "_ZN2hs12slimq_kernelILi0ELi4ELi2ELb0EEEvNS_8DevIndexENS_8DevSlimQENS_9SlimQArgsE": 6,
        "_ZN2hs14bf_scan_kernelILi0EEEvPKfPKmjjS2_jjjPNS_7BfEntryE": 5,
    }
    for name, waves in want.items():
        assert name in kern, f"{name} missing from resource_usage.txt"
        vgpr, scratch, occ = kern[name]
        assert occ >= waves and scratch == 0, f"{name}: {vgpr} VGPRs, {scratch} B scratch, {occ} waves/SIMD (budget {waves})"


def test_flat_kernel_visited_set_plan_divides_exactly():
    """The flat kernel's visited set keeps id i in bucket h mod nb with remainder h div nb, h = (i * odd) mod 2^B a bijection of the
    id space (csrc/flat_search.hip); the division is umulhi(h, m) >> s.  For every plan the host makes -- any index size, ef, launch
    size -- that must be exact over the whole id space and the remainder must fit the 15 bits a bucket slot has."""
    hs = load_product()
    rng = np.random.default_rng(11)
    shapes = [(1000, 10, 1), (100_000, 64, 100), (1_000_000, 70, 10_000), (1_000_000, 256, 1250), (17_500_000, 100, 200),
              (100_000_000, 256, 10_000), (2_000_000_000, 128, 10_000)]
    shapes += [(int(rng.integers(2, 1 << 31)), int(rng.integers(1, 257)), int(rng.integers(1, 40_000))) for _ in range(40)]
    for n, ef, nq in shapes:
        p = hs.debug_flat_plan(n, ef, nq)
        assert (1 << p["bits"]) >= n and p["bits"] <= 31
        if not p["ok"]:
            continue
        nb, m, s, B = p["nb"], p["mul"], p["sh"], p["bits"]
        assert ((1 << B) - 1) // nb <= 32767, (n, ef, nq, p)
        if B <= 22:
            h = np.arange(1 << B, dtype=np.uint64)
        else:
            edge = np.array([0, 1, nb - 1, nb, nb + 1, (1 << B) - 1, (1 << B) - 2, ((1 << B) // nb) * nb, ((1 << B) // nb) * nb - 1], np.uint64)
            mult = (rng.integers(1, (1 << B) // nb + 1, size=200_000).astype(np.uint64) * np.uint64(nb))
            mult = mult[mult < (1 << B)]
            h = np.concatenate([edge, rng.integers(0, 1 << B, size=1_000_000).astype(np.uint64), mult, mult - np.uint64(1)])
        q = ((h * np.uint64(m)) >> np.uint64(32)) >> np.uint64(s)
        assert np.array_equal(q, h // np.uint64(nb)), (n, ef, nq, p)
