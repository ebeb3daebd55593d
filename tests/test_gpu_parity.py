"""GPU parity tests (MI355X): the HIP search path, called through the C ABI, against
  (a) the compiled reference's own outputs (tests/golden, vanilla HNSW), and
  (b) the CPU oracle (oracle/) on the same index file and queries.
Bar: neighbour ids / labels bit-exact (including the reference's output order where it defines one),
fp32 distances bit-exact (the north star allows 1e-4 relative; we hold the stricter bar), counters equal.
"""
import os

import numpy as np
import pytest

from hsutil import GOLDEN, load_chal_encode, load_product, mixture

pytestmark = pytest.mark.gpu
L2, IP = 0, 1


@pytest.fixture(scope="module")
def hs():
    m = load_product()
    assert os.path.exists(m.LIB_PATH), "HIP extension missing: run __graft_entry__.build()"
    assert m.device_count() > 0, "no HIP device visible"
    return m


def _pq_sorted(d, l, c):
    """(dist,label) multiset per query, sorted, as comparable arrays."""
    out = []
    for i in range(len(c)):
        n = int(c[i])
        rec = sorted(zip(d[i, :n].view(np.uint32).tolist(), l[i, :n].tolist()))
        out.append(rec)
    return out


@pytest.mark.parametrize("name,metric,dim", [("l2_cont_d32", L2, 32), ("l2_int_d16", L2, 16), ("ip_d48", IP, 48),
                                             ("l2_cont_d20", L2, 20), ("l2_cont_d21", L2, 21), ("l2_cont_d10", L2, 10),
                                             ("ip_d20", IP, 20), ("ip_d21", IP, 21), ("ip_d10", IP, 10)])
def test_vanilla_vs_compiled_reference(hs, oracle, name, metric, dim):
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    path = os.path.join(GOLDEN, f"{name}.hnsw.bin")
    ix = hs.Index(path, hs.HS_KIND_HNSW, dim, metric)
    ox = oracle.load(path, "hnsw", metric, dim)
    k = int(g["k"])
    for ef in g["efs"]:
        ef = int(ef)
        ix.set_ef(ef)
        ox.set_ef(ef)
        r = ix.search_pq(g["queries"], k, want_stats=True)
        assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
        assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(g[f"ef{ef}_dists"], g[f"ef{ef}_labels"], g[f"ef{ef}_cnt"]), f"{name} ef={ef}"
        assert np.array_equal(r["stats"][:, 0], g[f"ef{ef}_calls"]), "distance-evaluation count differs from the reference"
        # raw heap arrays: identical layout to the oracle's libstdc++ heaps
        raw = ix.search_raw(g["queries"], k, mode=hs.HS_MODE_PQ)
        o = ox.search_pq(g["queries"], k)
        assert np.array_equal(raw["raw_sz"], o["raw_sz"])
        for i in range(len(raw["raw_sz"])):
            n = int(raw["raw_sz"][i])
            assert np.array_equal(raw["raw_i"][i, :n], o["raw_i"][i, :n])
            assert raw["raw_d"][i, :n].tobytes() == o["raw_d"][i, :n].tobytes()
        assert np.array_equal(raw["stats"][:, :3], o["counters"][:, :3])


# ---- HierarchicalNSWSlim on the GPU cross-pinned to the compiled vanilla reference (see tests/test_oracle_golden.py) -----
def _verbatim(tmp_path, name, **kw):
    ce = load_chal_encode()
    sp = tmp_path / f"{name}.verbatim.slim"
    sp.write_bytes(ce.vanilla_to_slim_verbatim(open(os.path.join(GOLDEN, f"{name}.hnsw.bin"), "rb").read(), **kw))
    return str(sp)


@pytest.mark.parametrize("name,metric,dim", [("l2_cont_d32", L2, 32), ("l2_int_d16", L2, 16), ("ip_d48", IP, 48),
                                             ("l2_cont_d20", L2, 20), ("l2_cont_d21", L2, 21), ("l2_cont_d10", L2, 10),
                                             ("ip_d20", IP, 20), ("ip_d21", IP, 21), ("ip_d10", IP, 10)])
def test_slim_verbatim_encoding_vs_compiled_reference(hs, tmp_path, name, metric, dim):
    """A reference-built graph re-encoded verbatim as a Slim file (written by Python, garbage in the stale pointer bytes):
    hs_search_batch in HS_MODE_SLIM_IDS (= HierarchicalNSWSlim::searchKnn(q,k,tableint*), hnswalg_slim.h:2030-2131) must give
    the compiled vanilla reference's k-set, fp32 distances and distance-call count (minus the entry distance only vanilla
    recomputes, hnswalg.h:347-351) -- strict kernel and fast kernel.  No oracle involved."""
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = hs.Index(_verbatim(tmp_path, name), hs.HS_KIND_SLIM, dim, metric)
    k = int(g["k"])
    for exact in (True, False):
        ix.set_exact_order(exact)
        for ef in g["efs"]:
            ef = int(ef)
            ix.set_ef(ef)
            r = ix.search_ids(g["queries"], k, want_dists=True, want_stats=True)
            assert np.array_equal(r["stats"][:, 0], g[f"ef{ef}_calls"] - 1), f"{name} ef={ef}: distance evaluations differ"
            for i in range(len(r["labels"])):
                got = sorted(zip(r["dists"][i].view(np.uint32).tolist(), r["labels"][i].tolist()))
                want = sorted(zip(g[f"ef{ef}_dists"][i].view(np.uint32).tolist(), g[f"ef{ef}_labels"][i].tolist()))
                if name != "l2_int_d16":
                    assert got == want, f"{name} ef={ef} exact={exact} query {i}"
                else:   # ties across the k-th boundary: nth_element and pop_heap pick by layout; the distance multiset is unique
                    assert [x[0] for x in got] == [x[0] for x in want], f"{name} ef={ef} exact={exact} query {i}"


def test_slim_filtered_verbatim_vs_compiled_reference(hs, oracle, tmp_path):
    """searchKnn(q,k,isIdAllowed) on the verbatim Slim file against the compiled vanilla reference's filtered search, on the
    queries where the two classes' different entry handling cannot matter: the level-0 entry passes the filter (Slim always
    seeds the result heap with it, hnswalg_slim.h:2100; vanilla only if allowed, hnswalg.h:347-362) and pre-marking the global
    enter point (hnswalg_slim.h:1796) changes nothing (checked on the oracle, with and without the pre-mark)."""
    name, dim = "l2_cont_d32", 32
    g = np.load(os.path.join(GOLDEN, f"{name}_filter.npz"))
    sp = _verbatim(tmp_path, name)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, dim)
    ox = oracle.load(sp, "slim", L2, dim)
    allowed = (ix.labels() % int(g["mod"]) != int(g["rem"])).astype(np.uint8)
    ox.set_filter(allowed)
    entry_ok = allowed[ox.entry(g["queries"])] == 1
    k = int(g["k"])
    compared = 0
    for ef in g["efs"]:
        ef = int(ef)
        ix.set_ef(ef); ox.set_ef(ef)
        ox.set_mark_ep(1); a = ox.search_pq(g["queries"], k)
        ox.set_mark_ep(0); b = ox.search_pq(g["queries"], k)
        same = np.all(a["labels"] == b["labels"], axis=1) & np.all(a["counters"][:, :3] == b["counters"][:, :3], axis=1)
        ok = entry_ok & same
        r = ix.search_filtered(g["queries"], k, allowed, want_stats=True)
        assert np.array_equal(r["cnt"][ok], g[f"ef{ef}_cnt"][ok])
        got, want = _pq_sorted(r["dists"], r["labels"], r["cnt"]), _pq_sorted(g[f"ef{ef}_dists"], g[f"ef{ef}_labels"], g[f"ef{ef}_cnt"])
        assert all(got[i] == want[i] for i in np.nonzero(ok)[0]), f"ef={ef}"
        assert np.array_equal(r["stats"][ok, 0], g[f"ef{ef}_calls"][ok] - 1)
        compared += int(ok.sum())
    assert compared >= 0.5 * len(entry_ok) * len(g["efs"])


def _slim_case(hs, oracle, tmp_path, base, queries, dim, metric, M, efC, efs, k=10, threads=8, fast=True, **slim_kw):
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    hs.build_hnsw(base, hp, metric=metric, M=M, ef_construction=efC, threads=threads)
    hs.convert_slim(hp, sp, dim, metric=metric, threads=threads, **slim_kw)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, dim, metric)
    ox = oracle.load(sp, "slim", metric, dim)
    for ef in efs:
        ix.set_ef(ef)
        ox.set_ef(ef)
        o = ox.search_ids(queries, k, threads=8)
        # --- strict kernel: the reference's array ORDER, raw heap arrays, counters ---
        ix.set_exact_order(True)
        r = ix.search_ids(queries, k, want_dists=True, want_stats=True)
        assert np.array_equal(r["labels"], o["labels"]), f"ef={ef}: labels differ from searchKnn(q,k,tableint*)"
        assert np.array_equal(r["stats"][:, :3], o["counters"][:, :3]), f"ef={ef}: counters differ"
        raw = ix.search_raw(queries, k)
        assert np.array_equal(raw["raw_sz"], o["raw_sz"])
        cap = max(ef, k)
        mask = np.arange(cap)[None, :] < raw["raw_sz"][:, None]
        assert np.array_equal(raw["raw_i"][mask], o["raw_i"][mask])
        assert raw["raw_d"][mask].tobytes() == o["raw_d"][mask].tobytes()
        # priority_queue overload (marks the enter point visited first)
        op = ox.search_pq(queries, k, threads=8)
        rp = ix.search_pq(queries, k)
        assert np.array_equal(rp["cnt"], op["cnt"])
        assert _pq_sorted(rp["dists"], rp["labels"], rp["cnt"]) == _pq_sorted(op["dists"], op["labels"], op["cnt"])
        # --- default (fast kernel + tie re-runs): identical k-subset per query, sorted by distance ---
        ix.set_exact_order(False)
        f = ix.search_ids(queries, k, want_dists=True, want_stats=True)
        assert np.array_equal(np.sort(f["labels"], axis=1), np.sort(o["labels"], axis=1)), f"ef={ef}: fast-path id sets differ"
        assert np.array_equal(f["stats"][:, :3], o["counters"][:, :3]), f"ef={ef}: fast-path counters differ"
        first_pass = f["stats"][:, 3] == 0
        if fast:  # the fast kernel needs threshold_level == 0 and ef > k; otherwise the strict kernel answers
            assert np.all(np.diff(f["dists"][first_pass], axis=1) >= 0)
        assert np.array_equal(np.sort(f["dists"], axis=1).view(np.uint32), np.sort(r["dists"], axis=1).view(np.uint32))
        fp = ix.search_pq(queries, k)
        assert _pq_sorted(fp["dists"], fp["labels"], fp["cnt"]) == _pq_sorted(op["dists"], op["labels"], op["cnt"])
    return ix, ox


def test_slim_from_golden_index(hs, oracle, tmp_path):
    g = np.load(os.path.join(GOLDEN, "l2_int_d16.npz"))
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(os.path.join(GOLDEN, "l2_int_d16.hnsw.bin"), sp, 16)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, 16)
    ox = oracle.load(sp, "slim", L2, 16)
    for ef in (10, 32, 64, 200):
        ix.set_ef(ef)
        ox.set_ef(ef)
        want = ox.search_ids(g["queries"], 10)["labels"]
        ix.set_exact_order(True)
        assert np.array_equal(ix.search_ids(g["queries"], 10)["labels"], want)
        ix.set_exact_order(False)  # tie-heavy data: many queries take the tie re-run, sets still identical
        got = ix.search_ids(g["queries"], 10, want_stats=True)
        assert np.array_equal(np.sort(got["labels"], axis=1), np.sort(want, axis=1))
        if ef > 10:
            assert (got["stats"][:, 3] == 1).sum() > 0, "expected distance ties across the k-th boundary on this data"


def test_slim_sift_like_d128(hs, oracle, tmp_path):
    base = mixture(20000, 128, 21, n_clusters=64, integer=True)
    q = mixture(400, 128, 22, n_clusters=64, integer=True)
    _slim_case(hs, oracle, tmp_path, base, q, 128, L2, 16, 200, [32, 64, 128, 256])


def test_slim_continuous_d96(hs, oracle, tmp_path):
    base = mixture(8000, 96, 23, lo=-1, hi=1, sigma=0.3)
    q = mixture(200, 96, 24, lo=-1, hi=1, sigma=0.3)
    _slim_case(hs, oracle, tmp_path, base, q, 96, L2, 16, 100, [16, 64])


def test_slim_gist_like_d960(hs, oracle, tmp_path):
    base = np.clip(mixture(3000, 960, 25, lo=0.2, hi=0.8, sigma=0.08), 0, 1)
    q = np.clip(mixture(64, 960, 26, lo=0.2, hi=0.8, sigma=0.08), 0, 1)
    _slim_case(hs, oracle, tmp_path, base, q, 960, L2, 16, 100, [64])


def test_slim_ip_d768(hs, oracle, tmp_path):
    base = mixture(3000, 768, 27, lo=-1, hi=1, sigma=0.5)
    q = mixture(64, 768, 28, lo=-1, hi=1, sigma=0.5)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _slim_case(hs, oracle, tmp_path, base.astype(np.float32), q.astype(np.float32), 768, IP, 16, 100, [64])


def test_slim_threshold_level(hs, oracle, tmp_path):
    base = mixture(6000, 32, 29)
    q = mixture(200, 32, 30)
    _slim_case(hs, oracle, tmp_path, base, q, 32, L2, 8, 100, [16, 64], fast=False, threshold_level=1)


def test_fallback_pass_is_exact(hs, oracle, tmp_path):
    """Starve the first-pass scratch so queries overflow into the whole-CU fallback: same answers."""
    base = mixture(6000, 32, 31)
    q = mixture(128, 32, 32)
    ix, ox = _slim_case(hs, oracle, tmp_path, base, q, 32, L2, 8, 100, [64])
    ix.set_ef(64)
    ox.set_ef(64)
    want = ox.search_ids(q, 10)["labels"]
    ix.set_capacity(cand_cap=80, hash_slots=256)
    for exact in (True, False):
        ix.set_exact_order(exact)
        r = ix.search_ids(q, 10, want_stats=True)
        if exact:  # the strict kernel keeps its heap in LDS only; the fast kernel spills to its tier-2 region instead
            assert (r["stats"][:, 3] == 2).sum() > 0, "expected some queries to take the whole-CU re-run"
        assert np.array_equal(np.sort(r["labels"], axis=1), np.sort(want, axis=1))
        if exact:
            assert np.array_equal(r["labels"], want)
    ix.set_capacity(0, 0)


def test_k_larger_than_ef_and_k_equals_n(hs, oracle, tmp_path):
    base = mixture(500, 16, 33)
    q = mixture(20, 16, 34)
    ix, ox = _slim_case(hs, oracle, tmp_path, base, q, 16, L2, 8, 50, [4], k=20, fast=False)  # ef = max(ef_, k)
    assert ix.info()["n"] == 500


@pytest.mark.parametrize("dim,metric", [(64, L2), (512, L2), (1024, L2), (512, IP), (1024, IP), (1536, IP), (160, L2), (384, IP)])
def test_slim_compiled_in_and_runtime_dims(hs, oracle, tmp_path, dim, metric):
    """dim % 16 == 0 shapes: the ones with their own kernel instantiation (64, 512, 1024 / IP 512, 1024, 1536) and two that
    take the runtime-dim kernel (160, 384)."""
    base = mixture(2500, dim, 71, lo=-1, hi=1, sigma=0.5)
    q = mixture(60, dim, 72, lo=-1, hi=1, sigma=0.5)
    if metric == IP:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    _slim_case(hs, oracle, tmp_path, base.astype(np.float32), q.astype(np.float32), dim, metric, 16, 100, [32, 100, 200])


@pytest.mark.parametrize("dim", (100, 200, 300, 70, 36, 12))
def test_slim_dims_off_the_simd16_path(hs, oracle, tmp_path, dim):
    """dim % 16 != 0: the reference's SIMD4 (d=100) and SIMD16+residual (d=70) L2 recipes, strict and fast kernels."""
    base = mixture(6000, dim, 61, integer=True)
    q = mixture(100, dim, 62, integer=True)
    _slim_case(hs, oracle, tmp_path, base, q, dim, L2, 16, 100, [32, 100])


@pytest.mark.parametrize("dim", (100, 200, 300, 70, 36, 12, 7, 3))
def test_slim_inner_product_off_the_simd16_path(hs, oracle, tmp_path, dim):
    """InnerProductSpace with dim % 16 != 0 (space_ip.h:374-382): SIMD4ExtAVX (100), SIMD16 + scalar rest (70), SIMD4 +
    scalar rest (7), scalar (3) -- strict and fast kernels, unit-norm rows."""
    base = mixture(5000, dim, 63, lo=-1, hi=1, sigma=0.5)
    q = mixture(100, dim, 64, lo=-1, hi=1, sigma=0.5)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    _slim_case(hs, oracle, tmp_path, base.astype(np.float32), q.astype(np.float32), dim, IP, 16, 100, [32, 100])


def test_load_from_memory_equals_load_from_file(hs, oracle, tmp_path):
    """hs_index_load_mem: the serialized bytes in a host buffer (vanilla, Slim, SlimQ) give the same index as the file; a
    truncated buffer reports the reference's corruption error."""
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    hp = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    q = g["queries"]
    a, b = hs.Index(hp, hs.HS_KIND_HNSW, 32), hs.Index(open(hp, "rb").read(), hs.HS_KIND_HNSW, 32)
    for ix in (a, b):
        ix.set_ef(32)
    ra, rb = a.search_pq(q, 10), b.search_pq(q, 10)
    assert np.array_equal(ra["labels"], rb["labels"]) and ra["dists"].tobytes() == rb["dists"].tobytes()
    assert a.info() == b.info()
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(hp, sp, 32)
    a, b = hs.Index(sp, hs.HS_KIND_SLIM, 32), hs.Index(open(sp, "rb").read(), hs.HS_KIND_SLIM, 32)
    for ix in (a, b):
        ix.set_ef(48)
    assert np.array_equal(a.search_ids(q, 10)["labels"], b.search_ids(q, 10)["labels"])
    base = mixture(1500, 64, 91)
    h64, s64, q64 = str(tmp_path / "h64.bin"), str(tmp_path / "s64.bin"), str(tmp_path / "q64.bin")
    hs.build_hnsw(base, h64, M=8, ef_construction=60, threads=8)
    hs.convert_slim(h64, s64, 64, threads=8)
    hs.convert_slimq(s64, 0, 64, base[:4].copy(), q64, threads=8)
    qq = mixture(50, 64, 92)
    outs = []
    for src in (q64, open(q64, "rb").read()):
        ix = hs.Index(src, hs.HS_KIND_SLIMQ, 64)
        ix.slimq_set_dataset(base)
        ix.slimq_set_tconst(7.0)
        ix.set_ef(40)
        outs.append(ix.slimq_search(qq, 10)["labels"])
    assert np.array_equal(outs[0], outs[1])
    with pytest.raises(hs.HsError, match="corrupted or unsupported"):
        hs.Index(open(sp, "rb").read()[:3000], hs.HS_KIND_SLIM, 32)


def test_index_size_is_the_references_formula(hs, tmp_path):
    """hs_info.index_size = indexSize() of the reference classes (hnswalg.h:1533-1547, hnswalg_slim.h:2435-2444), checked
    against an independent parse of the files."""
    import struct
    hp = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    raw = open(hp, "rb").read()
    off0, max_el, cnt, spe, label_off, off_data, maxlevel, ep, maxM, maxM0, M, mult, efc = struct.unpack_from("<6QiI3QdQ", raw, 0)
    pos = struct.calcsize("<6QiI3QdQ") + cnt * spe
    size_links_up = 4 + 4 * maxM
    want = max_el * (spe - 32 * 4 - 8) + max_el * 4
    for _ in range(cnt):
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4 + sz
        want += 8 + (sz + 1 if sz else 0)
        assert sz % size_links_up == 0
    assert pos == len(raw)
    assert hs.Index(hp, hs.HS_KIND_HNSW, 32).info()["index_size"] == want
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(hp, sp, 32)
    raw = open(sp, "rb").read()
    n, spe = struct.unpack_from("<2Q", raw, 0)
    el = np.frombuffer(raw, np.uint8, n * spe, 93).reshape(n, spe)
    level = el[:, 0:4].copy().view(np.int32)[:, 0].astype(np.int64)
    total = el[:, 4:8].copy().view(np.uint32)[:, 0].astype(np.int64)
    assert hs.Index(sp, hs.HS_KIND_SLIM, 32).info()["index_size"] == 16 * n + int((2 * level + 4 * total).sum())


def test_cpp_facade_matches_oracle(hs, oracle, tmp_path):
    """hnswlib-API caller (tests/facade_smoke.cpp): per-query searchKnn(q,k,tableint*) loop + searchKnnBatch on
    a Slim index, and searchKnnCloserFirst on a vanilla index, against the oracle."""
    import subprocess
    from hsutil import ROOT
    exe = os.path.join(ROOT, "hnsw-slim_amd", "facade_smoke")
    assert os.path.exists(exe)
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    q = np.ascontiguousarray(g["queries"][:40])
    qf = str(tmp_path / "q.f32")
    q.tofile(qf)
    hp = os.path.join(GOLDEN, "l2_cont_d32.hnsw.bin")
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(hp, sp, 32)
    out = str(tmp_path / "o.bin")
    subprocess.check_call([exe, "slim", sp, "32", qf, "40", "10", "48", out])
    got = np.fromfile(out, np.uint32).reshape(2, 40, 10)
    ox = oracle.load(sp, "slim", L2, 32)
    ox.set_ef(48)
    want = ox.search_ids(q, 10)["labels"]
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
    subprocess.check_call([exe, "hnsw", hp, "32", qf, "40", "10", "48", out])
    ov = oracle.load(hp, "hnsw", L2, 32)
    ov.set_ef(48)
    w = ov.search_pq(q, 10)
    rec = np.fromfile(out, np.dtype([("c", "<u4"), ("p", [("d", "<f4"), ("l", "<u8")], 10)]))
    assert np.array_equal(rec["c"], w["cnt"])
    # closer-first order == reverse of the priority_queue pop order
    assert np.array_equal(rec["p"]["l"], w["labels"][:, ::-1])
    assert rec["p"]["d"].tobytes() == np.ascontiguousarray(w["dists"][:, ::-1]).tobytes()


def test_ef_equal_k_on_the_fast_kernel(hs, oracle, tmp_path):
    """The reference's defaults (ef_ = 10, K = 10): ef == k, nothing is selected at the end, so ties across the
    capacity boundary decide the answer -- the fast kernel must notice them and replay (tie-heavy integer data)."""
    g = np.load(os.path.join(GOLDEN, "l2_int_d16.npz"))
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(os.path.join(GOLDEN, "l2_int_d16.hnsw.bin"), sp, 16)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, 16)
    ox = oracle.load(sp, "slim", L2, 16)
    replays = 0
    for ef, k in ((10, 10), (5, 32), (64, 64)):
        ix.set_ef(ef); ox.set_ef(ef)
        want = ox.search_ids(g["queries"], k)
        ok = want["raw_sz"] >= k
        got = ix.search_ids(g["queries"], k, want_stats=True)
        assert np.array_equal(np.sort(got["labels"][ok], axis=1), np.sort(want["labels"][ok], axis=1)), (ef, k)
        assert np.array_equal(got["stats"][:, :3], want["counters"][:, :3])
        replays += int((got["stats"][:, 3] == 1).sum())
        wp, gp = ox.search_pq(g["queries"], k), ix.search_pq(g["queries"], k)
        assert _pq_sorted(gp["dists"], gp["labels"], gp["cnt"]) == _pq_sorted(wp["dists"], wp["labels"], wp["cnt"])
    assert replays > 0, "expected boundary ties on this data"


def test_large_batch_runs_as_launch_groups(hs, oracle, tmp_path):
    """More queries than one launch group holds (32768): consecutive groups on the stream, same answers."""
    base = mixture(3000, 32, 81, integer=True)
    q = mixture(70001, 32, 82, integer=True)
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    hs.build_hnsw(base, hp, M=8, ef_construction=60, threads=8)
    hs.convert_slim(hp, sp, 32, threads=8)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, 32)
    ox = oracle.load(sp, "slim", L2, 32)
    ix.set_ef(40); ox.set_ef(40)
    got = ix.search_ids(q, 10, want_stats=True)
    want = ox.search_ids(q, 10, threads=8)
    assert np.array_equal(np.sort(got["labels"], axis=1), np.sort(want["labels"], axis=1))
    assert np.array_equal(got["stats"][:, :3], want["counters"][:, :3])
    pq, wq = ix.search_pq(q[32000:33500], 10), ox.search_pq(q[32000:33500], 10, threads=8)
    assert _pq_sorted(pq["dists"], pq["labels"], pq["cnt"]) == _pq_sorted(wq["dists"], wq["labels"], wq["cnt"])


@pytest.mark.parametrize("metric,dim", [(L2, 32), (IP, 48)])
def test_ordered_two_launch_pass(hs, oracle, tmp_path, metric, dim):
    """From 6144 queries per launch the fast / lean pass runs as three launches -- upper-level descent, queries ordered by
    the distance of their level-0 entry, level-0 search in that order: same answers and counters as the one-launch pass,
    for both index kinds, the (q,k) overloads that tag the enter point and a large ef (the lean kernel under this order:
    test_kernel_variants_parity_under_env)."""
    base = mixture(6000, dim, 91)
    q = mixture(7000, dim, 92)
    hp, sp = str(tmp_path / "h.bin"), str(tmp_path / "s.bin")
    hs.build_hnsw(base, hp, metric=metric, M=8, ef_construction=60, threads=8)
    hs.convert_slim(hp, sp, dim, metric=metric, threads=8)
    for path, kind, okind in ((sp, hs.HS_KIND_SLIM, "slim"), (hp, hs.HS_KIND_HNSW, "hnsw")):
        ix = hs.Index(path, kind, dim, metric)
        ox = oracle.load(path, okind, metric, dim)
        for ef in (24, 200):
            ix.set_ef(ef); ox.set_ef(ef)
            if okind == "slim":   # the id-array overload exists on the Slim class only
                got = ix.search_ids(q, 10, want_stats=True, want_dists=True)
                small = ix.search_ids(q[:500], 10, want_stats=True, want_dists=True)      # below the threshold: one launch
                want = ox.search_ids(q, 10, threads=8)
                assert np.array_equal(np.sort(got["labels"], axis=1), np.sort(want["labels"], axis=1)), (okind, ef)
                assert np.array_equal(got["stats"][:, :3], want["counters"][:, :3]), (okind, ef)
                assert np.array_equal(got["labels"][:500], small["labels"]) and got["dists"][:500].tobytes() == small["dists"].tobytes()
            pq, wq = ix.search_pq(q, 10, want_stats=True), ox.search_pq(q, 10, threads=8)
            assert np.array_equal(pq["cnt"], wq["cnt"])
            assert np.array_equal(pq["stats"][:, :3], wq["counters"][:, :3]), (okind, ef)
            assert _pq_sorted(pq["dists"], pq["labels"], pq["cnt"]) == _pq_sorted(wq["dists"], wq["labels"], wq["cnt"]), (okind, ef)


def test_cpp_facade_build_then_search(hs, oracle, tmp_path):
    """A caller that BUILDS through the hnswlib API (ctor, addPoint loop, saveIndex, convertFromHNSW, saveIndex, setEf,
    searchKnn -- hnsw_strategy.h / hnsw_slim_strategy.h): the saved files are the harness's (byte-identical to the
    reference's serial build), the search results the oracle's."""
    import subprocess
    from hsutil import ROOT
    exe = os.path.join(ROOT, "hnsw-slim_amd", "facade_smoke")
    g = np.load(os.path.join(GOLDEN, "l2_cont_d32.npz"))
    base, q = np.ascontiguousarray(g["base"][:800]), np.ascontiguousarray(g["queries"][:25])
    bf, qf, out, hp, sp = (str(tmp_path / f) for f in ("b.f32", "q.f32", "o.bin", "h.bin", "s.bin"))
    base.tofile(bf); q.tofile(qf)
    subprocess.check_call([exe, "build", bf, "32", qf, "25", "10", "40", out, "800", hp, sp])
    mine_h, mine_s = str(tmp_path / "h2.bin"), str(tmp_path / "s2.bin")
    hs.build_hnsw(base, mine_h, M=16, ef_construction=100, branching_factor="4", seed=100, threads=1)
    hs.convert_slim(mine_h, mine_s, 32)
    assert open(hp, "rb").read() == open(mine_h, "rb").read()
    assert open(sp, "rb").read() == open(mine_s, "rb").read()
    ox = oracle.load(sp, "slim", L2, 32)
    ox.set_ef(40)
    want = ox.search_ids(q, 10)["labels"]
    raw = np.fromfile(out, np.uint32)
    assert np.array_equal(raw[:250].reshape(25, 10), want) and raw[250] == 10


@pytest.mark.parametrize("name,dim", [("l2_cont_d32_del", 32), ("l2_int_d16_del", 16)])
def test_delete_marks_vs_compiled_reference(hs, name, dim):
    """Index saved by the reference after markDelete: both kernels take the !bare_bone_search branch."""
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ix = hs.Index(os.path.join(GOLDEN, f"{name}.hnsw.bin"), hs.HS_KIND_HNSW, dim)
    assert ix.info()["has_deleted"] == 1
    k = int(g["k"])
    for exact in (True, False):
        ix.set_exact_order(exact)
        for ef in g["efs"]:
            ef = int(ef)
            ix.set_ef(ef)
            r = ix.search_pq(g["queries"], k, want_stats=True)
            assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
            assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(g[f"ef{ef}_dists"], g[f"ef{ef}_labels"], g[f"ef{ef}_cnt"])
            assert np.array_equal(r["stats"][:, 0], g[f"ef{ef}_calls"])


def test_tiny_and_degenerate_inputs(hs, oracle, tmp_path):
    """n < k (priority_queue overload returns fewer than k), a single-element index, nq = 0."""
    for n in (1, 7):
        base = mixture(n, 16, 40 + n)
        hp, sp = str(tmp_path / f"h{n}.bin"), str(tmp_path / f"s{n}.bin")
        hs.build_hnsw(base, hp, M=4, ef_construction=10, threads=1)
        hs.convert_slim(hp, sp, 16)
        q = mixture(5, 16, 50)
        for kind, path, okind in ((hs.HS_KIND_HNSW, hp, "hnsw"), (hs.HS_KIND_SLIM, sp, "slim")):
            ix = hs.Index(path, kind, 16)
            ox = oracle.load(path, okind, L2, 16)
            ix.set_ef(10)
            ox.set_ef(10)
            r = ix.search_pq(q, 10)
            o = ox.search_pq(q, 10)
            assert np.array_equal(r["cnt"], o["cnt"]) and np.all(r["cnt"] == n)
            assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(o["dists"], o["labels"], o["cnt"])
            assert np.all(r["labels"][:, n:] == np.iinfo(np.uint64).max)
            empty = ix.search_pq(q[:0], 10)
            assert empty["labels"].shape == (0, 10)


@pytest.mark.parametrize("name,dim", [("l2_cont_d32", 32), ("l2_int_d16_del", 16)])
def test_filtered_search_vs_compiled_reference(hs, oracle, tmp_path, name, dim):
    """searchKnn(q, k, isIdAllowed): vanilla index against the compiled reference's output; the Slim
    conversion of the same graph against the oracle's restatement of hnswalg_slim.h:1783-1905."""
    g = np.load(os.path.join(GOLDEN, f"{name}_filter.npz"))
    hp = os.path.join(GOLDEN, f"{name}.hnsw.bin")
    ix = hs.Index(hp, hs.HS_KIND_HNSW, dim)
    allowed = (ix.labels() % int(g["mod"]) != int(g["rem"])).astype(np.uint8)
    k = int(g["k"])
    for ef in g["efs"]:
        ef = int(ef)
        ix.set_ef(ef)
        r = ix.search_filtered(g["queries"], k, allowed, want_stats=True)
        assert np.array_equal(r["cnt"], g[f"ef{ef}_cnt"])
        assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(g[f"ef{ef}_dists"], g[f"ef{ef}_labels"], g[f"ef{ef}_cnt"])
        assert np.array_equal(r["stats"][:, 0], g[f"ef{ef}_calls"])
    sp = str(tmp_path / "s.bin")
    hs.convert_slim(hp, sp, dim)
    sx = hs.Index(sp, hs.HS_KIND_SLIM, dim)
    ox = oracle.load(sp, "slim", L2, dim)
    ox.set_filter(allowed)
    for ef in (10, 40):
        sx.set_ef(ef)
        ox.set_ef(ef)
        r = sx.search_filtered(g["queries"], k, allowed, want_stats=True)
        o = ox.search_pq(g["queries"], k)
        assert np.array_equal(r["cnt"], o["cnt"])
        assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(o["dists"], o["labels"], o["cnt"])
        assert np.array_equal(r["stats"][:, :3], o["counters"][:, :3])


def test_wide_graphs_large_ef_large_k(hs, oracle, tmp_path):
    """M=32 (level-0 tiles of 64 ids), M=40 (degree 80: no tile, strict kernel answers), ef up to 400
    (8 result slots per lane), k=100."""
    base = mixture(5000, 32, 61, integer=True)
    q = mixture(100, 32, 62, integer=True)
    _slim_case(hs, oracle, tmp_path, base, q, 32, L2, 32, 100, [400], k=100, low_degree_m0=24)
    ix = hs.Index(str(tmp_path / "s.bin"), hs.HS_KIND_SLIM, 32)
    assert 32 < ix.info()["max_degree0"] <= 64
    # vanilla index with degree up to 80 (> 64): both modes fall back to the CSR path of the strict kernel
    hp = str(tmp_path / "h40.bin")
    hs.build_hnsw(base, hp, M=40, ef_construction=100, threads=8)
    vx = hs.Index(hp, hs.HS_KIND_HNSW, 32)
    assert vx.info()["max_degree0"] > 64
    ov = oracle.load(hp, "hnsw", L2, 32)
    for ef in (20, 150):
        vx.set_ef(ef)
        ov.set_ef(ef)
        r = vx.search_pq(q, 10, want_stats=True)
        o = ov.search_pq(q, 10)
        assert _pq_sorted(r["dists"], r["labels"], r["cnt"]) == _pq_sorted(o["dists"], o["labels"], o["cnt"])
        assert np.array_equal(r["stats"][:, :3], o["counters"][:, :3])
