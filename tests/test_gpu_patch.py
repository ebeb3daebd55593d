"""patchFromStream on a device-resident index (hs_index_patch): an index grown by further insertions, shipped as the reference's
diff stream (genPatch wire format), must answer exactly like the index loaded whole."""
import os

import numpy as np
import pytest

from hsutil import Oracle, load_chal_encode, load_product, mixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,n0,delta,integer", [(32, 3000, 400, False), (128, 8000, 1500, True)])
def test_patched_index_equals_index_loaded_whole(tmp_path, dim, n0, delta, integer):
    hs, ce, O = load_product(), load_chal_encode(), Oracle()
    base = mixture(n0 + delta, dim, 17, integer=integer)
    files = {}
    for tag, n in (("old", n0), ("new", n0 + delta)):
        hp, sp = str(tmp_path / f"{tag}.hnsw"), str(tmp_path / f"{tag}.slim")
        hs.build_hnsw(base[:n], hp, M=16, ef_construction=100, threads=1)   # serial: the first n0 insertions are the same in both
        hs.convert_slim(hp, sp, dim, threads=1)
        files[tag] = open(sp, "rb").read()
    patch, n_changed, n_added = ce.make_patch(files["old"], files["new"], dim, to_add=True)
    assert n_added == delta and n_changed > 0
    want_file = str(tmp_path / "expect.slim")
    open(want_file, "wb").write(ce.with_entry_of(files["new"], files["old"]))
    q = mixture(300, dim, 18, integer=integer)
    ix = hs.Index(str(tmp_path / "old.slim"), hs.HS_KIND_SLIM, dim, max_elements=n0 + delta + 16)
    ix.set_ef(48)
    before = ix.search_ids(q, 10)["labels"]
    ix.patch(patch, to_add=True)
    assert ix.info()["n"] == n0 + delta
    ref = hs.Index(want_file, hs.HS_KIND_SLIM, dim)
    ox = O.load(want_file, "slim", 0, dim)
    for ef in (10, 48, 100):
        for x in (ix, ref, ox):
            x.set_ef(ef)
        for exact in (True, False):
            ix.set_exact_order(exact); ref.set_exact_order(exact)
            a, b = ix.search_ids(q, 10, want_dists=True, want_stats=True), ref.search_ids(q, 10, want_dists=True, want_stats=True)
            assert np.array_equal(a["labels"], b["labels"]) and a["dists"].tobytes() == b["dists"].tobytes()
            assert np.array_equal(a["stats"][:, :3], b["stats"][:, :3])
        o = ox.search_ids(q, 10)
        assert np.array_equal(np.sort(a["labels"], 1), np.sort(o["labels"], 1)) and np.array_equal(a["stats"][:, :3], o["counters"][:, :3])
    ix.set_ef(48)
    assert not np.array_equal(ix.search_ids(q, 10)["labels"], before), "the patch changed nothing?"
    # refusals: not patchable / over capacity / truncated stream leaves the index as it was
    with pytest.raises(hs.HsError):
        ref.patch(patch)
    small = hs.Index(str(tmp_path / "old.slim"), hs.HS_KIND_SLIM, dim, max_elements=n0 + 1)
    with pytest.raises(hs.HsError):
        small.patch(patch, to_add=True)
    again = hs.Index(str(tmp_path / "old.slim"), hs.HS_KIND_SLIM, dim, max_elements=n0 + delta)
    again.set_ef(48)
    with pytest.raises(hs.HsError):
        again.patch(patch[: len(patch) // 2], to_add=True)
    assert np.array_equal(again.search_ids(q, 10)["labels"], before)
