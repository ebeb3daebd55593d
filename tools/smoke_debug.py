"""smoke() with a diff dump (debug aid)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hsutil import Oracle, load_product, mixture
hs = load_product()
base = mixture(4000, 128, 1, integer=True); q = mixture(64, 128, 2, integer=True)
with tempfile.TemporaryDirectory() as tmp:
    hp, sp = os.path.join(tmp, "h.bin"), os.path.join(tmp, "s.bin")
    hs.build_hnsw(base, hp, M=16, ef_construction=100, threads=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
    hs.convert_slim(hp, sp, 128, threads=8)
    ix = hs.Index(sp, hs.HS_KIND_SLIM, 128); ox = Oracle().load(sp, "slim", 0, 128)
    for ef in (64, 32, 100):
        ix.set_ef(ef); ox.set_ef(ef)
        for exact in (0, 1):
            ix.set_exact_order(bool(exact))
            got = ix.search_ids(q, 10, want_stats=True); want = ox.search_ids(q, 10)
            bad = [i for i in range(64) if not np.array_equal(got["labels"][i], want["labels"][i])]
            badset = [i for i in bad if set(got["labels"][i].tolist()) != set(want["labels"][i].tolist())]
            print(f"ef={ef} exact={exact}: {len(bad)} rows differ in order, {len(badset)} as sets; stats equal {np.array_equal(got['stats'][:, :3], want['counters'][:, :3])}; passes {np.bincount(got['stats'][:,3], minlength=3).tolist()}")
            for i in badset[:2]:
                print("   q", i, "gpu", got["labels"][i].tolist(), "\n       ora", want["labels"][i].tolist(), "pass", got["stats"][i, 3])
                d = ((base[want["labels"][i]] - q[i]) ** 2).sum(1); d2 = ((base[got["labels"][i]] - q[i]) ** 2).sum(1)
                print("       ora d", d.tolist(), "\n       gpu d", d2.tolist())
