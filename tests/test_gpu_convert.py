"""convertFromHNSW on the GPU (hs_convert_slim_gpu, csrc/convert_gpu.hip) against the CPU harness (hs_convert_slim, threads=1):
the output FILE must be byte-identical -- every distance, every by-distance std::sort (libstdc++ tie order included), every
pruning decision, the reverse-edge union and the re-prune of over-full lists taken the same way."""
import os

import numpy as np
import pytest

from hsutil import GOLDEN, load_product, mixture

pytestmark = pytest.mark.gpu
L2, IP = 0, 1


@pytest.fixture(scope="module")
def hs():
    m = load_product()
    assert m.device_count() > 0
    return m


def _same_file(hs, hp, dim, metric, tmp_path, **kw):
    a, b = str(tmp_path / "cpu.slim"), str(tmp_path / "gpu.slim")
    hs.convert_slim(hp, a, dim, metric=metric, threads=1, **kw)
    used, ms = hs.convert_slim_gpu(hp, b, dim, metric=metric, **kw)
    assert used, "the GPU path declined this shape"
    assert open(a, "rb").read() == open(b, "rb").read()
    return ms


@pytest.mark.parametrize("name,metric,dim", [("l2_cont_d32", L2, 32), ("l2_int_d16", L2, 16), ("ip_d48", IP, 48), ("l2_cont_d20", L2, 20),
                                             ("l2_cont_d21", L2, 21), ("l2_cont_d10", L2, 10), ("ip_d20", IP, 20), ("ip_d21", IP, 21),
                                             ("ip_d10", IP, 10), ("l2_int_d16_del", L2, 16)])
def test_gpu_convert_is_byte_identical_on_golden_graphs(hs, tmp_path, name, metric, dim):
    hp = os.path.join(GOLDEN, f"{name}.hnsw.bin")
    _same_file(hs, hp, dim, metric, tmp_path)
    _same_file(hs, hp, dim, metric, tmp_path, threshold_level=1)
    _same_file(hs, hp, dim, metric, tmp_path, top_degree_M0=16, low_degree_m0=4, top_degree_M=8, low_degree_m=2, top_degree_percent=0.3)


@pytest.mark.parametrize("dim,metric,integer", [(128, L2, True), (96, L2, False), (64, IP, False), (100, L2, True)])
def test_gpu_convert_m16_graphs_with_ties(hs, tmp_path, dim, metric, integer):
    """M=16: level-0 lists of up to 32 ids (std::sort's introsort branch), tiny integer range => many equal distances; hubs whose
    reverse-edge union exceeds the capacity get re-pruned."""
    if integer:
        base = mixture(30000, dim, 5, lo=0, hi=6, sigma=1.5, integer=True, n_clusters=8)
    else:
        base = mixture(30000, dim, 6, lo=-1, hi=1, sigma=0.4, n_clusters=8)
    if metric == IP:
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    hp = str(tmp_path / "h.bin")
    hs.build_hnsw(base.astype(np.float32), hp, metric=metric, M=16, ef_construction=100, threads=8)
    _same_file(hs, hp, dim, metric, tmp_path)
    _same_file(hs, hp, dim, metric, tmp_path, low_degree_m0=24, top_degree_percent=0.1)


def test_gpu_convert_declines_wide_graphs(hs, tmp_path):
    base = mixture(3000, 32, 7)
    hp, out = str(tmp_path / "h40.bin"), str(tmp_path / "o.slim")
    hs.build_hnsw(base, hp, M=40, ef_construction=80, threads=8)     # maxM0 = 80 > 32: CPU path, same file
    used, _ = hs.convert_slim_gpu(hp, out, 32)
    assert not used
    ref = str(tmp_path / "r.slim")
    hs.convert_slim(hp, ref, 32, threads=1)
    assert open(out, "rb").read() == open(ref, "rb").read()
