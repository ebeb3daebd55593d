"""RaBitQ pieces of the SlimQ path (product host code) against outputs of the compiled rabitqlib.

The fixture tests/golden/rabitq_ref.npz was written by tests/golden/make_golden.py from oracle/_ref/ref_rabitq,
a driver over the reference's unmodified third_party/rabitqlib headers.  Integer outputs (sign codes, 4-bit
query bit planes) and the rotation must match bit for bit; the float factors go through Eigen reductions
whose summation order depends on alignment and vector width (and, for the inner-product metric, cancel), so
they are pinned to 1e-4 of the column's magnitude.
"""
import os

import numpy as np
import pytest

from hsutil import GOLDEN, load_product

G = np.load(os.path.join(GOLDEN, "rabitq_ref.npz"))
DIMS = (128, 96, 768)


@pytest.fixture(scope="module")
def P():
    return load_product()


@pytest.mark.parametrize("dim", DIMS)
def test_rotation_bit_exact(P, dim):
    p = f"d{dim}_"
    for src, dst in (("x", "rx"), ("q", "rq")):
        got = P.rabitq_rotate(dim, G[p + "flip"], G[p + src])
        assert np.array_equal(got.view(np.uint32), G[p + dst].view(np.uint32))


@pytest.mark.parametrize("dim", DIMS)
def test_data_codes_and_factors(P, dim):
    p = f"d{dim}_"
    codes, fac = P.rabitq_quantize_data(G[p + "rx"], G[p + "rc"], int(G[p + "metric"]))
    assert np.array_equal(codes, G[p + "codes"])
    scale = np.abs(G[p + "fac"]).max(axis=0, keepdims=True)
    assert np.all(np.abs(fac - G[p + "fac"]) <= 1e-4 * scale)


@pytest.mark.parametrize("dim", DIMS)
def test_query_preparation(P, dim):
    p = f"d{dim}_"
    q3, bins = P.rabitq_prepare_query(G[p + "rq"], float(G[p + "t_const"]))
    scale = np.abs(G[p + "q3"]).max(axis=0, keepdims=True)
    assert np.all(np.abs(q3 - G[p + "q3"]) <= 1e-4 * scale)
    assert np.array_equal(bins, G[p + "bins"])


@pytest.mark.parametrize("dim", DIMS)
def test_estimator_exact_given_reference_operands(P, dim):
    """With the reference's own codes, factors and query constants the estimator is pure fp32 arithmetic in a
    fixed order plus integer popcounts: bit-exact."""
    p = f"d{dim}_"
    est = P.rabitq_estimate(G[p + "codes"], G[p + "fac"], G[p + "q3"], G[p + "bins"], G[p + "g_add"], G[p + "g_err"])
    assert np.array_equal(est.view(np.uint32), G[p + "est"].view(np.uint32))
