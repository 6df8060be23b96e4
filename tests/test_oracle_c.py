"""The C oracle (oracle/iss_oracle.c) against the goldens and the numpy oracle."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle as corc
from oracle import ref_numpy as orc

G = load_golden()
RT = dict(rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("case", G.cases("iss"), ids=lambda c: c["name"])
def test_iss_cases(case):
    X = G.x_of(case)
    lookup, total = orc._weight_lookup(case.get("weighting"), X, X)
    out = corc.iss_transform(X, case["words"], case["mode"], case["alphas"],
                             lookup, total, semiring=case.get("semiring", "Reals"))
    if "series" in case:
        out = out[:, case["series"], :]
    np.testing.assert_allclose(out, G[case["out"]], **RT)


def test_operator_entry():
    Z = G["U_6_3_40"]
    word = np.array(orc.parse_word("[12][2][33]"), dtype=np.int32)
    alpha = np.array([.6, .2, .5], dtype=np.float32)
    lk = G["op/lookup"]
    np.testing.assert_allclose(corc.iterated_sum_fast(Z, word, alpha, lk, 2, False),
                               G["op/fast_nontotal_E2"], **RT)
    np.testing.assert_allclose(corc.iterated_sum_fast(Z, word, alpha, lk, 3, True),
                               G["op/fast_total_E3"], **RT)
    with pytest.raises(ValueError):
        corc.iterated_sum_fast(Z[:, :2], word, alpha, lk, 2, False)


def test_small_kernels():
    X = G["U_6_3_40"]
    np.testing.assert_array_equal(corc.increments(X, 3), orc.increments(X, 3))
    np.testing.assert_array_equal(corc.l1_sum(X), orc.l1_sum(X))
    A = G["S_8_50"]
    cuts = orc.transformed_cuts(8, 50, [20, -1])
    q = np.array([-np.inf, -0.3, 0.0, 0.4, np.inf])
    np.testing.assert_array_equal(corc.npi_backend(A, cuts, q), orc.npi_backend(A, cuts, q))
    np.testing.assert_allclose(corc.mpi_backend(A, cuts, q), orc.mpi_backend(A, cuts, q), **RT)
    np.testing.assert_array_equal(corc.end_transform(A, cuts), orc.end_transform(A, cuts))


def test_threads_agree():
    X = G["U_7_2_129"]
    W = G.manifest["words"]["3,2"]["words"]
    a = corc.iss_transform(X, W, "EXTENDED", nthreads=1)
    b = corc.iss_transform(X, W, "EXTENDED", nthreads=4)
    np.testing.assert_array_equal(a, b)
