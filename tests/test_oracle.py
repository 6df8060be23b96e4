"""The oracle (oracle/ref_numpy.py) pinned against vectors produced by the
reference itself (tests/golden/make_golden.py) and the hand-computed vectors of
the reference's own tests."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_numpy as orc

G = load_golden()
RT = dict(rtol=1e-12, atol=1e-12)


def _lookup(case, X, X_raw=None):
    return orc._weight_lookup(case.get("weighting"), X, X if X_raw is None else X_raw)


def test_arctic_x1_hand_computed():
    # reference tests/signature/test_semiring.py:10-33
    out = orc.iss_transform(G["X_1"], ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"],
                            semiring="Arctic")
    np.testing.assert_allclose(out, G["iss/arctic_x1_six_words_expected"], rtol=1e-12)


def test_x1_hand_computed():
    # reference tests/signature/test_simple.py:11-34
    out = orc.iss_transform(G["X_1"], ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"])
    np.testing.assert_allclose(out, G["iss/x1_six_words_expected"], rtol=1e-12)


@pytest.mark.parametrize("key", sorted(G.manifest["words"]))
def test_of_weight_and_plan(key):
    w, d = map(int, key.split(","))
    ent = G.manifest["words"][key]
    mine = orc.of_weight_strings(w, d)
    assert sorted(mine) == sorted(ent["words"])
    assert mine == ent["words"]          # same CPython set order here
    assert orc.cache_plan(ent["words"]) == ent["plan"]
    assert [orc.parse_word(s) for s in ent["words"]] == ent["exps"]


def test_cache_plan_golden():
    for ent in G.manifest["cacheplan"]:
        plan = orc.cache_plan(ent["words"])
        assert plan == ent["plan"]
        assert orc.plan_labels(ent["words"], plan) == ent["labels"]
    assert G.manifest["cacheplan"][0]["plan"] == [4, 5, 2, 3, 3, 1, 1, 1, 2]


def test_parse():
    for s, rows in G.manifest["parse"].items():
        assert orc.parse_word(s) == rows
    with pytest.raises(ValueError):
        orc.parse_word("[1][")


@pytest.mark.parametrize("case", G.cases("iss"), ids=lambda c: c["name"])
def test_iss_cases(case):
    X = G.x_of(case)
    lookup, total = _lookup(case, X)
    out = orc.iss_transform(X, case["words"], case["mode"], case["alphas"],
                            lookup, total, case.get("semiring", "Reals"))
    assert out.shape[0] == case["K"]
    if "series" in case:
        out = out[:, case["series"], :]
    np.testing.assert_allclose(out, G[case["out"]], **RT)


@pytest.mark.parametrize("case", G.manifest.get("iss_argmax", []), ids=lambda c: c["name"])
def test_arctic_argmax_cases(case):
    """Arctic(argmax=True), fruits/iss/semiring.py:239-284: values and back-tracked positions
    of the maxima, bit for bit (max and integer positions are exact)."""
    X = G.x_of(case)
    lookup, _ = _lookup(case, X)
    out = orc.iss_transform(X, case["words"], case["mode"], case["alphas"], lookup,
                            False, "Arctic", argmax=True)
    assert out.shape[0] == case["K"]
    np.testing.assert_array_equal(out, G[case["out"]])


def test_operator_entry():
    Z = G["U_6_3_40"]
    word = np.array(orc.parse_word("[12][2][33]"), dtype=np.int32)
    alpha = np.array([.6, .2, .5], dtype=np.float32)
    lk = orc.lookup_indices(6, 40, True, 2.0)
    np.testing.assert_allclose(lk, G["op/lookup"], **RT)
    np.testing.assert_allclose(
        orc.iterated_sum_fast(Z, word, alpha, lk, 2, False),
        G["op/fast_nontotal_E2"], **RT)
    np.testing.assert_allclose(
        orc.iterated_sum_fast(Z, word, alpha, lk, 3, True),
        G["op/fast_total_E3"], **RT)


@pytest.mark.parametrize("case", G.cases("l1"), ids=lambda c: c["name"])
def test_l1_lookup(case):
    kw = case["kw"]
    out = orc.lookup_l1(G[case["x"]], kw.get("relative", False), kw.get("scale", 50))
    np.testing.assert_allclose(out, G[case["out"]], **RT)


@pytest.mark.parametrize("case", G.cases("inc"), ids=lambda c: c["name"])
def test_inc(case):
    out = orc.inc_transform(G[case["x"]], **case["kw"])
    np.testing.assert_allclose(out, G[case["out"]], **RT)


@pytest.mark.parametrize("case", G.cases("sieve"), ids=lambda c: c["name"])
def test_sieves(case):
    sv = orc.SieveOracle(case["kind"], **case["kw"])
    A = G[case["x"]]
    sv.fit(G[case["fit"]] if case["fit"] else A)
    out = sv.transform(A)
    np.testing.assert_allclose(out, G[case["out"]], **RT)


@pytest.mark.parametrize("case", G.cases("fruit"), ids=lambda c: c["name"])
def test_fruit(case):
    X = G[case["x"]]
    fitted = orc.fruit_fit(case["spec"], X, case["np_seed"])
    out = orc.fruit_transform(case["spec"], fitted, X)
    assert out.shape[1] == case["nfeatures"]
    np.testing.assert_allclose(out, G[case["out"]], **RT)
    if "x_test" in case:
        np.testing.assert_allclose(
            orc.fruit_transform(case["spec"], fitted, G[case["x_test"]]),
            G[case["out_test"]], **RT)


@pytest.mark.parametrize("case", G.manifest.get("coswiss_random", []), ids=lambda c: c["name"])
def test_coswiss_random_variants(case):
    """ffn / dropout CosWISS (fruits/iss/cos.py:51-164) with the weights / indices the
    reference's fit drew; the ffn sums follow numba's sequential np.sum while the goldens
    come from the un-jitted run (numpy's pairwise order): 1e-12 instead of bit equality."""
    kw = case["kw"]
    ffn = (G[case["A"]], G[case["b"]], G[case["C"]]) if "A" in case else None
    drop = G[case["dropout_indices"]] if "dropout_indices" in case else None
    out = orc.coswiss_transform(G[case["x"]], case["words"], case["freqs"], kw.get("exponent", 2),
                                kw.get("total_weighting", False), ffn=ffn, dropout_indices=drop)
    np.testing.assert_allclose(out, G[case["out"]], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("case", G.manifest.get("coswiss", []), ids=lambda c: c["name"])
def test_coswiss(case):
    kw = case["kw"]
    out = orc.coswiss_transform(G[case["x"]], case["words"], case["freqs"],
                                kw.get("exponent", 2), kw.get("total_weighting", False))
    np.testing.assert_allclose(out, G[case["out"]], rtol=1e-11, atol=1e-12)
    for s, w in case["weightings"].items():
        got = orc.coswiss_weightings(len(orc.parse_word(s)), kw.get("exponent", 2),
                                     kw.get("total_weighting", False))
        assert got.tolist() == w


@pytest.mark.parametrize("case", [c for c in G.manifest.get("lookups", []) if c["kind"] == "L2"],
                         ids=lambda c: c["name"])
def test_l2_lookup(case):
    kw = case["kw"]
    out = orc.lookup_l2(G[case["x"]], kw.get("relative", False), kw.get("scale", 50))
    np.testing.assert_allclose(out, G[case["out"]], rtol=1e-12, atol=1e-14)
