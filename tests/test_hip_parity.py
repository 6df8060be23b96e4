"""GPU parity: the HIP path (through the C ABI) against the golden vectors the
reference produced and against the oracle on seeded inputs.

Tolerances: float64 results must agree to 1e-6 relative (BASELINE.json
north_star).  A parallel scan re-associates the sums, so on zero-mean inputs the
comparison is row-norm-wise (|delta| <= 1e-6 * max|ref| per (k, n) row); on the
U[0,1) inputs the reference's own tests use it is also element-wise rtol=1e-6.
Observed differences are ~1e-13.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden, gen_input
from oracle import c_oracle as corc
from oracle import ref_numpy as orc

pytestmark = pytest.mark.gpu

G = load_golden()
RTOL = 1e-6


@pytest.fixture(scope="module")
def fr():
    import fruits_amd
    from fruits_amd import _native as nat
    nat.require_device()
    return fruits_amd


def rowwise_close(got, ref, rtol=RTOL):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape
    flat_g = got.reshape(-1, got.shape[-1])
    flat_r = ref.reshape(-1, ref.shape[-1])
    scale = np.max(np.abs(flat_r), axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    err = np.max(np.abs(flat_g - flat_r) / scale)
    assert err <= rtol, f"row-wise relative error {err:.3e} > {rtol}"
    return err


def make_weighting(fr, spec):
    if spec is None:
        return None
    kw = {k: v for k, v in spec.items() if k != "kind"}
    return getattr(fr.iss.weighting, spec["kind"])(**kw)


def make_iss(fr, case):
    ws = [fr.words.SimpleWord(s) for s in case["words"]]
    if case.get("alphas"):
        for w, a in zip(ws, case["alphas"]):
            if a is not None:
                w.alpha = a
    return fr.ISS(ws, mode=getattr(fr.ISSMode, case["mode"]),
                  semiring=getattr(fr.semiring, case.get("semiring", "Reals"))(),
                  weighting=make_weighting(fr, case.get("weighting")))


@pytest.mark.parametrize("case", G.cases("iss"), ids=lambda c: c["name"])
def test_iss_golden(fr, case):
    X = G.x_of(case)
    iss = make_iss(fr, case)
    out = iss.fit_transform(X)
    assert out.shape == (case["K"], X.shape[0], X.shape[2])
    assert [iss.label(i) for i in range(iss.n_iterated_sums())] == case["labels"]
    if "series" in case:
        out = out[:, case["series"], :]
    ref = G[case["out"]]
    if case.get("semiring") == "Arctic":
        # (max, +): max is associative, the letter sums round like the reference's and
        # the path-length lookups are summed sequentially like np.cumsum: bit-exact
        np.testing.assert_array_equal(out, ref)
        return
    rowwise_close(out, ref)
    if "U_" in case.get("x", "") or case.get("x_gen", {}).get("dist") == "uniform":
        if not any("-" in w for w in case["words"]):
            np.testing.assert_allclose(out, ref, rtol=RTOL)


def test_x1_hand_computed(fr):
    # reference tests/signature/test_simple.py:11-41
    ws = [fr.words.SimpleWord(s) for s in ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"]]
    res = list(fr.ISS(ws).batch_transform(G["X_1"], batch_size=1))
    assert len(res) == 6
    for i, r in enumerate(res):
        np.testing.assert_allclose(G["iss/x1_six_words_expected"][i], r[0], atol=1e-12)
    np.testing.assert_allclose(
        G["iss/x1_six_words_expected"][0],
        fr.ISS([ws[0].copy()]).fit_transform(G["X_1"])[0], atol=1e-12)


def test_arctic_hand_computed_and_identity(fr):
    # reference tests/signature/test_semiring.py:10-33 and test_simple.py:81-88
    ws = [fr.words.SimpleWord(s) for s in ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"]]
    res = list(fr.ISS(ws, semiring=fr.semiring.Arctic()).batch_transform(G["X_1"], batch_size=1))
    for i, r in enumerate(res):
        np.testing.assert_allclose(G["iss/arctic_x1_six_words_expected"][i], r[0], atol=1e-12)
    X = G["X_1"]
    a = fr.ISS([fr.words.SimpleWord("[1][2]")], semiring=fr.semiring.Arctic()).fit_transform(-X)
    b = fr.ISS([fr.words.SimpleWord("[-1][-2]")], semiring=fr.semiring.Arctic()).fit_transform(X)
    np.testing.assert_array_equal(a, b)
    with pytest.raises(NotImplementedError):   # argmax needs EXTENDED mode (iss.py:37-40)
        fr.ISS([fr.words.SimpleWord("[1]")], semiring=fr.semiring.Arctic(argmax=True)).fit_transform(X)


@pytest.mark.parametrize("case", G.manifest.get("iss_argmax", []), ids=lambda c: c["name"])
def test_arctic_argmax_golden(fr, case):
    """Arctic(argmax=True) (fruits/iss/semiring.py:239-284) against the reference's own
    outputs: running maxima and back-tracked positions, bit for bit."""
    X = G.x_of(case)
    ws = [fr.words.SimpleWord(s) for s in case["words"]]
    if case["alphas"] is not None:
        for w, a in zip(ws, case["alphas"]):
            if a is not None:
                w.alpha = a
    iss = fr.ISS(ws, mode=fr.ISSMode.EXTENDED, semiring=fr.semiring.Arctic(argmax=True),
                 weighting=make_weighting(fr, case["weighting"]))
    assert iss.n_iterated_sums() == case["K"]
    out = iss.fit_transform(X)
    np.testing.assert_array_equal(out, G[case["out"]])
    # word by word (the generator of iss.py:152-185) and through the per-word operator
    blocks = list(iss.batch_transform(X, batch_size=1))
    np.testing.assert_array_equal(np.concatenate(blocks, axis=0), out)


@pytest.mark.parametrize("T", [1, 2, 63, 1024, 1500])
def test_arctic_argmax_random(fr, T):
    rng = np.random.default_rng(T)
    X = rng.standard_normal((7, 2, T)).cumsum(axis=2)
    X[1] = np.round(X[1])            # plateaus and exact ties: `>=` keeps the earlier index
    words = ["[1][2][-1]", "[2]", "[12][1][1][2]"]
    iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                 semiring=fr.semiring.Arctic(argmax=True))
    out = iss.fit_transform(X)
    ref = orc.iss_transform(X, words, "EXTENDED", semiring="Arctic", argmax=True)
    np.testing.assert_array_equal(out, ref)
    # a fruit on top: sieves see values and positions alike (materialising path)
    fruit = fr.Fruit()
    fruit.add(iss.copy(), fr.sieving.END, fr.sieving.NPI)
    feats = fruit.fit_transform(X)
    assert feats.shape == (7, 2 * iss.n_iterated_sums())
    np.testing.assert_array_equal(feats[:, 0::2], ref[:, :, -1].T)


@pytest.mark.parametrize("T", [2, 63, 300, 1024, 1500])
@pytest.mark.parametrize("weighting", [None, {"kind": "Indices", "scale": 2.0}])
def test_arctic_argmax_fused(fr, monkeypatch, T, weighting):
    """Arctic(argmax=True) under sieves (fr_pipeline_set_argmax: the running maxima materialised,
    every argmax row formed in LDS for the ops that look at it) against the oracle's
    FruitSlice (fruits/fruit.py:538-550 over semiring.py:239-284) and against the materialising
    path (fr_arctic_argmax + one launch per sieve) with the same thresholds: values, positions,
    their first and second differences, integer and float cuts."""
    if os.environ.get("FRUITS_AMD_FUSED_ARGMAX", "1") == "0" or os.environ.get("FRUITS_AMD_FUSED", "1") == "0":
        pytest.skip("the knob sweep switched the argmax pipeline off")
    rng = np.random.default_rng(T + (7 if weighting else 0))
    X = rng.standard_normal((9, 2, T)).cumsum(axis=2)
    X[1] = np.round(X[1])            # plateaus and exact ties: `>=` keeps the earlier index
    cut = [max(T // 3, 1), -1]
    spec = {"slices": [{
        "preps": [], "iss": [{"words": ["[1][2][-1]", "[2]", "[12][1][1][2]"], "mode": "EXTENDED",
                              "semiring": "Arctic", "argmax": True, "weighting": weighting}],
        "sieves": [{"kind": "NPI", "q": [0.25, 0.5, 1.0]}, {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
                   {"kind": "MPI", "q": [0.3, 1.0], "inc": 2, "cut": cut}, {"kind": "MPI", "q": [-1.0, 0.5]},
                   {"kind": "END", "cut": cut}, {"kind": "NPI", "q": [0.5, 1.0], "cut": [0.5, -1]}],
        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(3)
    fruit.fit(X)
    slc = fruit.get_slice()
    got = fruit.transform(X)
    pipe = slc._fused(T)
    assert pipe is not None and pipe.rows == slc.niteratedsums() == 3 + 6 + 1 + 1 + 4 + 10
    monkeypatch.setenv("FRUITS_AMD_FUSED_ARGMAX", "0")
    slc._fused_cache = {}
    assert slc._fused(T) is None
    mat = fruit.transform(X)
    per_sum = [type(sv).__name__ != "MPI" for sv in slc._sieves for _ in range(sv.nfeatures())]
    counts = np.array(per_sum * slc.niteratedsums())
    np.testing.assert_array_equal(got[:, counts], mat[:, counts])       # counts and picked values
    np.testing.assert_allclose(got[:, ~counts], mat[:, ~counts], rtol=1e-12, atol=1e-12)
    # the oracle, with the GPU's thresholds (the fit is compared on its own elsewhere)
    np.random.seed(3)
    fitted = orc.fruit_fit(spec, X)
    fit_parity(fruit, fitted, "argmax")
    transplant_thresholds(fruit, fitted)
    monkeypatch.setenv("FRUITS_AMD_FUSED_ARGMAX", "1")
    ref = np.nan_to_num(orc.fruit_transform(spec, fitted, X))
    got = fruit.transform(X)
    assert slc._fused(T) is not None
    np.testing.assert_array_equal(got[:, counts], ref[:, counts])
    np.testing.assert_allclose(got[:, ~counts], ref[:, ~counts], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("T", [5, 511, 1024, 1025, 3000])
def test_arctic_ragged_and_long_chains(fr, T):
    rng = np.random.default_rng(T)
    X = rng.standard_normal((4, 2, T))
    words = [str(w) for w in fr.words.alternate_sign(
        [fr.words.SimpleWord(24 * "[1]"), fr.words.SimpleWord(12 * "[1][2]")])]
    for weighting in (None, {"kind": "Indices", "scale": 3.0},
                      {"kind": "Indices", "scale": 3.0, "total": True}):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     semiring=fr.semiring.Arctic(), weighting=make_weighting(fr, weighting))
        lookup, total = orc._weight_lookup(weighting, X, X)
        ref = corc.iss_transform(X, words, "EXTENDED", None, lookup, total, semiring="Arctic")
        np.testing.assert_array_equal(iss.fit_transform(X), ref)


@pytest.mark.parametrize("T", [5, 511, 1024, 1025, 3000])
def test_bayesian_semiring(fr, T):
    # (max, x), fruits/iss/semiring.py:461-571; reference tests/signature/test_weighting.py:220-278
    rng = np.random.default_rng(T)
    X = rng.random((4, 3, T)) * 0.9 + 0.05            # the semiring lives on [0, 1]
    words = ["[1]", "[12][3][2213]", "[1][-2]", "[3][1][2][1]", "[12][3]"]
    for weighting in (None, {"kind": "Indices", "scale": 3.0},
                      {"kind": "L2", "scale": 1.0, "total": True}):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     semiring=fr.semiring.Bayesian(), weighting=make_weighting(fr, weighting))
        lookup, total = orc._weight_lookup(weighting, X, X)
        ref = corc.iss_transform(X, words, "EXTENDED", None, lookup, total, semiring="Bayesian")
        out = iss.fit_transform(X)
        if weighting is None:
            np.testing.assert_array_equal(out, ref)    # products in order, max is exact
        else:
            np.testing.assert_allclose(out, ref, rtol=1e-9)
    # the operator entry
    B = fr.semiring.Bayesian()
    word = fr.words.SimpleWord("[12][3][2213]").table()
    alpha = np.array([.6, .2, .3], dtype=np.float32)
    lk = orc.lookup_l2(X, False, 1.0)
    for total in (False, True):
        np.testing.assert_allclose(
            B.iterated_sum_fast(X, word, alpha, lk, 2, total),
            orc.bayesian_iterated_sum_fast(X, word, alpha, lk, 2, total), rtol=1e-9)


def test_bayesian_fused_pipeline(fr):
    rng = np.random.default_rng(8)
    X = rng.random((16, 2, 300)) * 0.9 + 0.05
    spec = {"slices": [{"iss": [{"words": G.manifest["words"]["3,2"]["words"], "mode": "EXTENDED",
                                 "semiring": "Bayesian"}],
                        "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "MPI", "inc": 0},
                                   {"kind": "END"}],
                        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(4)
    fruit.fit(X)
    assert fruit.get_slice()._fused(300) is not None
    got = fruit.transform(X)
    ref, expo = oracle_features(spec, X, X, np_seed=4)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    compare_features(got, ref, labels, expo)


@pytest.mark.parametrize("packed", ["0", "1"])
@pytest.mark.parametrize("T", [1, 2, 7, 64, 100, 128, 129, 255, 256, 257, 300, 383, 384, 385])
def test_short_series_kernels(fr, monkeypatch, packed, T):
    """T <= 256 runs on the wave-per-series kernel (four series per workgroup) by default;
    FRUITS_HIP_DEBUG=packed=0 keeps the cooperative kernel.  Both against the C oracle, with a
    series count that leaves waves without work, deep words (8 register levels), a
    per-series (L1) and a broadcast (Indices) lookup."""
    monkeypatch.setenv("FRUITS_HIP_DEBUG", f"packed={packed}")
    rng = np.random.default_rng(T)
    X = rng.standard_normal((13, 2, T)) / 2
    words = G.manifest["words"]["4,2"]["words"][:40] + ["[1][2][1][2][1][2][1][2]", "[2][-1]"]
    if T > 256 and T % 2:
        words = words[:40]      # <= 4 register levels: the 384-element wave-per-series variant
    for weighting in (None, {"kind": "Indices", "scale": 2.0},
                      {"kind": "L1", "scale": 3.0, "total": True}):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     weighting=make_weighting(fr, weighting))
        lookup, total = orc._weight_lookup(weighting, X, X)
        ref = corc.iss_transform(X, words, "EXTENDED", None, lookup, total)
        rowwise_close(iss.fit_transform(X), ref)
    # the other semirings share the kernel: Arctic bit-exact, Bayesian bit-exact unweighted
    Xp = np.abs(X) + 0.1
    for name in ("Arctic", "Bayesian"):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     semiring=getattr(fr.semiring, name)())
        np.testing.assert_array_equal(iss.fit_transform(Xp),
                                      corc.iss_transform(Xp, words, "EXTENDED", semiring=name))


def _random_word(rng, D):
    L = int(rng.integers(1, 6))
    letters = []
    for _ in range(L):
        el = ""
        for _ in range(int(rng.integers(1, 4))):
            d = int(rng.integers(1, D + 1))
            el += (f"(-{d})" if d > 9 else f"-{d}") if rng.random() < 0.2 else (f"({d})" if d > 9 else str(d))
        letters.append("[" + el + "]")
    return "".join(letters)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "60"))))
def test_random_differential(fr, seed):
    """Random word lists (duplicates, negative exponents, shared prefixes), shapes, modes,
    semirings and weightings against the C oracle."""
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 24, 40]))   # multiples of 8: XCD mapping
    T = int(rng.choice([1, 2, 3, 17, 64, 129, 256, 300, 385, 513, 700, 1025, 1500]))
    if os.environ.get("FRUITS_TEST_RANDOM_BIG"):          # soak runs: long series, many of them
        T = int(rng.choice([2049, 3000, 4097, 5000, 1024, 384]))
        N = int(rng.choice([33, 100, 257, 300]))
    words = [_random_word(rng, D) for _ in range(int(rng.integers(1, 13)))]
    if rng.random() < 0.5:
        words += [words[0], words[-1]]                       # duplicates
    mode = "EXTENDED" if rng.random() < 0.6 else "SINGLE"
    semiring = str(rng.choice(["Reals", "Reals", "Arctic", "Bayesian"]))
    weighting = [None, {"kind": "Indices", "scale": 2.0},
                 {"kind": "Indices", "scale": 2.0, "total": True},
                 {"kind": "L1", "scale": 3.0}, {"kind": "L1", "scale": 3.0, "total": True}][
        int(rng.integers(0, 5))]
    X = rng.random((N, D, T)) * 0.9 + 0.3                     # away from 0: 1/x letters
    ws = [fr.words.SimpleWord(s) for s in words]
    alphas = None
    if weighting is not None and rng.random() < 0.5:
        alphas = [rng.random(len(w)).round(2).tolist() for w in ws]
        for w, a in zip(ws, alphas):
            w.alpha = a
    iss = fr.ISS(ws, mode=getattr(fr.ISSMode, mode), semiring=getattr(fr.semiring, semiring)(),
                 weighting=make_weighting(fr, weighting))
    out = iss.fit_transform(X)
    lookup, total = orc._weight_lookup(weighting, X, X)
    ref = corc.iss_transform(X, words, mode, alphas, lookup, total, semiring=semiring)
    assert out.shape == ref.shape
    if semiring == "Reals":
        rowwise_close(out, ref)
    elif weighting is None:
        np.testing.assert_array_equal(out, ref)               # max is exact
    else:
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "40"))))
def test_random_fruit_differential(fr, seed, monkeypatch):
    """Random single-slice fruits (preparateurs, word lists, weightings, sieves with cuts,
    bands and differencing orders) through Fruit.fit / transform - the fused launch where
    the sieves allow it - against the numpy oracle."""
    rng = np.random.default_rng(5000 + seed)
    if seed % 3 == 2:      # a third of the cases through the materialising sieve kernels
        monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    D = int(rng.integers(1, 4))
    N = int(rng.integers(6, 20))
    T = int(rng.choice([24, 65, 128, 200, 300, 400, 513, 600, 1030]))
    Dp = 2 * D if rng.random() < 0.3 else D
    preps = [[{"kind": "INC"}], [], [{"kind": "STD"}],
             [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}]][
        3 if Dp != D else int(rng.integers(0, 3))]
    words = sorted({_random_word(rng, Dp).replace("-", "") for _ in range(int(rng.integers(1, 10)))})
    semiring = str(rng.choice(["Reals", "Reals", "Arctic"]))
    weighting = [None, None, {"kind": "Indices", "scale": 2.0}, {"kind": "L1", "scale": 2.0},
                 {"kind": "Indices", "scale": 2.0, "total": True},
                 {"kind": "L1", "scale": 2.0, "total": True}][int(rng.integers(0, 6))]
    sieves = []
    for _ in range(int(rng.integers(1, 5))):
        kind = str(rng.choice(["NPI", "MPI", "END"]))
        cut = sorted({int(c) for c in rng.integers(1, T, size=int(rng.integers(0, 3)))}) + [-1]
        if rng.random() < 0.35:    # float cuts: per-series coquantile positions
            cut = [float(c) for c in rng.choice([0.2, 0.35, 0.5, 0.8], size=int(rng.integers(1, 3)),
                                                replace=False)] + cut[:int(rng.integers(0, 2))]
        if kind == "END":
            sieves.append({"kind": "END", "cut": cut})
        else:
            q = [[0.0, 1.0], [-1.0, 1.0], [0.5, 1.0], [0.25, 0.5, 1.0], [-1.0, 0.3, 0.7, 1.0]][
                int(rng.integers(0, 5))]
            sieves.append({"kind": kind, "cut": cut, "q": q,
                           "inc": int(rng.integers(0, 6))})   # 3..5 at T = 1030: the HIGHORD instances
    spec = {"slices": [{"preps": preps,
                        "iss": [{"words": words, "mode": str(rng.choice(["EXTENDED", "SINGLE"])),
                                 "semiring": semiring, "weighting": weighting}],
                        "sieves": sieves, "fit_sample_size": 1.0}]}
    X = rng.standard_normal((N, D, T)).cumsum(axis=2) / np.sqrt(T)
    fruit = build_fruit(fr, spec)
    np.random.seed(seed)
    fruit.fit(X)
    got = fruit.transform(X)
    ref, expo = oracle_features(spec, X, X, np_seed=seed)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    assert got.shape == ref.shape
    # fitted quantiles are data points of the (small) fit sample, a running maximum (Arctic)
    # has long plateaus, the cumulative sum of a standardised series ends at 0 +- rounding:
    # all of these are exact threshold ties, identified element by element by the oracle's
    # exposure - everything else has to match exactly
    compare_features(got, ref, labels, expo)
    # ... and the two halves separately: the fitted thresholds, and the transform alone with
    # the oracle's thresholds under the tight exposure
    strict_transform_parity(fruit, spec, X, X, labels, np_seed=seed,
                            what=f"random fruit {seed} ({semiring})")


def test_plateaus_and_custom_weightings(fr):
    # host-built lookups (fruits/iss/weighting.py:41-66,213-256) consumed by the same kernel
    rng = np.random.default_rng(21)
    X = rng.random((5, 2, 90))
    words = ["[1][2]", "[12][2][1]", "[2]"]
    for weighting in (fr.iss.weighting.Plateaus(4, scale=3.0),
                      fr.iss.weighting.Plateaus(3, reverse=True, scale=2.0, total=True),
                      fr.iss.weighting.Custom(lambda Z: np.cumsum(np.abs(Z[:, 0, :]), axis=1) / 30.0),
                      fr.iss.weighting.Custom(lambda Z: Z[:, 1, :] * 2.0, total=True)):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     weighting=weighting)
        out = iss.fit_transform(X)
        lookup = np.asarray(weighting.get_lookup(X), dtype=np.float64)
        ref = corc.iss_transform(X, words, "EXTENDED", None, lookup, weighting.total)
        rowwise_close(out, ref)


def test_theoretical_identity(fr):
    # reference tests/signature/test_simple.py:44-51: standardised x => <[1][1]>_T = -T/2
    X = np.random.default_rng(5).random((25, 1, 100))
    X = (X - X.mean(axis=2, keepdims=True)) / X.std(axis=2, keepdims=True)
    res = fr.ISS([fr.words.SimpleWord("[1][1]")]).fit_transform(X)
    np.testing.assert_allclose(np.ones(25) * -50, res[0, :, -1], rtol=1e-9)


def test_negative_word_identity(fr):
    # reference tests/signature/test_simple.py:76-79
    X = G["X_1"]
    a = fr.ISS([fr.words.SimpleWord("[1][2]")]).fit_transform(1 / (X + 10))
    b = fr.ISS([fr.words.SimpleWord("[-1][-2]")]).fit_transform(X + 10)
    np.testing.assert_allclose(a, b, rtol=1e-12)


def test_extended_equals_stacked_single(fr):
    # reference tests/signature/test_cache.py:85-124
    X = np.random.default_rng(6).random((10, 3, 100))
    e = G.manifest["cacheplan"][0]
    ext = fr.ISS([fr.words.SimpleWord(s) for s in e["words"]],
                 mode=fr.ISSMode.EXTENDED).fit_transform(X)
    single = fr.ISS([fr.words.SimpleWord(s) for s in e["labels"]]).fit_transform(X)
    np.testing.assert_allclose(single, ext, rtol=1e-12)


def test_operator_entry(fr):
    # Semiring.iterated_sum_fast == fr_iterated_sum_fast_host
    Z = G["U_6_3_40"]
    word = fr.words.SimpleWord("[12][2][33]").table()
    alpha = np.array([.6, .2, .5], dtype=np.float32)
    lk = G["op/lookup"]
    R = fr.semiring.Reals()
    np.testing.assert_allclose(R.iterated_sum_fast(Z, word, alpha, lk, 2, False),
                               G["op/fast_nontotal_E2"], rtol=RTOL)
    np.testing.assert_allclose(R.iterated_sum_fast(Z, word, alpha, lk, 3, True),
                               G["op/fast_total_E3"], rtol=RTOL)
    un = R.iterated_sums(Z, fr.words.SimpleWord("[12][2][33]"), 3)
    np.testing.assert_allclose(
        un, orc.iterated_sums(Z, orc.parse_word("[12][2][33]"), extended=3), rtol=RTOL)
    with pytest.raises(IndexError):
        R.iterated_sum_fast(Z[:, :2], word, alpha, lk, 2, False)
    with pytest.raises(TypeError):
        R.iterated_sum_fast(Z.astype(np.float32), word, alpha, lk, 2, False)


@pytest.mark.parametrize("T", [1, 2, 3, 5, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025,
                               2047, 2048, 2500, 4096, 5000])
def test_ragged_lengths(fr, T):
    rng = np.random.default_rng(T)
    X = rng.random((3, 2, T))
    words = ["[1]", "[12]", "[1][2]", "[1][2][11]", "[2][2]", "[2][1][1]"]
    for weighting in (None, {"kind": "Indices", "scale": 3.0},
                      {"kind": "Indices", "scale": 3.0, "total": True}):
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                     weighting=make_weighting(fr, weighting))
        lookup, total = orc._weight_lookup(weighting, X, X)
        ref = corc.iss_transform(X, words, "EXTENDED", None, lookup, total)
        np.testing.assert_allclose(iss.fit_transform(X), ref, rtol=RTOL)


def test_empty_batch(fr):
    out = fr.ISS([fr.words.SimpleWord("[1][1]")]).fit_transform(np.zeros((0, 1, 16)))
    assert out.shape == (1, 0, 16)


@pytest.mark.parametrize("groups", [1, 2, 3, 9])
def test_groups_agree(fr, groups):
    from fruits_amd import _native as nat
    X = gen_input({"seed": 3, "dist": "normal", "shape": [16, 3, 700]})
    ws = fr.words.of_weight(2, 3)
    iss = fr.ISS(ws, mode=fr.ISSMode.EXTENDED)
    Xd = nat.to_device(X)
    a = nat.to_host(iss.transform_device(Xd, groups=1))
    b = nat.to_host(iss.transform_device(Xd, groups=groups))
    np.testing.assert_array_equal(a, b)


def test_shared_vs_unshared_plan(fr):
    from fruits_amd import _native as nat
    X = gen_input({"seed": 4, "dist": "uniform", "shape": [5, 2, 300]})
    ws = fr.words.of_weight(4, 2)
    cp = fr.iss.CachePlan(ws)._plan
    Xd = nat.to_device(X)
    a = nat.Plan([w.table() for w in ws], cp, share_prefixes=True).run(Xd)
    b = nat.Plan([w.table() for w in ws], cp, share_prefixes=False).run(Xd)
    np.testing.assert_allclose(nat.to_host(a), nat.to_host(b), rtol=1e-12)


def test_config2_full_size(fr):
    """BASELINE configs[1]: of_weight(2, 3) EXTENDED on (2048, 3, 1024), against the
    C oracle on the whole tensor (norm-wise on N(0,1), element-wise on U[0,1))."""
    W = G.manifest["words"]["2,3"]["words"]
    for spec, elementwise in (({"seed": 0, "dist": "normal", "shape": [2048, 3, 1024]}, False),
                              ({"seed": 1, "dist": "uniform", "shape": [2048, 3, 1024]}, True)):
        X = gen_input(spec)
        iss = fr.ISS([fr.words.SimpleWord(s) for s in W], mode=fr.ISSMode.EXTENDED)
        out = iss.fit_transform(X)
        ref = corc.iss_transform(X, W, "EXTENDED")
        rowwise_close(out, ref)
        if elementwise:
            np.testing.assert_allclose(out, ref, rtol=RTOL)
        # size-independent properties: [a] rows are plain cumsums, linear in x^2
        np.testing.assert_allclose(out[0], np.cumsum(X[:, 0] ** 2, axis=1), rtol=1e-9)
    # the metric's "48 words": words[i % 15], SINGLE -> duplicated rows are identical
    X = gen_input({"seed": 0, "dist": "normal", "shape": [256, 3, 1024]})
    W48 = [W[i % 15] for i in range(48)]
    out = fr.ISS([fr.words.SimpleWord(s) for s in W48]).fit_transform(X)
    assert out.shape == (48, 256, 1024)
    for i in range(15, 48):
        np.testing.assert_array_equal(out[i], out[i % 15])
    rowwise_close(out[:15], corc.iss_transform(X, W[:15], "SINGLE"))


@pytest.mark.parametrize("case", G.cases("l1"), ids=lambda c: c["name"])
def test_l1_lookup(fr, case):
    kw = dict(case["kw"])
    wt = fr.iss.weighting.L1(**kw)
    X = G[case["x"]]
    wt._cache = fr.cache.SharedSeedCache(X)
    np.testing.assert_array_equal(wt.get_lookup(X), G[case["out"]])     # sequential sum: exact


@pytest.mark.parametrize("case", G.manifest.get("lookups", []), ids=lambda c: c["name"])
def test_other_lookups(fr, case):
    wt = getattr(fr.iss.weighting, case["kind"])(**case["kw"])
    X = G[case["x"]]
    wt._cache = fr.cache.SharedSeedCache(X)
    np.testing.assert_array_equal(wt.get_lookup(X), G[case["out"]])
    from fruits_amd import _native as nat
    dev = nat.to_host(wt.lookup_device(nat.to_device(X)))
    np.testing.assert_array_equal(np.broadcast_to(dev, G[case["out"]].shape), G[case["out"]])


@pytest.mark.parametrize("case", G.cases("inc"), ids=lambda c: c["name"])
def test_inc(fr, case):
    inc = fr.preparation.INC(**case["kw"])
    X = G[case["x"]]
    out = inc.fit_transform(X)
    np.testing.assert_array_equal(out, G[case["out"]])   # one subtraction: bit-exact
    np.testing.assert_array_equal(inc.copy().fit_transform(X), out)


@pytest.mark.parametrize("case", G.cases("sieve"), ids=lambda c: c["name"])
def test_sieves(fr, case):
    kw = dict(case["kw"])
    if "q" in kw:
        kw["q"] = tuple(kw["q"])
    sv = getattr(fr.sieving, case["kind"])(**kw)
    A = G[case["x"]]
    sv.fit(G[case["fit"]] if case["fit"] else A)
    out = sv.transform(A)
    assert [sv.label(i) for i in range(sv.nfeatures())] == case["labels"]
    if case["kind"] == "MPI":
        np.testing.assert_allclose(out, G[case["out"]], rtol=1e-9, atol=1e-12)
    else:
        np.testing.assert_array_equal(out, G[case["out"]])   # counts / gathers: exact
    # (a copy forgets coquantile_norm, in the reference too: segment.py:90-91, increment.py:83-84)
    if "coquantile_norm" not in kw:
        np.testing.assert_array_equal(sv.copy().fit_transform(A) if not case["fit"] else out, out)


@pytest.mark.parametrize("case", G.manifest.get("coswiss", []), ids=lambda c: c["name"])
def test_coswiss_golden(fr, case):
    # reference tests/signature/test_cosine.py checks the same quantity against a
    # brute-force definition at rtol 1e-5; here: the reference's output itself
    X = G[case["x"]]
    cw = fr.CosWISS([fr.words.SimpleWord(s) for s in case["words"]], case["freqs"], **case["kw"])
    assert cw.n_iterated_sums() == len(case["labels"])
    assert [cw.label(i) for i in range(cw.n_iterated_sums())] == case["labels"]
    for s, w in case["weightings"].items():
        assert cw._get_weightings(fr.words.SimpleWord(s)).tolist() == w
    ref = G[case["out"]]
    out = cw.fit_transform(X)
    assert out.shape == ref.shape
    scale = np.abs(ref).max(axis=2, keepdims=True)
    assert np.all(np.abs(out - ref) <= RTOL * np.maximum(np.abs(ref), 1e-3 * scale))
    # batch_transform yields batch_size words x all frequencies at a time
    F = len(case["freqs"])
    parts = list(cw.batch_transform(X, batch_size=1))
    assert len(parts) == len(case["words"]) and parts[0].shape[0] == F
    np.testing.assert_array_equal(np.concatenate(parts), out)


def test_coswiss_brute_force(fr):
    # the definition the reference tests against (tests/signature/test_cosine.py:8-40):
    # sum_{j<k<=t} x_j y_k cos(pi (k-j) / (f (T-1)))^s, in plain loops
    rng = np.random.default_rng(5)
    X = rng.random((3, 2, 24))
    T, f = 24, 0.7
    fq = float(np.float32(f))
    for s in (1, 2):
        out = fr.CosWISS([fr.words.SimpleWord("[1][2]")], [f], exponent=s).fit_transform(X)
        g = np.pi * np.arange(T) / (fq * (T - 1))
        ref = np.zeros((3, T))
        for k in range(T):
            for j in range(k):
                ref[:, k:] += (X[:, 0, j] * X[:, 1, k] * np.cos(g[k] - g[j]) ** s)[:, None]
        np.testing.assert_allclose(out[0], ref, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("words,exponent,total", [
    (["[1][1][1][1]"], 3, False),            # reference test_cosine.py:86-110
    (["[1][1][1]"], 4, False),               # reference test_cosine.py:113-134
    (["[1][1][1]", "[11][-1]"], 4, True),
    (["[1]", "[2][1]", "[12][2][33]", "[3][11]", "[11][23][1]"], 2, True),
    (["[1][2]", "[2][1][1]", "[3]"], 5, True),     # exponents 5-8: factorised kernels too
    (["[1][2][3]", "[11][2]"], 6, False),
    (["[2][1]", "[1][1][2]"], 7, True),
    (["[1][2]", "[3][1][2]"], 8, False),
])
def test_coswiss_vs_oracle(fr, words, exponent, total):
    X = np.random.default_rng(len(words) + exponent).random((10, 3, 50)) + 0.25
    freqs = [0.05, 0.5, 0.45]                 # fruit_reduced uses i/20
    cw = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=exponent,
                    total_weighting=total)
    out = cw.fit_transform(X)
    ref = orc.coswiss_transform(X, words, freqs, exponent, total)
    assert out.shape == ref.shape
    # the terms cancel (alternating trig products): tolerance relative to the row scale
    scale = np.abs(ref).max(axis=2, keepdims=True)
    assert np.all(np.abs(out - ref) <= RTOL * np.maximum(np.abs(ref), 1e-3 * scale))


@pytest.mark.parametrize("T", [130, 300, 450, 700, 1024, 1030, 2051])
@pytest.mark.parametrize("total", [False, True])
def test_coswiss_long_series(fr, T, total):
    # chunk sizes 512 / 1024, several chunks (carries), odd lengths (scalar accesses)
    rng = np.random.default_rng(T)
    X = rng.standard_normal((6, 2, T)) / np.sqrt(T)
    words = ["[1]", "[2][1]", "[1][2][2]", "[12][1][-2][1]"] if T < 2000 else ["[1]", "[2][1]"]
    X[:, 1] = np.abs(X[:, 1]) + 0.5
    freqs = [0.15, 0.5]
    for exponent in (1, 2):
        cw = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=exponent,
                        total_weighting=total)
        out = cw.fit_transform(X)
        ref = orc.coswiss_transform(X, words, freqs, exponent, total)
        scale = np.abs(ref).max(axis=2, keepdims=True)
        assert np.all(np.abs(out - ref) <= RTOL * np.maximum(np.abs(ref), 1e-3 * scale))


def test_coswiss_short_series_cooperative(fr, monkeypatch):
    # T <= 384 runs one wave per (series, word, frequency) unit; FRUITS_HIP_DEBUG=packed=0 keeps the
    # cooperative kernel - both against the oracle
    X = np.random.default_rng(9).random((7, 2, 100)) + 0.25
    words, freqs = ["[1]", "[2][1]", "[1][2][2]"], [0.15, 0.5]
    ref = orc.coswiss_transform(X, words, freqs, 2, True)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    for packed in ("1", "0"):
        monkeypatch.setenv("FRUITS_HIP_DEBUG", f"packed={packed}")
        out = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=2,
                         total_weighting=True).fit_transform(X)
        assert np.all(np.abs(out - ref) <= RTOL * np.maximum(np.abs(ref), 1e-3 * scale))


def test_coswiss_term_path(fr, monkeypatch):
    # exponents beyond the factorised kernels use the reference's term-by-term form
    X = np.random.default_rng(3).random((4, 2, 40)) + 0.25
    words, freqs = ["[1][2]", "[2][1][1]"], [0.3, 0.5]
    ref5 = orc.coswiss_transform(X, words, freqs, 9, True)
    cw5 = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=9,
                     total_weighting=True)
    assert not cw5._native()
    out5 = cw5.fit_transform(X)
    scale = np.abs(ref5).max(axis=2, keepdims=True)
    assert np.all(np.abs(out5 - ref5) <= RTOL * np.maximum(np.abs(ref5), 1e-3 * scale))
    # and both forms agree where both exist
    cw2 = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=2)
    a = cw2.fit_transform(X)
    monkeypatch.setenv("FRUITS_AMD_COSWISS_TERMS", "1")
    b = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=2).fit_transform(X)
    np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("T", [96, 1024, 1100])
@pytest.mark.parametrize("total", [False, True])
def test_coswiss_fused_pipeline(fr, monkeypatch, T, total):
    # NEW(INC) -> STD -> CosWISS -> NPI / MPI (inc 0, 1, 2) + END in one launch
    # (experiments/fruit_reduced.py:52-69) vs the materialised path vs the oracle
    rng = np.random.default_rng(T + total)
    X = rng.standard_normal((24, 1, T)).cumsum(axis=2)
    spec = {"slices": [{
        "preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
        "iss": [{"kind": "CosWISS", "words": ["[1]", "[2]", "[1][2]", "[2][1][1]"],
                 "freqs": [0.05, 0.25], "exponent": 2, "total_weighting": total,
                 "mode": "SINGLE"}],
        "sieves": [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0},
                   {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
                   {"kind": "NPI", "q": [0.5, 1.0], "inc": 2},
                   {"kind": "MPI", "q": [0.5, 1.0], "inc": 0},
                   {"kind": "MPI", "q": [0.5, 1.0], "inc": 1},
                   {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
                   {"kind": "END"}],
        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    fruit.fit(X)
    assert fruit.get_slice()._fused(T) is not None
    fused = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    fruit.get_slice()._fused_cache = {}
    plain = fruit.transform(X)
    ref, expo = oracle_features(spec, X, X)
    compare_features(fused, ref, labels, expo, what=f"coswiss fused T={T} total={total}")
    compare_features(plain, ref, labels, expo, what=f"coswiss materialised T={T} total={total}")


def test_nan_to_num_epilogue(fr):
    # Fruit.transform ends with np.nan_to_num(result, nan=0.0) (fruits/fruit.py:172)
    from fruits_amd import _native as nat
    x = np.array([[0.0, np.nan, np.inf], [-np.inf, 1.5, -0.0]])
    got = nat.to_host(nat.nan_to_num(nat.to_device(x)))
    np.testing.assert_array_equal(got, np.nan_to_num(x, nan=0.0))
    # through a fruit: 1/x on a series with zeros gives inf / nan iterated sums
    X = np.zeros((3, 1, 16))
    X[1] = 1.0
    fruit = fr.Fruit()
    fruit.add(fr.ISS([fr.words.SimpleWord("[-1]"), fr.words.SimpleWord("[1][-1]")]),
              fr.sieving.END)
    feats = fruit.fit_transform(X)
    with np.errstate(all="ignore"):
        its = orc.iss_transform(X, ["[-1]", "[1][-1]"], "SINGLE")
    want = np.nan_to_num(np.stack([its[0][:, -1], its[1][:, -1]], axis=1), nan=0.0)
    np.testing.assert_array_equal(feats, want)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "30"))))
def test_random_coswiss_differential(fr, seed):
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.integers(1, 4))
    N = int(rng.integers(1, 7))
    T = int(rng.choice([2, 5, 33, 128, 129, 300, 385, 600, 1100]))
    words = [_random_word(rng, D) for _ in range(int(rng.integers(1, 6)))]
    words = [w for w in words if len(orc.parse_word(w)) <= 4] or ["[1]"]
    freqs = [float(f) for f in rng.choice([0.05, 0.15, 0.25, 0.5, 0.75, 1.0, 2.0],
                                          size=int(rng.integers(1, 4)), replace=False)]
    exponent = int(rng.integers(1, 9))
    total = bool(rng.random() < 0.5)
    X = rng.random((N, D, T)) * 0.9 + 0.3
    out = fr.CosWISS([fr.words.SimpleWord(s) for s in words], freqs, exponent=exponent,
                     total_weighting=total).fit_transform(X)
    ref = orc.coswiss_transform(X, words, freqs, exponent, total)
    assert out.shape == ref.shape
    scale = np.abs(ref).max(axis=2, keepdims=True)
    # (a row whose terms cancel to ~1e-229 of the others - high exponents on a few steps - is
    # judged on the scale of the whole result, not its own)
    floor = 1e-9 * np.abs(ref).max()
    assert np.all(np.abs(out - ref) <= RTOL * np.maximum(np.maximum(np.abs(ref), 1e-3 * scale), floor))


def test_entry_points_capture_into_a_hip_graph(fr):
    """fr_iss_run and fr_pipeline_run only enqueue work: captured into a
    HIP graph and replayed they reproduce the eager results - cooperative, wave-per-series
    and CosWISS kernels, materialising and fused."""
    import torch
    from fruits_amd import _native as nat
    rng = np.random.default_rng(12)
    for T in (100, 1024):
        X = rng.standard_normal((32, 2, T)).cumsum(axis=2) / np.sqrt(T)
        Xd = nat.to_device(X)
        iss = fr.ISS(fr.words.of_weight(3, 2), mode=fr.ISSMode.EXTENDED,
                     weighting=fr.iss.weighting.Indices())
        plan = iss._plan(0, len(iss.words))
        lk = iss.lookup_device(Xd)
        out = torch.empty((plan.rows, 32, T), dtype=torch.float64, device=Xd.device)
        work = torch.empty(max(plan.workspace_bytes(32, T, 1), 1), dtype=torch.uint8, device=Xd.device)
        eager = plan.run(Xd, lk, work=work).clone()
        cw = fr.CosWISS(fr.words.of_weight(2, 2), [0.25, 0.5], exponent=2)
        cplan = cw._plan(0, len(cw.words))
        cout = torch.empty((cplan.rows, 32, T), dtype=torch.float64, device=Xd.device)
        cwork = torch.empty(max(cplan.workspace_bytes(32, T, 0), 1), dtype=torch.uint8, device=Xd.device)
        ceager = cplan.run(Xd, None, work=cwork).clone()
        fruit = fr.Fruit()
        fruit.add(iss.copy(), fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.MPI(q=(0.5, 1.0)), fr.sieving.END)
        fruit.fit(X)
        slc = fruit.get_slice()
        pipe = slc._fused(T)
        assert pipe is not None
        feats = torch.empty((32, pipe.n_features), dtype=torch.float64, device=Xd.device)
        pwork = torch.empty(int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, 32, 1)) + 1,
                            dtype=torch.uint8, device=Xd.device)
        slc._attach(fr.cache.SharedSeedCache(X))
        feager = pipe.run(Xd, lk, work=pwork).clone()
        torch.cuda.synchronize()
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                plan.run(Xd, lk, out=out, work=work)
                cplan.run(Xd, None, out=cout, work=cwork)
                pipe.run(Xd, lk, feats=feats, work=pwork)
        out.zero_(); cout.zero_(); feats.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager) and torch.equal(cout, ceager)
        np.testing.assert_allclose(feats.cpu().numpy(), feager.cpu().numpy(), rtol=1e-12)


def test_capture_without_prior_eager_run(fr):
    """fr_plan_prepare / fr_pipeline_prepare do the one-time upload: a plan that has never
    run eagerly is captured into a graph; an UNPREPARED plan refuses to run inside a
    capture (it would allocate and synchronise) instead of breaking it."""
    import torch
    from fruits_amd import _native as nat
    from oracle import ref_numpy as orc_np
    rng = np.random.default_rng(21)
    N, T = 24, 1024
    X = rng.standard_normal((N, 2, T)).cumsum(axis=2) / np.sqrt(T)
    Xd = nat.to_device(X)
    words = fr.words.of_weight(3, 2)
    # materialising plan, never run before
    iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
    plan = iss._plan(0, len(words))
    out = torch.zeros((plan.rows, N, T), dtype=torch.float64, device=Xd.device)
    # fused pipeline, never run before
    fruit = fr.Fruit()
    fruit.add(fr.ISS(words, mode=fr.ISSMode.EXTENDED), fr.sieving.NPI, fr.sieving.END)
    fruit.fit(X)                      # unfitted sieves: no device work on this plan
    slc = fruit.get_slice()
    pipe = slc._fused(T)
    feats = torch.zeros((N, pipe.n_features), dtype=torch.float64, device=Xd.device)
    pwork = torch.empty(int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, N, 0)) + 1,
                        dtype=torch.uint8, device=Xd.device)
    # an unprepared plan inside a capture: refused, the capture survives
    # (a plan of the record interpreter: the pre-compiled static programs of the standard
    # word sets read no device tables and have nothing to prepare)
    other = fr.ISS(fr.words.of_weight(3, 2))._plan(0, 6)
    oout = torch.zeros((other.rows, N, T), dtype=torch.float64, device=Xd.device)
    plan.prepare(N, T)
    pipe.prepare(N)
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            plan.run(Xd, None, out=out)
            pipe.run(Xd, None, feats=feats, work=pwork)
            with pytest.raises(ValueError, match="fr_plan_prepare"):
                other.run(Xd, None, out=oout)
    g.replay()
    torch.cuda.synchronize()
    ref = orc_np.iss_transform(X, [str(w) for w in words], "EXTENDED")
    rowwise_close(out.cpu().numpy(), ref)
    np.testing.assert_allclose(feats.cpu().numpy(), fruit.transform(X), rtol=1e-12)
    # outside a capture the unprepared plan uploads by itself
    other.run(Xd, None, out=oout)
    assert plan.fits(T) and plan.fits(100)


def test_capture_of_a_plan_in_pieces(fr, monkeypatch):
    """The enqueue-only contract holds for a plan in pieces too: after fr_pipeline_prepare (tables,
    kernels of every piece type) a run - one launch per piece type, the band means' finalize, the
    gather of the row blocks - is captured into a graph and replays to the eager result."""
    import torch
    from fruits_amd import _native as nat
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "0")
    _debug_knobs(monkeypatch, pieces=1, piece_min=50, piece_nodes=16)
    rng = np.random.default_rng(33)
    N, T = 24, 1024
    X = rng.standard_normal((N, 2, T)).cumsum(axis=2) / 5.0
    fruit = fr.Fruit("pieces in a graph")
    fruit.add(fr.ISS(fr.words.of_weight(4, 2), mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices()))
    fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.MPI(q=(0.5, 1.0)), fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(1)
    fruit.fit(X)
    slc = fruit.get_slice()
    pipe = slc._fused(T)
    pipe.prepare(N)
    if pipe.pieces_loaded() == 0:
        pytest.skip("hipRTC is not installed")
    Xd = nat.to_device(X)
    lk = slc.get_iss()[0].lookup_device(Xd)
    feats = torch.zeros((N, pipe.n_features), dtype=torch.float64, device=Xd.device)
    work = torch.empty(int(nat.lib().fr_pipeline_workspace_bytes(pipe._h, N, int(lk.shape[0]))) + 1,
                       dtype=torch.uint8, device=Xd.device)
    eager = pipe.run(Xd, lk, work=work).clone()
    torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            pipe.run(Xd, lk, feats=feats, work=work)
    feats.zero_()
    g.replay()
    torch.cuda.synchronize()
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = torch.tensor(["MPI" not in lb for lb in labels], device=feats.device)
    assert bool((feats[:, exact] == eager[:, exact]).all())
    torch.testing.assert_close(feats, eager, rtol=1e-12, atol=0.0)


def test_end_cut_out_of_range_raises(fr):
    """END(cut=c) with c - 1 outside [-T, T-1]: the reference raises IndexError
    (np.take_along_axis); fused and unfused paths agree on that."""
    rng = np.random.default_rng(2)
    X = rng.standard_normal((4, 1, 50))
    for cut in (51, 1000, [-130, -120]):
        with pytest.raises(IndexError):
            fr.sieving.END(cut=cut).fit_transform(X[:, 0, :])
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord("[1]")]), fr.sieving.END(cut=cut))
        fruit.fit(X)
        with pytest.raises(IndexError):
            fruit.transform(X)
    # the extremes the reference accepts
    ref = X[:, 0, :].cumsum(axis=1)
    fruit = fr.Fruit()
    fruit.add(fr.ISS([fr.words.SimpleWord("[1]")]), fr.sieving.END(cut=[50, 1, -50, -1]))
    fruit.fit(X)
    np.testing.assert_allclose(fruit.transform(X), ref[:, [0, 0, 49, 49]], rtol=1e-12)


@pytest.mark.parametrize("case", G.manifest.get("coswiss_random", []), ids=lambda c: c["name"])
def test_coswiss_random_variants_golden(fr, case):
    """The ffn / dropout variants (fruits/iss/cos.py:51-164) against the reference's outputs:
    a seeded fit draws the same weights / indices as the reference's (same generator calls in
    the same order), the transform agrees to 1e-6 (observed ~1e-15; the goldens come from
    numpy's pairwise np.sum, the kernel follows numba's sequential one)."""
    kw = case["kw"]
    X = G[case["x"]]
    cw = fr.CosWISS([fr.words.SimpleWord(s) for s in case["words"]], case["freqs"], **kw)
    assert cw.requires_fitting
    with pytest.raises(RuntimeError):
        cw.transform(X)
    np.random.seed(case["np_seed"])
    cw.fit(X)
    if "A" in case:
        for name, mine in (("A", cw._A), ("b", cw._b), ("C", cw._C)):
            np.testing.assert_array_equal(mine, G[case[name]])
    if "dropout_indices" in case:
        np.testing.assert_array_equal(cw._dropout_indices, G[case["dropout_indices"]])
    out = cw.transform(X)
    ref = G[case["out"]]
    assert out.shape == ref.shape
    rowwise_close(out, ref)
    np.testing.assert_allclose(out, ref, rtol=RTOL, atol=1e-12)
    blocks = list(cw.batch_transform(X, batch_size=1))
    np.testing.assert_array_equal(np.concatenate(blocks, axis=0), out)


@pytest.mark.parametrize("T", [96, 1024, 1100])
def test_coswiss_dropout_fused_and_long(fr, monkeypatch, T):
    """Dropout through the fused pipeline (packed, cooperative and multi-chunk kernels)
    against the materialising path and the oracle."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((10, 2, T)).cumsum(axis=2) / 4.0
    words = ["[1]", "[1][2]", "[2][1][1]"]
    cw = fr.CosWISS([fr.words.SimpleWord(s) for s in words], [0.25, 0.5], exponent=2,
                    total_weighting=True, dropout=0.15)
    fruit = fr.Fruit()
    fruit.add(cw, fr.sieving.NPI, fr.sieving.END)
    np.random.seed(3)
    fruit.fit(X)
    fitted = fruit.get_slice().get_iss()[0]
    assert fruit.get_slice()._fused(T) is not None
    fused = fruit.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    fruit.get_slice()._fused_cache = {}
    plain = fruit.transform(X)
    ref = orc.coswiss_transform(X, words, [0.25, 0.5], 2, True,
                                dropout_indices=fitted._dropout_indices)
    np.testing.assert_allclose(fused[:, 1::2], ref[:, :, -1].T, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(plain[:, 1::2], ref[:, :, -1].T, rtol=RTOL, atol=1e-9)
    rowwise_close(fitted.transform(X), ref)
    # indices drawn on the fit's length apply to any series they fit into
    # (tmp[dropout[k]] = 0, fruits/iss/cos.py:84); one beyond the input's end is an IndexError
    longer = np.concatenate([X, X[:, :, :7]], axis=2)
    rowwise_close(fitted.transform(longer),
                  orc.coswiss_transform(longer, words, [0.25, 0.5], 2, True,
                                        dropout_indices=fitted._dropout_indices))
    assert fitted._dropout_indices.max() >= T // 2
    with pytest.raises(IndexError):
        fitted.transform(X[:, :, :T // 2].copy())


def test_coswiss_unsupported(fr):
    cw = fr.CosWISS([fr.words.SimpleWord("[1]")], [0.5], dropout=0.5, exponent=9)
    with pytest.raises(NotImplementedError):
        cw.transform_device(None)
    with pytest.raises(ValueError):
        fr.CosWISS([object()], [0.5])


def build_fruit(fr, spec):
    fruit = fr.Fruit(spec.get("name", ""))
    for sl in spec["slices"]:
        fruit.cut()
        for p in sl.get("preps", []):
            kw = {k: v for k, v in p.items() if k not in ("kind", "inner")}
            if p["kind"] == "NEW":
                inner = p.get("inner")
                obj = None if inner is None else getattr(fr.preparation, inner["kind"])(
                    **{k: v for k, v in inner.items() if k != "kind"})
                fruit.add(fr.preparation.NEW(obj))
            else:
                fruit.add(getattr(fr.preparation, p["kind"])(**kw))
        for i in sl["iss"]:
            ws = [fr.words.SimpleWord(s) for s in i["words"]]
            if i.get("kind") == "CosWISS":
                fruit.add(fr.CosWISS(freqs=i["freqs"], words=ws, exponent=i.get("exponent", 2),
                                     total_weighting=i.get("total_weighting", False)))
                continue
            semiring = getattr(fr.semiring, i.get("semiring", "Reals"))
            fruit.add(fr.ISS(ws, mode=getattr(fr.ISSMode, i["mode"]),
                             semiring=semiring(argmax=True) if i.get("argmax") else semiring(),
                             weighting=make_weighting(fr, i.get("weighting"))))
        for s in sl["sieves"]:
            kw = {k: (tuple(v) if k == "q" else v) for k, v in s.items() if k != "kind"}
            fruit.add(getattr(fr.sieving, s["kind"])(**kw))
        if "fit_sample_size" in sl:
            fruit.get_slice().fit_sample_size = sl["fit_sample_size"]
    return fruit


def compare_features(got, ref, labels, expo=None, rtol=RTOL, what="", count_frac=0.01,
                     mean_rel=0.25, min_off=0, count_max=1):
    """Features against the oracle's.

    ``expo`` (oracle ``fruit_exposure``: per entry the number of elements within 1e-10 of
    a band threshold - exact ties with a fitted quantile that is a data point, plateaus of
    a running maximum, absorbed increments next to 0): the criterion is then BY
    CONSTRUCTION - a counting feature (NPI) must equal the reference exactly wherever no
    element is exposed and may differ by at most the number of exposed elements elsewhere;
    band means (MPI) must agree to rtol where nothing is exposed (one element entering or
    leaving a band moves its mean arbitrarily); value features (END) always to rtol.
    The observed numbers go to the parity report printed at the end of the run.

    Without ``expo`` (comparisons of two GPU paths, no oracle at hand): counts may differ
    by ``count_max`` on at most ``count_frac`` of the entries."""
    assert got.shape == ref.shape
    is_count = np.array(["NPI" in lb for lb in labels])
    is_mean = np.array(["MPI" in lb for lb in labels])
    val = ~is_count & ~is_mean
    if val.any():
        # atol: END of e.g. <[1]> on standardised data is an exact-zero sum, i.e.
        # pure rounding noise (1e-15) in the reference and here
        np.testing.assert_allclose(got[:, val], ref[:, val], rtol=rtol, atol=1e-9)
    if expo is not None:
        assert expo.shape == got.shape
        from conftest import PARITY_REPORT
        d = np.abs(got[:, is_count] - ref[:, is_count])
        e = expo[:, is_count]
        rec = {"what": what or os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1],
               "entries": int(d.size), "exposed": int((e > 0).sum()), "differ": int((d > 0).sum()),
               "differ_unexposed": int(((d > 0) & (e == 0)).sum()),
               "max_d": float(d.max()) if d.size else 0.0}
        PARITY_REPORT.append(rec)
        print(f"[parity] {rec}")
        assert rec["differ_unexposed"] == 0, rec
        assert np.all(d <= e), (rec, float((d - e).max()))
        if is_mean.any():
            g, r, em = got[:, is_mean], ref[:, is_mean], expo[:, is_mean]
            tight = em == 0
            off = np.abs(g - r) > rtol * np.abs(r) + 1e-9
            assert not (off & tight).any(), (int((off & tight).sum()), what)
        return rec
    if is_mean.any():
        # a band mean inherits the count's sensitivity: when one on-threshold element
        # enters or leaves the band the mean moves by ~1/population
        g, r = got[:, is_mean], ref[:, is_mean]
        off = np.abs(g - r) > rtol * np.abs(r) + 1e-9
        assert off.sum() <= max(count_frac * off.size, min_off)
        if mean_rel is not None:
            assert np.all(np.abs(g - r)[off] <= mean_rel * np.abs(r)[off] + 1e-9)
    if is_count.any():
        d = np.abs(got[:, is_count] - ref[:, is_count])
        if count_max is not None:
            assert d.max() <= count_max
        assert (d > 0).sum() <= max(count_frac * d.size, min_off)
    return None


def transplant_thresholds(fruit, fitted):
    """The oracle's fitted thresholds into the sieve copies of a fitted GPU fruit: the transform
    comparison then tests the transform alone (same thresholds on both sides) and the fit is
    compared on its own (fit_parity)."""
    for slc, rows in zip(fruit, orc.fitted_thresholds(fitted)):
        if not rows:
            continue        # no sieve of the slice is fitted
        assert len(rows) == len(slc._sieves_extended)
        for mine, theirs in zip(slc._sieves_extended, rows):
            for sv, q in zip(mine, theirs):
                if q is not None and sv.requires_fitting:
                    sv._quantiles = q.copy()
        slc._fused_cache = {}


def fit_parity(fruit, fitted, what=""):
    """Thresholds fitted on the GPU (order statistics selected on the device from the GPU's own
    iterated sums) against the oracle's.  An order statistic is a value of the (differenced) row
    (or the interpolation of two): it can differ from the oracle's by what the values differ by -
    the re-association of the scan, and a difference of two sums inherits the rounding of the
    sums however small it is itself.  So the deviation is measured against the magnitude of the
    rows the sieve was fitted on: at most 1e-13 of it (observed: up to 1e-15)."""
    from conftest import FIT_REPORT
    n = same = 0
    worst = 0.0
    for slc, rows, scales in zip(fruit, orc.fitted_thresholds(fitted), orc.fitted_scales(fitted)):
        if not rows:
            continue
        for mine, theirs, sc_row in zip(slc._sieves_extended, rows, scales):
            for sv, q, sc in zip(mine, theirs, sc_row):
                if q is None or not sv.requires_fitting:
                    continue
                g = np.asarray(sv._quantiles, dtype=np.float64)
                assert g.shape == q.shape
                fin = np.isfinite(q)
                assert np.array_equal(np.isfinite(g), fin) and np.array_equal(g[~fin], q[~fin])
                if not fin.any():
                    continue
                scale = max(np.abs(q[fin]).max(), sc, 1e-300)
                dev = np.abs(g[fin] - q[fin]).max() / scale
                n += int(fin.sum())
                same += int((g[fin] == q[fin]).sum())
                worst = max(worst, float(dev))
                assert dev <= 1e-13, (what, g, q, sc)
    FIT_REPORT.append({"what": what, "thresholds": n, "identical": same, "max_rel": worst})


def _max_plus_columns(spec, fruit):
    """Per feature column: does it belong to a slice over the Arctic / Bayesian semiring?  Such
    rows are running maxima - long plateaus of bit-equal values, so that a threshold which IS
    such a value (or the threshold 0 under a differencing sieve) ties with whole runs of
    elements at once.  Columns of sums (Reals, CosWISS) tie one data point at a time."""
    mask = []
    for sl, slc in zip(spec["slices"], fruit):
        mp = any(i.get("semiring", "Reals") != "Reals" for i in sl["iss"])
        mask += [mp] * slc.nfeatures()
    return np.array(mask, dtype=bool)


def compare_strict(got, ref, labels, expo, means, rtol=RTOL, what="", max_plus=None):
    """Features of the GPU transform run with the ORACLE's thresholds against the oracle's, with
    the tight exposure (SieveOracle.exposure, tight=True): counts equal wherever nothing is
    exposed, within the number of exposed elements elsewhere; band means to rtol where nothing
    is exposed, and ONE OF the candidate means (exposed elements moved across the threshold)
    where up to four elements are; values (END) to rtol.  ``max_plus``: per column, see
    _max_plus_columns - the two classes are recorded (and barred) separately."""
    from conftest import STRICT_REPORT
    assert got.shape == ref.shape == expo.shape
    labels = np.asarray(labels)
    if max_plus is None:
        max_plus = np.zeros(got.shape[1], dtype=bool)
    is_count = np.array(["NPI" in lb for lb in labels])
    is_mean = np.array(["MPI" in lb for lb in labels])
    val = ~is_count & ~is_mean
    if val.any():
        np.testing.assert_allclose(got[:, val], ref[:, val], rtol=rtol, atol=1e-9)
    recs = []
    for cls, cols in (("sum", is_count & ~max_plus), ("max-plus", is_count & max_plus)):
        if not cols.any():
            continue
        d = np.abs(got[:, cols] - ref[:, cols])
        e = expo[:, cols]
        rec = {"what": what or os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1],
               "entries": int(d.size), "exposed": int((e > 0).sum()), "differ": int((d > 0).sum()),
               "differ_unexposed": int(((d > 0) & (e == 0)).sum()),
               "max_d": float(d.max()) if d.size else 0.0, "means_checked": 0,
               "exposed_elements": int(e.sum()), "series": int(got.shape[0]), "cls": cls}
        STRICT_REPORT.append(rec)
        recs.append(rec)
        print(f"[strict] {rec}")
        assert rec["differ_unexposed"] == 0, rec
        assert np.all(d <= e), (rec, float((d - e).max()))
        if cls == "sum":
            # sums: a tie is ONE data point sitting on a threshold that was fitted from it
            assert rec["max_d"] <= 1, rec
        else:
            # max-plus rows are formed by max / + / rounded products only, in the reference's
            # order, from inputs that are bit-equal to the reference's (INC, and since round 4
            # STD in numpy's summation order, csrc/pairwise.h): plateau ties fall the same
            # way on both sides - the counts are EQUAL
            assert rec["differ"] == 0, rec
    if is_mean.any():
        cols = np.nonzero(is_mean)[0]
        g, r, em = got[:, is_mean], ref[:, is_mean], expo[:, is_mean]
        off = np.abs(g - r) > rtol * np.abs(r) + 1e-9
        assert not (off & (em == 0)).any(), (int((off & (em == 0)).sum()), what)
        checked = 0
        for n, jj in zip(*np.nonzero(em > 0)):
            cands = means.get((int(n), int(cols[jj])))
            if cands is None:
                continue          # more than four exposed elements: a plateau
            c = np.asarray(cands)
            assert np.any(np.abs(g[n, jj] - c) <= rtol * np.abs(c) + 1e-9), (what, n, cols[jj], g[n, jj], cands)
            checked += 1
        if recs:
            recs[0]["means_checked"] = checked
    return recs


def strict_transform_parity(fruit, spec, X_fit, X, labels, np_seed=None, what=""):
    """The two halves of the parity claim, separately: (1) the thresholds the GPU fitted against
    the oracle's; (2) the GPU transform WITH the oracle's thresholds against the oracle's
    features under the tight exposure.  Leaves the fruit with the oracle's thresholds."""
    if np_seed is not None:
        np.random.seed(np_seed)
    fitted = orc.fruit_fit(spec, X_fit)
    fit_parity(fruit, fitted, what)
    transplant_thresholds(fruit, fitted)
    means = {}
    ref, expo = orc.fruit_transform_exposure(spec, fitted, X, rel=1e-13, tight=True, means=means)
    return compare_strict(fruit.transform(X), ref, labels, expo, means, what=what,
                          max_plus=_max_plus_columns(spec, fruit))


def oracle_features(spec, X_fit, X, np_seed=None):
    """(reference features, exposure) of the numpy oracle fitted on X_fit."""
    if np_seed is not None:
        np.random.seed(np_seed)
    fitted = orc.fruit_fit(spec, X_fit)
    return orc.fruit_transform_exposure(spec, fitted, X)


@pytest.mark.parametrize("case", G.cases("fruit"), ids=lambda c: c["name"])
def test_fruit_golden(fr, case):
    X = G[case["x"]]
    fruit = build_fruit(fr, case["spec"])
    assert fruit.nfeatures() == case["nfeatures"]
    assert [fruit.label(i) for i in range(fruit.nfeatures())] == case["labels"]
    assert [fruit.label(i, verbose=2) for i in range(len(case["labels_v2"]))] == case["labels_v2"]
    assert fruit.summary() == case["summary"]
    if case["np_seed"] is not None:
        np.random.seed(case["np_seed"])
    fruit.fit(X)
    out = fruit.transform(X)
    # the oracle's fit equals the reference's (tests/test_oracle.py pins its features to
    # these goldens); it supplies the tie exposure of every entry
    ref, expo = oracle_features(case["spec"], X, X, np_seed=case["np_seed"])
    np.testing.assert_allclose(ref, G[case["out"]], rtol=1e-9, atol=1e-12)
    compare_features(out, G[case["out"]], case["labels"], expo, what="golden " + case["name"])
    strict_transform_parity(fruit, case["spec"], X, X, case["labels"], np_seed=case["np_seed"],
                            what="golden " + case["name"])
    if "x_test" in case:
        Xt = G[case["x_test"]]
        np.random.seed(case["np_seed"]) if case["np_seed"] is not None else None
        fitted = orc.fruit_fit(case["spec"], X)
        compare_features(fruit.transform(Xt), G[case["out_test"]], case["labels"],
                         orc.fruit_exposure(case["spec"], fitted, Xt),
                         what="golden " + case["name"] + " (test batch)")
    with pytest.raises(RuntimeError):
        build_fruit(fr, case["spec"]).transform(X)


def test_c_abi_raw_device_entry(fr):
    """Calls fr_iss_run / fr_increments with raw device pointers from fr_malloc
    (no torch tensors): what a cgo / plain-ctypes maintainer would bind."""
    from fruits_amd import _native as nat
    L = nat.lib()
    X = G["U_6_3_40"]
    N, D, T = X.shape
    plan = nat.Plan([fr.words.SimpleWord(s).table() for s in ["[11]", "[1][2]"]], [1, 2])
    dX, dO = C.c_void_p(), C.c_void_p()
    assert L.fr_malloc(C.byref(dX), C.c_int64(X.nbytes)) == 0
    assert L.fr_malloc(C.byref(dO), C.c_int64(3 * N * T * 8)) == 0
    out = np.zeros((3, N, T))
    assert L.fr_memcpy_h2d(dX, X.ctypes.data_as(C.c_void_p), C.c_int64(X.nbytes), None) == 0
    rc = L.fr_iss_run(plan._h, dX, C.c_int64(N), C.c_int64(D), C.c_int64(T), None,
                      C.c_int64(0), dO, C.c_int64(N * T), C.c_int64(T), None, C.c_int64(0),
                      C.c_int32(0), None)
    assert rc == 0, nat.last_error()
    assert L.fr_memcpy_d2h(out.ctypes.data_as(C.c_void_p), dO, C.c_int64(out.nbytes), None) == 0
    assert L.fr_stream_sync(None) == 0
    ref = orc.iss_transform(X, ["[11]", "[1][2]"], "EXTENDED")
    np.testing.assert_allclose(out, ref, rtol=RTOL)
    L.fr_free(dX)
    L.fr_free(dO)


@pytest.mark.parametrize("name", ["cfg3_small", "reduced_coswiss_small", "fruit_reduced_verbatim"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_word_sharded_blocks_reassemble(fr, world, name):
    """The word-sharded pipeline (fruits_amd.parallel) run rank after rank on one GPU with
    a loop-back gather: every slice (Reals, Arctic, CosWISS) equals the unsharded
    transform - each rank's share is ONE fused launch."""
    from fruits_amd import parallel as par
    from fruits_amd.cache import SharedSeedCache
    case = [c for c in G.cases("fruit") if c["name"] == name][0]
    X = G[case["x"]]
    fruit = build_fruit(fr, case["spec"])
    np.random.seed(case["np_seed"])
    fruit.fit(X)
    ref = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    is_mean = np.array(["MPI" in lb for lb in labels])
    col0 = 0
    for slc in fruit:
        iss = slc.get_iss()[0]
        strings = [str(w) for w in iss.words]
        depths = [iss._depth(i) for i in range(len(strings))]
        per_sum = sum(s.nfeatures() for s in slc.get_sieves())
        parts = par.shard_words(strings, depths, world)
        maps = par.column_map(parts, depths, per_sum)
        out = np.zeros((X.shape[0], slc.nfeatures()))
        from fruits_amd import _native as nat
        for r in range(world):
            cache = SharedSeedCache(X)
            if parts[r]:
                assert slc._fused(X.shape[2], indices=parts[r]) is not None
            block = par._device_block(slc, iss, cache.input_device(X), cache, parts[r], depths,
                                      per_sum).cpu().numpy()
            assert block.shape[1] == len(maps[r])
            out[:, maps[r]] = block
        want = ref[:, col0:col0 + slc.nfeatures()]
        got = np.nan_to_num(out)
        m = is_mean[col0:col0 + slc.nfeatures()]
        np.testing.assert_array_equal(got[:, ~m], want[:, ~m])
        # MPI band sums are accumulated with float atomics (order varies run to run)
        np.testing.assert_allclose(got[:, m], want[:, m], rtol=1e-12, atol=1e-300)
        col0 += slc.nfeatures()
    if world == 1:
        full = par.transform_sharded(fruit, X, rank=0, world=1)
        np.testing.assert_allclose(full, ref, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("world", [2, 3])
def test_word_sharded_argmax(fr, world):
    """A rank's share of the words of an Arctic(argmax=True) slice is one argmax pipeline
    (fr_pipeline_set_argmax over the share's words); the shares reassemble to the unsharded
    transform."""
    if os.environ.get("FRUITS_AMD_FUSED_ARGMAX", "1") == "0" or os.environ.get("FRUITS_AMD_FUSED", "1") == "0":
        pytest.skip("the knob sweep switched the argmax pipeline off")
    from fruits_amd import parallel as par
    from fruits_amd.cache import SharedSeedCache
    rng = np.random.default_rng(world)
    X = rng.standard_normal((11, 2, 257)).cumsum(axis=2)
    fruit = fr.Fruit()
    fruit.add(fr.ISS(fr.words.of_weight(3, 2), mode=fr.ISSMode.EXTENDED,
                     semiring=fr.semiring.Arctic(argmax=True)))
    fruit.add(fr.sieving.NPI(q=(0.4, 1.0), inc=1), fr.sieving.END, fr.sieving.MPI(q=(0.5, 1.0)))
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(0)
    fruit.fit(X)
    ref = fruit.transform(X)
    slc = fruit.get_slice()
    iss = slc.get_iss()[0]
    strings = [str(w) for w in iss.words]
    depths = [iss._depth(i) for i in range(len(strings))]
    per_sum = sum(s.nfeatures() for s in slc.get_sieves())
    parts = par.shard_words(strings, depths, world)
    maps = par.column_map(parts, depths, per_sum)
    out = np.zeros_like(ref)
    for r in range(world):
        cache = SharedSeedCache(X)
        pipe = slc._fused(X.shape[2], indices=parts[r])
        assert pipe is not None and pipe.rows == sum(depths[i] for i in parts[r])
        out[:, maps[r]] = par._device_block(slc, iss, cache.input_device(X), cache, parts[r], depths,
                                            per_sum).cpu().numpy()
    np.testing.assert_allclose(np.nan_to_num(out), ref, rtol=1e-12, atol=1e-300)


def test_argmax_pipeline_limits(fr):
    """What fr_pipeline_set_argmax refuses keeps the materialising path (same features): series
    whose maxima and positions exceed a workgroup's LDS, a sieve that differences three times."""
    if os.environ.get("FRUITS_AMD_FUSED_ARGMAX", "1") == "0" or os.environ.get("FRUITS_AMD_FUSED", "1") == "0":
        pytest.skip("the knob sweep switched the argmax pipeline off")
    rng = np.random.default_rng(5)
    for T, inc in ((7000, 1), (300, 3)):
        X = rng.standard_normal((3, 1, T)).cumsum(axis=2)
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord("[1][1][1][1]"), fr.words.SimpleWord("[1][-1]")],
                         mode=fr.ISSMode.EXTENDED, semiring=fr.semiring.Arctic(argmax=True)))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=inc), fr.sieving.END)
        fruit.get_slice().fit_sample_size = 1.0
        np.random.seed(0)
        feats = fruit.fit_transform(X)
        assert fruit.get_slice()._fused(T) is None
        rows = orc.iss_transform(X, ["[1][1][1][1]", "[1][-1]"], "EXTENDED", semiring="Arctic", argmax=True)
        np.testing.assert_array_equal(feats[:, 1::2], rows[:, :, -1].T)
    # (at T = 4096 a word of three letters still fits: 32 KB + 3 x 8 KB)
    X = rng.standard_normal((3, 1, 4096)).cumsum(axis=2)
    fruit = fr.Fruit()
    fruit.add(fr.ISS([fr.words.SimpleWord("[1][1][1]")], mode=fr.ISSMode.EXTENDED,
                     semiring=fr.semiring.Arctic(argmax=True)))
    fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(0)
    feats = fruit.fit_transform(X)
    assert fruit.get_slice()._fused(4096) is not None
    rows = orc.iss_transform(X, ["[1][1][1]"], "EXTENDED", semiring="Arctic", argmax=True)
    np.testing.assert_array_equal(feats[:, 1::2], rows[:, :, -1].T)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_word_sharded_coswiss_ffn(fr, world):
    """A CosWISS with the randomised ffn (every (word, frequency) reads its own transformed copy
    of the input, fruits/iss/cos.py:93-137) under word sharding: a rank's words go through the
    word-by-word fused launches, never through a pipeline on the plain batch - the blocks
    re-assemble to the unsharded transform; also with a rank that holds ONE word."""
    from fruits_amd import parallel as par
    from fruits_amd.cache import SharedSeedCache
    T = 700
    rng = np.random.default_rng(21)
    X = rng.standard_normal((19, 2, T)).cumsum(axis=2) / 6.0
    fruit = fr.Fruit("ffn sharded")
    fruit.add(fr.preparation.INC)
    words = [fr.words.SimpleWord(s) for s in ["[1]", "[1][2]", "[2][1][1]", "[2]"]]
    fruit.add(fr.CosWISS(freqs=[0.1, 0.35], words=words, exponent=2, ffn_size=5))
    fruit.add(fr.sieving.NPI(q=(0.4, 1.0)), fr.sieving.NPI(inc=2), fr.sieving.END(cut=[T // 2, -1]))
    slc = fruit.get_slice()
    slc.fit_sample_size = 1.0
    np.random.seed(6)
    fruit.fit(X)
    ref = fruit.transform(X)
    iss = slc.get_iss()[0]
    strings = [str(w) for w in iss.words]
    depths = [iss._depth(i) for i in range(len(strings))]
    per_sum = sum(s.nfeatures() for s in slc.get_sieves())
    parts = par.shard_words(strings, depths, world)
    maps = par.column_map(parts, depths, per_sum)
    out = np.zeros_like(ref)
    for r in range(world):
        cache = SharedSeedCache(X)
        block = par._device_block(slc, iss, cache.input_device(X), cache, parts[r], depths, per_sum)
        assert block.shape[1] == len(maps[r])
        out[:, maps[r]] = block.cpu().numpy()
    np.testing.assert_array_equal(np.nan_to_num(out), ref)
    # the unsharded transform in between must not disturb a word's plan (its input stride)
    np.testing.assert_array_equal(fruit.transform(X), ref)


@pytest.mark.parametrize("name", ["readme", "cfg3_small", "cfg3_small_unweighted", "twi_small",
                                  "twi_small_hot", "x1_two_slices_end", "reduced_slice1_small",
                                  "reduced_arctic_small", "twi_arctic_small"])
def test_fused_matches_materialised(fr, name, monkeypatch):
    """The fused ISS+sieve launch against the materialising path (fr_iss_run +
    fr_sieve per iterated sum): END and inc=0 features are bit-identical, inc=1
    counts may differ by one where a lane boundary rounds differently."""
    case = [c for c in G.cases("fruit") if c["name"] == name][0]
    X = G[case["x"]]
    fruit = build_fruit(fr, case["spec"])
    np.random.seed(case["np_seed"])
    fruit.fit(X)
    assert any(s._fused(X.shape[2]) is not None for s in fruit._slices)
    fused = fruit.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    for s in fruit._slices:
        s._fused_cache = {}
    plain = fruit.transform(X)
    labels = case["labels"]
    end = np.array(["END" in lb for lb in labels])
    np.testing.assert_array_equal(fused[:, end], plain[:, end])
    ref, expo = oracle_features(case["spec"], X, X, np_seed=case["np_seed"])
    compare_features(fused, ref, labels, expo, what="fused " + name)
    compare_features(plain, ref, labels, expo, what="materialised " + name)


@pytest.mark.parametrize("T", [7, 64, 511, 1024, 1500, 3000])
def test_fused_ragged_and_multichunk(fr, T):
    rng = np.random.default_rng(T)
    X = rng.standard_normal((5, 2, T))
    spec = {"slices": [{"preps": [{"kind": "INC"}],
                        "iss": [{"words": G.manifest["words"]["3,2"]["words"], "mode": "EXTENDED"}],
                        "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "NPI", "inc": 0},
                                   {"kind": "MPI", "cut": [T // 2, -1]}, {"kind": "END", "cut": [1, -1]},
                                   {"kind": "NPI", "q": [0.5, 1.0], "inc": 2},
                                   {"kind": "MPI", "inc": 2, "cut": [max(T // 3, 1), -1]}],
                        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(3)
    fruit.fit(X)
    assert fruit._slices[0]._fused(T) is not None
    got = fruit.transform(X)
    ref, expo = oracle_features(spec, X, X, np_seed=3)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    compare_features(got, ref, labels, expo, what=f"ragged / multi-chunk T={T}")


@pytest.mark.parametrize("T", [200, 300, 511, 1024, 1500])
@pytest.mark.parametrize("chain", ["INC", "INC3", "NEW_INC", "STD", "INC_STD", "NEW_INC_STD0"])
def test_fused_preparation(fr, monkeypatch, chain, T):
    """INC / NEW(INC) / STD formed while the fused launch stages the RAW rows
    (fr_pipeline_set_preparation) against the same pipeline on a materialised prepared
    input - bit for bit (same arithmetic, same summation order of the statistics) - and
    against the oracle; Reals + Indices and Arctic slices; the wave-per-series kernels of short
    series (T = 200, 300: round 4), one chunk and several."""
    preps = {"INC": [{"kind": "INC"}], "INC3": [{"kind": "INC", "shift": 3}],
             "NEW_INC": [{"kind": "NEW", "inner": {"kind": "INC"}}], "STD": [{"kind": "STD"}],
             "INC_STD": [{"kind": "INC"}, {"kind": "STD"}],
             "NEW_INC_STD0": [{"kind": "NEW", "inner": {"kind": "INC"}},
                              {"kind": "STD", "var": False}]}[chain]
    D = 2
    Dp = 2 * D if chain.startswith("NEW") else D
    rng = np.random.default_rng(T + len(chain))
    X = rng.standard_normal((12, D, T)).cumsum(axis=2) / 4.0
    words = ["[1]", "[1][%d]" % Dp, "[%d][1%d]" % (Dp, Dp), "[%d]" % Dp]
    spec = {"slices": [
        {"preps": preps, "iss": [{"words": words, "mode": "EXTENDED",
                                  "weighting": {"kind": "Indices", "scale": 3.0}}],
         "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "MPI", "inc": 0}, {"kind": "END"}],
         "fit_sample_size": 1.0},
        {"preps": preps, "iss": [{"words": ["[1][%d][1]" % Dp], "mode": "EXTENDED",
                                  "semiring": "Arctic"}],
         "sieves": [{"kind": "NPI", "q": [0.3, 1.0], "inc": 2}, {"kind": "END", "cut": [T // 2, -1]}],
         "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(1)
    fruit.fit(X)
    got = fruit.transform(X)
    for slc in fruit:
        pipe = slc._fused(T)
        assert pipe is not None and pipe.raw_dims == D      # the raw input went in
    monkeypatch.setenv("FRUITS_AMD_FUSED_PREP", "0")
    plain = fruit.transform(X)
    for slc in fruit:
        assert slc._fused(T).raw_dims == 0
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = np.array(["MPI" not in lb for lb in labels])   # (band sums: float atomics)
    np.testing.assert_array_equal(got[:, exact], plain[:, exact])
    np.testing.assert_allclose(got, plain, rtol=1e-12, atol=1e-300)
    ref, expo = oracle_features(spec, X, X, np_seed=1)
    compare_features(got, ref, labels, expo, what=f"fused preparation {chain} T={T}")
    monkeypatch.delenv("FRUITS_AMD_FUSED_PREP")
    strict_transform_parity(fruit, spec, X, X, labels, np_seed=1, what=f"fused preparation {chain} T={T}")


@pytest.mark.parametrize("T", [300, 700, 1500])
@pytest.mark.parametrize("chain", ["INC", "NEW_INC_STD", "STD0"])
def test_fused_preparation_coswiss(fr, monkeypatch, chain, T):
    """The same for a CosWISS slice (round 4): the kernel forms the prepared rows where it reads a
    letter's rows from the RAW input (coswiss.h, load_prepared_row) - wave-per-unit kernels
    (T = 300), the cooperative one, several time chunks; dropout rides along; equal to the
    pipeline on a materialised prepared input bit for bit, and to the oracle."""
    preps = {"INC": [{"kind": "INC"}],
             "NEW_INC_STD": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
             "STD0": [{"kind": "STD", "var": False}]}[chain]
    D = 2
    Dp = 2 * D if chain.startswith("NEW") else D
    rng = np.random.default_rng(T + len(chain))
    X = rng.standard_normal((14, D, T)).cumsum(axis=2) / 4.0
    words = ["[1]", "[1][%d]" % Dp, "[%d][1%d]" % (Dp, Dp)]
    spec = {"slices": [
        {"preps": preps, "iss": [{"kind": "CosWISS", "words": words, "freqs": [0.125, 0.5], "exponent": 2,
                                  "total_weighting": chain != "INC"}],
         "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "NPI", "inc": 2}, {"kind": "END"}],
         "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(1)
    fruit.fit(X)
    default = fruit.transform(X)
    assert fruit.get_slice()._fused(T).raw_dims == 0        # (not by default: slower, see _fusable_preparation)
    monkeypatch.setenv("FRUITS_AMD_FUSED_PREP", "2")
    got = fruit.transform(X)
    pipe = fruit.get_slice()._fused(T)
    assert pipe is not None and pipe.raw_dims == D          # the raw input went in
    monkeypatch.setenv("FRUITS_AMD_FUSED_PREP", "0")
    plain = fruit.transform(X)
    assert fruit.get_slice()._fused(T).raw_dims == 0
    np.testing.assert_array_equal(got, plain)
    np.testing.assert_array_equal(default, plain)
    ref, expo = oracle_features(spec, X, X, np_seed=1)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    compare_features(got, ref, labels, expo, what=f"fused preparation CosWISS {chain} T={T}")


@pytest.mark.parametrize("T", [1, 5, 7, 8, 9, 127, 128, 129, 1000, 1024, 1025, 4096, 8192, 8193, 20000])
def test_std_bit_identical_to_numpy(fr, T):
    """STD is np.mean / np.std along the contiguous axis (fruits/preparation/transform.py:141-147):
    numpy's pairwise summation in buffers of 8192 elements is a deterministic order and the device
    kernel follows it (csrc/pairwise.h) - the standardised rows are EQUAL, not close; rows of
    very different magnitude, constant rows, var=False, INC in front (the fused statistics
    pre-pass sees x[t] - x[t-1] formed on the fly)."""
    rng = np.random.default_rng(100 + T)
    N = 60 if T <= 4096 else 6
    X = rng.standard_normal((N, 2, T)) * np.exp(rng.uniform(-8, 8, size=(N, 2, 1)))
    X[::7] = X[::7].cumsum(axis=2)
    if N > 3:
        X[3, 0] = 2.5                       # a constant row: std 0, eps decides
        X[2, 1] = 0.0
    for var in (True, False):
        got = fr.preparation.STD(var=var).transform(X)
        np.testing.assert_array_equal(got, orc.std_transform(X, var=var))
    if T > 1:
        inc = orc.inc_transform(X)
        got = fr.preparation.STD().transform(fr.preparation.INC().transform(X))
        np.testing.assert_array_equal(got, orc.std_transform(inc))


@pytest.mark.parametrize("T", [65, 1024, 2100])
def test_std_fused_staging_is_exact(fr, T):
    """The same through the fused launch (statistics pre-pass + standardisation in the staging):
    an Arctic slice behind NEW(INC) -> STD ends in max / + only, so its END features must EQUAL
    the oracle's, and so must its counting features (plateau ties included)."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((16, 2, T)).cumsum(axis=2) * 3.0
    spec = {"slices": [
        {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
         "iss": [{"words": ["[1]", "[1][2]", "[3][1][4]", "[2][-1]"], "mode": "EXTENDED", "semiring": "Arctic"}],
         "sieves": [{"kind": "END", "cut": [T // 3, -1]}, {"kind": "NPI", "q": [0.0, 1.0], "inc": 2},
                    {"kind": "NPI", "q": [0.0, 1.0], "inc": 1}, {"kind": "NPI", "q": [0.0, 1.0], "inc": 3}]}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(0)
    fruit.fit(X)
    assert fruit.get_slice()._fused(T) is not None
    got = fruit.transform(X)
    np.random.seed(0)
    ref = orc.fruit_transform(spec, orc.fruit_fit(spec, X), X)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("T", [300, 700, 1300])
def test_coswiss_ffn_fused(fr, T, monkeypatch):
    """CosWISS with the randomised ffn: every (word, frequency) reads its own transformed input, so
    the slice fuses word by word (FruitSlice._transform_ffn_fused) - same features as the unfused
    path (materialised sums + sieve kernels) on the same fitted fruit."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((21, 2, T)).cumsum(axis=2) / 6.0
    fruit = fr.Fruit("ffn")
    fruit.add(fr.preparation.INC)
    words = [fr.words.SimpleWord(s) for s in ["[1]", "[1][2]", "[2][1][1]"]]
    fruit.add(fr.CosWISS(freqs=[0.1, 0.35], words=words, exponent=2, ffn_size=5))
    fruit.add(fr.sieving.NPI(q=(0.4, 1.0)), fr.sieving.MPI(q=(0.4, 1.0), inc=0), fr.sieving.NPI(inc=2),
              fr.sieving.END(cut=[T // 2, -1]))
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(6)
    fruit.fit(X)
    got = fruit.transform(X)
    pipe = fruit.get_slice()._fused(T, indices=(1,))
    assert pipe is not None                       # the word-by-word pipelines exist
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    counts = np.array(["NPI" in lb for lb in labels])
    ends = np.array(["END" in lb for lb in labels])
    np.testing.assert_allclose(got[:, ends], plain[:, ends], rtol=1e-12, atol=1e-300)
    # (counts: the fused epilogue forms an increment as the step of a sequential sum, the sieve
    # kernels as the difference of two stored values - an element ON a fitted threshold, itself a
    # data value, may fall on either side)
    d = np.abs(got[:, counts] - plain[:, counts])
    assert d.max() <= 2 and (d > 0).mean() <= 0.02, (d.max(), (d > 0).mean())
    means = ~counts & ~ends
    close = np.isclose(got[:, means], plain[:, means], rtol=1e-6, atol=1e-12)
    assert close.mean() >= 0.98


@pytest.mark.parametrize("T", [385, 700, 1500])
def test_letter_whose_exponents_cancel(fr, T):
    """[2-2] multiplies by nothing (its node has no factor): the iterated sums against the C
    oracle, and a fused fruit over the same words against the numpy oracle - the node loop of the
    fused walk reads its first factor from the record, so such a node takes the factor-table path."""
    rng = np.random.default_rng(T)
    X = rng.random((9, 3, T)) * 0.9 + 0.3
    words = ["[323][2-2][1]", "[2-2]", "[1][3-3][3-3][2]"]
    iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED)
    out = iss.fit_transform(X)
    ref = corc.iss_transform(X, words, "EXTENDED", None, None, False, semiring="Reals")
    rowwise_close(out, ref)
    spec = {"slices": [{"preps": [], "iss": [{"words": words, "mode": "EXTENDED"}],
                        "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "END"}],
                        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(2)
    fruit.fit(X)
    got = fruit.transform(X)
    assert fruit.get_slice()._fused(T) is not None
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    ref_f, expo = oracle_features(spec, X, X, np_seed=2)
    compare_features(got, ref_f, labels, expo, what=f"cancelling letter T={T}")


@pytest.mark.parametrize("prep", ["INC", "STD"])
def test_custom_weighting_sees_the_prepared_input(fr, monkeypatch, prep):
    """A Custom weighting (or any user subclass) builds its lookup from what the ISS is handed -
    the PREPARED input (fruits/iss/weighting.py:65-66): behind INC / STD such a slice must not
    take the fused preparation (which never writes the prepared tensor).  Fused launch vs the
    materialised preparation vs the unfused path, all the same features."""
    T = 600
    rng = np.random.default_rng(77)
    X = rng.standard_normal((9, 2, T)).cumsum(axis=2) / 5.0

    def build():
        fruit = fr.Fruit("custom")
        fruit.add(fr.preparation.INC if prep == "INC" else fr.preparation.STD)
        h = lambda Z: np.cumsum(np.abs(Z[:, 0, :]), axis=1) / (1.0 + np.abs(Z[:, 0, :]).sum(axis=1, keepdims=True))
        fruit.add(fr.ISS(fr.words.of_weight(2, dim=2), mode=fr.ISSMode.EXTENDED,
                         weighting=fr.iss.weighting.Custom(h)))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
        fruit.get_slice().fit_sample_size = 1.0
        np.random.seed(3)
        fruit.fit(X)
        return fruit
    fruit = build()
    got = fruit.transform(X)
    assert fruit.get_slice()._fusable_preparation(T) is None
    assert fruit.get_slice()._fused(T).raw_dims == 0
    monkeypatch.setenv("FRUITS_AMD_FUSED_PREP", "0")
    np.testing.assert_array_equal(got, build().transform(X))
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build().transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    compare_features(got, plain, labels, what=f"custom weighting behind {prep}")


@pytest.mark.parametrize("T", [60, 600, 1100])
@pytest.mark.parametrize("semiring", ["Reals", "Arctic"])
def test_chained_iss_fuses_its_last_stage(fr, monkeypatch, T, semiring):
    """A chain of ISS (fruits/fruit.py:440-454): the last ISS and the sieves run as one fused
    launch per row of the chain in front of it - same features as the unfused path (a sieve launch
    per iterated sum and sieve) and as the oracle; short, one-chunk and multi-chunk series."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((7, 2, T)).cumsum(axis=2) / 6.0
    spec = {"slices": [{"preps": [{"kind": "INC"}],
                        "iss": [{"words": ["[1]", "[12]", "[2][1]"], "mode": "EXTENDED", "semiring": semiring},
                                {"words": ["[1]", "[1][1]", "[11]"], "mode": "SINGLE", "semiring": semiring,
                                 "weighting": {"kind": "Indices", "scale": 2.0}}],
                        "sieves": [{"kind": "NPI", "q": [0.4, 1.0]}, {"kind": "MPI", "inc": 0, "cut": [T // 3, -1]},
                                   {"kind": "END"}],
                        "fit_sample_size": 1.0}]}
    fruit = build_fruit(fr, spec)
    np.random.seed(2)
    fruit.fit(X)
    slc = fruit.get_slice()
    assert slc._fused(T) is not None and slc._fused(T, chain_row=3) is not None
    assert slc._fused(T, chain_row=3) is not slc._fused(T)
    got = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    ref, expo = oracle_features(spec, X, X, np_seed=2)
    compare_features(got, ref, labels, expo, what=f"chained ISS {semiring} T={T}")
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    slc._fused_cache = {}
    plain = fruit.transform(X)
    compare_features(got, plain, labels, what=f"chained ISS {semiring} T={T} vs unfused")


@pytest.mark.parametrize("plan_too", [False, True], ids=["sieves", "sieves+plan"])
@pytest.mark.parametrize("which", ["counts", "bands_means", "arctic_total", "multi_chunk", "cuts",
                                   "large_plan", "large_plan_chunks"])
def test_fused_kernel_compiled_for_its_pipeline(fr, which, plan_too, monkeypatch):
    """fr_pipeline_prepare compiles the pipeline's own kernel (hipRTC): the fused walk with the
    sieves' kind / differencing order / shape / cuts as immediates - and fr_pipeline_compile_plan
    one that knows the plan: a small plan as straight-line code (fwalk_static), of a large one
    (more than 128 nodes) the node shapes (fwalk_shaped).  Same features as the generic kernel
    that decodes every record and op - bit for bit (band means: their wave sums are added in LDS
    in arrival order)."""
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "0")     # (the generic kernel first, then the own one)
    if not plan_too:
        monkeypatch.setenv("FRUITS_HIP_DEBUG", ",".join(
            v for v in (os.environ.get("FRUITS_HIP_DEBUG", ""), "fused_static=0") if v))
    T = {"multi_chunk": 1500, "large_plan_chunks": 2100}.get(which, 700)
    if which.startswith("large_plan"):
        _debug_knobs(monkeypatch, piece_nodes=40)          # (pieces that compile in seconds)
    rng = np.random.default_rng(len(which))
    X = rng.standard_normal((40, 2, T)).cumsum(axis=2) / 5.0
    fruit = fr.Fruit(which)
    fruit.add(fr.preparation.INC)
    if which.startswith("large_plan"):
        words = list(fr.words.of_weight(5, dim=2)) + [fr.words.SimpleWord("[1][-2][11][2]")]
        fruit.add(fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices()))
    elif which == "arctic_total":
        fruit.add(fr.ISS(fr.words.of_weight(3, dim=2), mode=fr.ISSMode.EXTENDED, semiring=fr.iss.semiring.Arctic(),
                         weighting=fr.iss.weighting.Indices(total=True)))
    else:
        fruit.add(fr.ISS(fr.words.of_weight(3, dim=2), mode=fr.ISSMode.EXTENDED,
                         weighting=fr.iss.weighting.Indices()))
    if which in ("counts", "large_plan_chunks"):
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.NPI, fr.sieving.END)
    elif which == "cuts":
        fruit.add(fr.sieving.NPI(cut=[T // 3, -1], q=(0.25, 0.5, 1.0)), fr.sieving.END(cut=[T // 2, -1]))
    else:
        fruit.add(fr.sieving.NPI(q=(0.3, 0.7, 1.0), inc=0), fr.sieving.MPI(q=(0.3, 0.7, 1.0), inc=0),
                  fr.sieving.NPI(inc=2), fr.sieving.MPI(inc=1), fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(4)
    fruit.fit(X)
    slc = fruit.get_slice()
    generic = fruit.transform(X)
    pipe = slc._fused(T)
    assert pipe.jit_loaded() == 0
    pipe.prepare(X.shape[0])
    if pipe.jit_loaded() == 0:
        pytest.skip("hipRTC is not installed")
    pieces_on = "pieces=0" not in os.environ.get("FRUITS_HIP_DEBUG", "")     # (tools/gpu_knobs.sh)
    if which.startswith("large_plan") and plan_too and pieces_on:
        assert pipe.pieces_loaded() >= 2      # (more than 128 nodes: in pieces, a kernel per piece type)
    else:
        assert pipe.jit_loaded(static_only=True) == (1 if plan_too else 0)
    own = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = np.array(["MPI" not in lb for lb in labels])
    np.testing.assert_array_equal(own[:, exact], generic[:, exact])
    np.testing.assert_allclose(own, generic, rtol=1e-12, atol=1e-300)
    fruit.fit(X)                                   # new thresholds: the compiled kernel is dropped
    assert slc._fused(T).jit_loaded() == 0


def _debug_knobs(monkeypatch, **knobs):
    keep = [v for v in os.environ.get("FRUITS_HIP_DEBUG", "").split(",")
            if v and v.split("=")[0] not in knobs]
    monkeypatch.setenv("FRUITS_HIP_DEBUG", ",".join(keep + [f"{k}={v}" for k, v in knobs.items()]))


@pytest.mark.parametrize("which", ["total_inc", "arctic", "chunks", "bands_means", "repeats",
                                   "high_orders", "bayesian_l1", "wide_root", "many_ops"])
def test_large_plan_in_pieces(fr, which, monkeypatch):
    """A plan of more than 128 nodes runs IN PIECES (csrc/plan.h PiecedProgram, walk_fused.h
    fwalk_pieces): chains walked by the record loop, bodies - whole sub-tries, equal ones one
    type - as straight-line code, one kernel per type, the features in walk order and gathered
    back.  Here with small pieces on plans of 60-120 nodes (the knobs piece_min / piece_nodes;
    seconds of compiler): the same features as the record loop, bit for bit (band means: wave
    sums meet in LDS in arrival order) - weightings, semirings, several time chunks (carries
    counted per unit), repeated words (more than two output rows per node), high differencing
    orders, a root with many small sub-tries (on a batch whose size is no multiple of 8: plain
    unit numbering), nine feature ops per output row on two time chunks (a unit's features exceed
    the LDS window: the kernels keep the flush test and every chunk adds onto the last one's)."""
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "0")
    T = {"chunks": 2100, "high_orders": 1500, "many_ops": 1100}.get(which, 600)
    rng = np.random.default_rng(len(which))
    D = 3 if which == "wide_root" else 2
    X = rng.standard_normal((21 if which == "wide_root" else 24, D, T)).cumsum(axis=2) / 5.0
    W = fr.iss.weighting
    mode = fr.ISSMode.EXTENDED
    if which == "repeats":
        words, mode = list(fr.words.of_weight(3, dim=2)) * 3 + list(fr.words.of_weight(4, dim=2)), fr.ISSMode.SINGLE
        kw = dict(weighting=W.Indices())
    elif which == "wide_root":
        words = list(fr.words.of_weight(2, dim=3)) + list(fr.words.of_weight(3, dim=3))
        kw = {}
    else:
        words = list(fr.words.of_weight(4, dim=2)) + [fr.words.SimpleWord("[1][-2][11][2]")]
        kw = {"indices": dict(weighting=W.Indices()),
              "total_inc": dict(weighting=W.Indices(total=True)),
              "arctic": dict(semiring=fr.iss.semiring.Arctic(), weighting=W.Indices(total=True)),
              "chunks": dict(weighting=W.Indices()),
              "many_ops": dict(weighting=W.Indices()),
              "bands_means": dict(weighting=W.Indices()),
              "high_orders": dict(weighting=W.Indices()),
              "bayesian_l1": dict(semiring=fr.iss.semiring.Bayesian(), weighting=W.L1())}[which]
        if which in ("arctic", "bayesian_l1"):
            words = words[:-1]
    fruit = fr.Fruit(which)
    fruit.add(fr.preparation.INC)
    fruit.add(fr.ISS(words, mode=mode, **kw))
    if which in ("bands_means", "arctic"):
        fruit.add(fr.sieving.NPI(q=(0.3, 0.7, 1.0), inc=0), fr.sieving.MPI(q=(0.3, 0.7, 1.0), inc=0),
                  fr.sieving.NPI(inc=2), fr.sieving.MPI(inc=1), fr.sieving.END)
    elif which == "high_orders":
        fruit.add(fr.sieving.NPI(inc=3), fr.sieving.NPI(q=(0.4, 1.0), inc=-1), fr.sieving.END(cut=[T // 2, -1]))
    elif which == "many_ops":
        fruit.add(fr.sieving.NPI(q=(0.2, 0.4, 0.6, 0.8, 1.0)), fr.sieving.NPI(q=(0.1, 0.5, 1.0), inc=0),
                  fr.sieving.END(cut=[T // 3, 2 * T // 3, -1]))
    else:
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.NPI, fr.sieving.END)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(4)
    fruit.fit(X)
    slc = fruit.get_slice()
    pipe = slc._fused(T)
    assert pipe is not None and pipe.plan.nodes >= 60
    _debug_knobs(monkeypatch, pieces=0)
    pipe.prepare(X.shape[0], plan_too=False)       # (the record loop with the sieves as immediates)
    if pipe.jit_loaded() == 0:
        pytest.skip("hipRTC is not installed")
    loop = fruit.transform(X)
    assert pipe.pieces_loaded() == 0
    # (units of several items behind one staging by default; of one item each on two of the cases)
    _debug_knobs(monkeypatch, pieces=1, piece_min=50, piece_nodes=12,
                 piece_unit=1 if which in ("chunks", "repeats") else 0)
    cover = pipe.plan.pieces(12)
    assert cover is not None and len(cover["types"]) >= 2
    pipe.prepare(X.shape[0])
    assert pipe.pieces_loaded() == len(cover["types"])
    pieces = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = np.array(["MPI" not in lb for lb in labels])
    np.testing.assert_array_equal(pieces[:, exact], loop[:, exact])
    np.testing.assert_allclose(pieces, loop, rtol=1e-12, atol=1e-300)
    np.random.seed(4)
    fruit.fit(X)                                   # new thresholds: kernels and tables are dropped
    assert slc._fused(T).pieces_loaded() == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "4"))))
def test_random_pieces_differential(fr, seed, monkeypatch):
    """Random plans (mixed word sets with shared prefixes, repeated words, negative exponents),
    semirings, weightings, sieves and series lengths, cut into random pieces (4 ... 24 nodes, units
    of one item or many): the plan in pieces against the record loop - bit for bit (60 cases run
    once in round 4: FRUITS_TEST_RANDOM_CASES=60)."""
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "0")
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.integers(1, 4))
    T = int(rng.choice([400, 600, 1024, 1100, 2100]))
    N = int(rng.choice([9, 16, 24]))
    X = rng.standard_normal((N, D, T)).cumsum(axis=2) / 5.0
    pool = []
    for w in range(1, 5):
        pool += [str(x) for x in fr.words.of_weight(w, dim=D)]
    pool += ["[1][-1][1]", "[1][1][1][1][1]", "[%d][%d]" % (D, D)]
    mode = fr.ISSMode.EXTENDED if rng.random() < 0.7 else fr.ISSMode.SINGLE
    picks = rng.choice(len(pool), size=min(int(rng.integers(25, 70)), len(pool)),
                       replace=mode == fr.ISSMode.SINGLE)
    words = [fr.words.SimpleWord(pool[i]) for i in picks]
    semi = rng.choice(["Reals", "Reals", "Arctic", "Bayesian"])
    W = fr.iss.weighting
    weighting = [None, W.Indices(), W.Indices(total=True), W.L1()][int(rng.integers(0, 4))]
    fruit = fr.Fruit(f"random pieces {seed}")
    if rng.random() < 0.7:
        fruit.add(fr.preparation.INC)
    fruit.add(fr.ISS(words, mode=mode, semiring=getattr(fr.iss.semiring, semi)(), weighting=weighting))
    S = fr.sieving
    sieves = [[S.NPI(q=(0.5, 1.0)), S.END()],
              [S.NPI(), S.MPI(), S.END(cut=[T // 2, -1])],
              [S.NPI(q=(0.25, 0.75), inc=2), S.MPI(q=(0.3, 1.0), inc=0), S.NPI(inc=0)],
              [S.NPI(cut=[T // 3, -1], q=(0.2, 0.6, 1.0)), S.NPI(inc=3), S.END()]][int(rng.integers(0, 4))]
    fruit.add(*sieves)
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(seed)
    fruit.fit(X)
    pipe = fruit.get_slice()._fused(T)
    if pipe is None or pipe.plan.nodes < 12:
        pytest.skip("not a fused pipeline / too small a plan")
    _debug_knobs(monkeypatch, pieces=0)
    pipe.prepare(N, plan_too=False)
    if pipe.jit_loaded() == 0:
        pytest.skip("no kernel of its own (hipRTC missing, or per-row sieve shapes)")
    loop = fruit.transform(X)
    piece = int(rng.integers(4, 25))
    _debug_knobs(monkeypatch, pieces=1, piece_min=8, piece_nodes=piece, piece_unit=int(rng.choice([0, 1, 40])))
    pipe.prepare(N)
    assert pipe.pieces_loaded() >= 1
    got = fruit.transform(X)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    exact = np.array(["MPI" not in lb for lb in labels])
    np.testing.assert_array_equal(got[:, exact], loop[:, exact])
    np.testing.assert_allclose(got, loop, rtol=1e-12, atol=1e-300)


def test_own_kernel_compiled_in_the_background(fr, tmp_path, monkeypatch):
    """Fruit.transform asks for a large pipeline's own kernel without waiting for the compiler:
    launches before the kernel is loaded take the generic instance, later ones the compiled one -
    all with the same features."""
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))   # (a cold cache: a real compilation)
    monkeypatch.setenv("FRUITS_HIP_JIT_BUNDLE", "")      # (and none of the kernels shipped with the build)
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "1")
    rng = np.random.default_rng(11)
    X = rng.standard_normal((1024, 2, 1024)).cumsum(axis=2) / 5.0
    fruit = fr.Fruit("background")
    fruit.add(fr.preparation.INC)
    fruit.add(fr.ISS(fr.words.of_weight(4, dim=2), mode=fr.ISSMode.EXTENDED,
                     weighting=fr.iss.weighting.Indices()))
    fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
    fruit.get_slice().fit_sample_size = 0.1
    np.random.seed(3)
    fruit.fit(X)
    first = fruit.transform(X)                       # 920 MiB of iterated sums: asks, does not wait
    pipe = fruit.get_slice()._fused(1024)
    assert getattr(pipe, "_pending", None) is not None
    during = [fruit.transform(X) for _ in range(3)]
    pipe._pending.result(timeout=120)
    loaded = pipe.jit_loaded()
    after = fruit.transform(X)
    for other in during + [after]:
        np.testing.assert_array_equal(other, first)
    if loaded == 0:
        pytest.skip("hipRTC is not installed")
    np.random.seed(3)
    fruit.fit(X)                                     # a refit drops the kernel with the thresholds
    again = fruit.get_slice()._fused(1024)
    assert again.jit_loaded() == 0
    # ... and what the first compilation left in the disk cache is there at once, without a compiler
    assert again.prepare_cached(X.shape[0]) >= 1
    np.testing.assert_array_equal(fruit.transform(X), first)


def test_fused_pipeline_is_cached_per_length(fr):
    """FruitSlice._fused keeps ONE pipeline per series length until the next fit - also for a
    slice with float (coquantile) cuts, whose cut columns must not replace the cache key."""
    rng = np.random.default_rng(5)
    X = rng.standard_normal((8, 1, 520)).cumsum(axis=2)
    fruit = fr.Fruit("cache")
    fruit.add(fr.preparation.INC, fr.ISS(fr.words.of_weight(2, dim=1), mode=fr.ISSMode.EXTENDED))
    fruit.add(fr.sieving.NPI(cut=[0.3, -1], q=(0.5, 1.0)), fr.sieving.END(cut=[0.5, -1]))
    fruit.get_slice().fit_sample_size = 1.0
    np.random.seed(0)
    fruit.fit(X)
    slc = fruit.get_slice()
    first = slc._fused(520)
    assert first is not None and slc._fused(520) is first
    fruit.transform(X)
    assert slc._fused(520) is first
    fruit.fit(X)
    assert slc._fused(520) is not first


@pytest.mark.parametrize("name", ["cfg3_small", "reduced_slice1_small", "readme", "readme_fullfit"])
def test_device_fit_equals_host_fit(fr, name, monkeypatch):
    """Thresholds fitted from device-selected order statistics are bit-identical to
    np.quantile on the downloaded rows."""
    case = [c for c in G.cases("fruit") if c["name"] == name][0]
    X = G[case["x"]]
    qs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FRUITS_AMD_DEVICE_FIT", flag)
        fruit = build_fruit(fr, case["spec"])
        np.random.seed(case["np_seed"])
        fruit.fit(X)
        qs.append([[np.asarray(sv._quantiles) for sv in row if hasattr(sv, "_quantiles")]
                   for slc in fruit._slices for row in slc._sieves_extended])
    assert len(qs[0]) == len(qs[1]) > 0
    for a, b in zip(qs[0], qs[1]):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


def test_select_ranks_against_numpy(fr):
    from fruits_amd import _native as nat
    rng = np.random.default_rng(9)
    A = rng.standard_normal((3, 7, 129))
    A[1, :, ::5] = 0.0            # ties
    A[2] *= 1e-300                # denormal range / sign handling
    Ad = nat.to_device(A)
    rows, incs, ranks, want = [], [], [], []
    for r in range(3):
        for inc in (0, 1, 2):
            flat = np.sort(orc.pre_transform(A[r], inc).ravel())
            for k in (0, 1, flat.size // 2, flat.size - 2, flat.size - 1):
                rows.append(r); incs.append(inc); ranks.append(k); want.append(flat[k])
    got = nat.select_ranks(Ad, rows, incs, ranks)
    np.testing.assert_array_equal(got, np.array(want))


def test_fit_of_many_slices(fr):
    """A fruit of more slices than selections may be in flight: Fruit.fit ends the oldest ones as
    it goes; the thresholds equal those of the slices fitted one by one."""
    rng = np.random.default_rng(2)
    X = rng.standard_normal((24, 2, 120)).cumsum(axis=2)
    def build():
        fruit = fr.Fruit()
        for i in range(11):
            fruit.cut()
            fruit.add(fr.ISS([fr.words.SimpleWord("[1][2]"), fr.words.SimpleWord("[%d]" % (1 + i % 2))],
                             mode=fr.ISSMode.EXTENDED))
            fruit.add(fr.sieving.NPI(q=(0.1 + 0.07 * i, 1.0), inc=i % 3), fr.sieving.END)
            fruit.get_slice().fit_sample_size = 1.0
        return fruit
    a, b = build(), build()
    np.random.seed(0)
    a.fit(X)
    np.random.seed(0)
    for slc in b:
        slc.fit(X)
    for sa, sb in zip(a, b):
        assert len(sa._sieves_extended) == len(sb._sieves_extended) == sa.niteratedsums()
        for ra, rb in zip(sa._sieves_extended, sb._sieves_extended):
            np.testing.assert_array_equal(ra[0]._quantiles, rb[0]._quantiles)
    b._fitted = True
    np.testing.assert_array_equal(a.transform(X), b.transform(X))


def test_selections_in_flight(fr):
    """fr_select_ranks_begin / _end: several selections queued behind one another (what Fruit.fit
    does with the slices of a fruit), ended in another order; every one owns a scratch blob until it
    is ended - the ninth in flight is refused (FR_E_LIMIT) and accepted once one has ended; a
    handle nobody asks is released when it is dropped."""
    from fruits_amd import _native as nat
    rng = np.random.default_rng(21)
    blocks, sels, wants = [], [], []
    for i in range(8):
        A = rng.standard_normal((2, 5 + i, 300 + 17 * i))
        if i % 3 == 0:
            A[0, :, ::2] = 0.25                                  # heavy ties: all eight digits
        Ad = nat.to_device(A)
        rows, incs, ranks, want = [], [], [], []
        for r in range(2):
            for inc in (0, 1):
                flat = np.sort(orc.pre_transform(A[r], inc).ravel())
                for k in (0, flat.size // 3, flat.size // 3 + 1, flat.size - 1):
                    rows.append(r); incs.append(inc); ranks.append(k); want.append(flat[k])
        blocks.append((Ad, rows, incs, ranks))
        sels.append(nat.Selection(Ad, rows, incs, ranks))
        wants.append(np.array(want))
    with pytest.raises(nat.NativeError, match="in flight"):
        nat.Selection(*blocks[0])
    for i in (5, 0, 7):
        np.testing.assert_array_equal(sels[i].result(), wants[i])
        np.testing.assert_array_equal(sels[i].result(), wants[i])     # (the values stay)
    again = nat.Selection(*blocks[0])                                  # a blob is free again
    np.testing.assert_array_equal(again.result(), wants[0])
    del sels[1], sels[2]                                               # dropped without a result()
    import gc
    gc.collect()
    np.testing.assert_array_equal(nat.select_ranks(*blocks[3]), wants[3])
    for i, s_ in enumerate(sels):
        s_.result()
    fr.release_scratch()
    np.testing.assert_array_equal(nat.select_ranks(*blocks[4]), wants[4])


# --------------------------------------------------------------------------
# BASELINE configs[2..4] at their full sizes: the experiment fruits verbatim
# --------------------------------------------------------------------------
_SIEVES7 = [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0}, {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
            {"kind": "NPI", "q": [0.5, 1.0], "inc": 2}, {"kind": "MPI", "q": [0.5, 1.0], "inc": 0},
            {"kind": "MPI", "q": [0.5, 1.0], "inc": 1}, {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
            {"kind": "END"}]


def _experiment_spec(fr, which):
    """experiments/fruit_reduced.py, fruit_general.py, fruit_twi.py as test specs."""
    def ow(w, d):
        return [str(x) for x in fr.words.of_weight(w, d)]

    def alt(n):
        return [str(x) for x in fr.words.alternate_sign([
            fr.words.SimpleWord(n * "[1]"), fr.words.SimpleWord(n * "[2]"),
            fr.words.SimpleWord((n // 2) * "[1][2]"), fr.words.SimpleWord((n // 2) * "[2][1]")])]
    new_inc = {"kind": "NEW", "inner": {"kind": "INC"}}
    if which == "twi":
        return {"name": "Time Warping Invariant Fruit", "slices": [
            {"preps": [{"kind": "INC"}],
             "iss": [{"words": ow(9, 1), "mode": "EXTENDED", "weighting": {"kind": "L1"}}],
             "sieves": [{"kind": "NPI"}, {"kind": "MPI"}, {"kind": "END"}], "fit_sample_size": 1.0},
            {"iss": [{"words": [str(x) for x in fr.words.alternate_sign(
                [fr.words.SimpleWord(48 * "[1]")])], "mode": "EXTENDED", "semiring": "Arctic"}],
             "sieves": [{"kind": "NPI"}, {"kind": "END"}], "fit_sample_size": 1.0}]}
    wr, na, cw = (4, 24, 3) if which == "reduced" else (6, 48, 4)
    cos_words = sum((ow(w, 2) for w in range(1, cw + 1)), [])
    return {"name": which, "slices": [
        {"preps": [new_inc, {"kind": "STD"}],
         "iss": [{"words": ow(wr, 2), "mode": "EXTENDED", "weighting": {"kind": "Indices"}}],
         "sieves": _SIEVES7, "fit_sample_size": 1.0},
        {"preps": [new_inc],
         "iss": [{"words": alt(na), "mode": "EXTENDED", "semiring": "Arctic"}],
         "sieves": _SIEVES7, "fit_sample_size": 1.0}] + [
        {"preps": [new_inc, {"kind": "STD"}],
         "iss": [{"kind": "CosWISS", "words": cos_words,
                  "freqs": [i / 20 for i in range(1, 11, 2)], "exponent": e,
                  "total_weighting": True, "mode": "SINGLE"}],
         "sieves": _SIEVES7, "fit_sample_size": 1.0} for e in (1, 2)]}


def _spread_rows(N, first, count=64):
    """`count` series outside the fit sample, spread over the whole batch so that every
    residue n mod 8 occurs (workgroups are dealt round-robin over the 8 XCDs) and every
    region of the unit order - first and last spans of the persistent grid - is hit."""
    step = max(((N - first) // count) // 8 * 8, 8)     # a multiple of 8: the residue is r % 8
    rows = np.array([first + r * step + (r % 8) for r in range(count)])
    rows = np.unique(np.clip(rows, first, N - 1))
    assert len(set(rows % 8)) == 8
    return np.concatenate([rows, [N - 1]]) if rows[-1] != N - 1 else rows


@pytest.mark.parametrize("which,shape", [("reduced", (2048, 3, 1024)),
                                         ("general", (8192, 3, 1024)),
                                         ("twi", (8192, 6, 4096))],
                         ids=["config3_fruit_reduced", "config4_fruit_general", "config5_fruit_twi"])
def test_experiment_fruits_full_size(fr, which, shape):
    """BASELINE configs[2], [3], [4] at their full batch sizes on one GPU (config 5 names no
    N: 8192): experiments/fruit_reduced.py, fruit_general.py, fruit_twi.py
    verbatim.  Fitted on the first 24 series, transformed as ONE batch (every slice one
    fused launch); 64+ series spread over the batch are checked (a) against the numpy
    oracle fitted on the same sample - counts exactly wherever no element sits on a
    threshold, and SURVEY.md section 7's bar on top (<= 1 count on <= 0.1 % of the entries) -
    and (b) through batch independence."""
    rng = np.random.default_rng(len(which))
    X = rng.standard_normal(shape).cumsum(axis=2) / 8.0
    spec = _experiment_spec(fr, which)
    fruit = build_fruit(fr, spec)
    n_fit = 24
    np.random.seed(5)
    fruit.fit(X[:n_fit])
    T = shape[2]
    for slc in fruit:
        assert slc._fused(T) is not None          # every slice is ONE fused launch
    feats = fruit.transform(X)
    assert feats.shape == (shape[0], fruit.nfeatures())
    assert np.isfinite(feats).all()
    # series outside the fit sample: a fitted extreme quantile IS a data point of the
    # sample, i.e. an exact threshold tie for the series that holds it
    idx = _spread_rows(shape[0], n_fit)
    labels = [fruit.label(i) for i in range(fruit.nfeatures())]
    Xs, Xfit = np.ascontiguousarray(X[idx]), X[:n_fit].copy()
    del X
    # (b) batch independence (MPI band sums are accumulated with float atomics)
    alone = fruit.transform(Xs)
    np.testing.assert_allclose(feats[idx], alone, rtol=1e-10, atol=1e-12)
    # (a) the oracle, fitted on the same sample, on the selected series
    np.random.seed(5)
    ofit = orc.fruit_fit(spec, Xfit)
    ref, expo = orc.fruit_transform_exposure(spec, ofit, Xs)
    rec = compare_features(feats[idx], ref, labels, expo, what=f"full size {which} {shape}")
    assert rec["max_d"] <= 1 and rec["differ"] <= 1e-3 * rec["entries"], rec
    # (c) the two halves separately: the fitted thresholds, and the transform with the oracle's
    # thresholds under the tight exposure - SURVEY.md section 7's bar again
    fit_parity(fruit, ofit, f"full size {which}")
    transplant_thresholds(fruit, ofit)
    means = {}
    ref, expo = orc.fruit_transform_exposure(spec, ofit, Xs, rel=1e-13, tight=True, means=means)
    for rec in compare_strict(fruit.transform(Xs), ref, labels, expo, means,
                              what=f"full size {which} {shape}", max_plus=_max_plus_columns(spec, fruit)):
        assert rec["max_d"] <= 1 and rec["differ"] <= 1e-3 * rec["entries"], rec


def test_word_sharded_config4_full_size(fr):
    """BASELINE configs[3] at its full size, the word list of fruit_general's first slice
    (of_weight(6,2), 956 words, K = 1351) sharded over 8 ranks run one after the other on
    this GPU with a loop-back gather (fruits_amd.parallel): every rank's share is ONE fused
    launch and the re-assembled (N, F) matrix equals the unsharded transform bit for bit
    (NPI counts and END values; no float atomics in these sieves)."""
    import torch
    from fruits_amd import parallel as par
    from fruits_amd.cache import SharedSeedCache
    N, D, T = 8192, 3, 1024
    X = np.random.default_rng(4).standard_normal((N, D, T)).cumsum(axis=2) / 8.0
    fruit = fr.Fruit("general slice 1")
    fruit.add(fr.preparation.INC)
    iss = fr.ISS(fr.words.of_weight(6, 2), mode=fr.ISSMode.EXTENDED,
                 weighting=fr.iss.weighting.Indices())
    fruit.add(iss, fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
    slc = fruit.get_slice()
    slc.fit_sample_size = 1.0
    np.random.seed(0)
    fruit.fit(X[:64])
    ref = fruit.transform(X)
    assert ref.shape == (N, 2702)
    strings = [str(w) for w in iss.words]
    depths = [iss._depth(i) for i in range(len(strings))]
    per_sum = sum(s.nfeatures() for s in slc.get_sieves())
    world = 8
    parts = par.shard_words(strings, depths, world)
    assert sorted(i for p in parts for i in p) == list(range(len(strings)))
    maps = par.column_map(parts, depths, per_sum)
    cache = SharedSeedCache(X)
    out = torch.zeros((N, slc.nfeatures()), dtype=torch.float64, device="cuda")
    for r in range(world):
        assert slc._fused(T, indices=parts[r]) is not None
        block = par._device_block(slc, iss, cache.input_device(X), cache, parts[r], depths, per_sum)
        assert block.shape == (N, len(maps[r]))
        out[:, torch.as_tensor(maps[r], device="cuda")] = block
    np.testing.assert_array_equal(np.nan_to_num(out.cpu().numpy()), ref)


@pytest.mark.parametrize("wd", [(1, 1), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (3, 1)], ids=str)
@pytest.mark.parametrize("mode", ["EXTENDED", "SINGLE"])
def test_static_programs(fr, monkeypatch, wd, mode):
    """The pre-compiled static programs of the standard word sets (walk_static_inst.hip):
    against the C oracle, and against the record interpreter (same sums; a*b+c may contract
    differently, so 1e-12 row-wise instead of bit equality).  Batches below and above one
    resident round, N not a multiple of 8 (plain unit numbering), T below the chunk (bounds
    checks), several launches in a row (the register prefetch of single-group programs)."""
    from fruits_amd import _native as nat
    w, d = wd
    words = fr.words.of_weight(w, dim=d)
    iss = fr.ISS(words, mode=getattr(fr.ISSMode, mode))
    plan = iss._plan(0, len(words))
    assert plan.static_schedule(1) is not None
    strs = [str(x) for x in words]
    # ((1601, 448): series of 385 ... 512 elements run the program too on cache-sized batches,
    # half of its 1024-element chunk idle - round 4)
    for N, T, dist in ((37, 1024, "normal"), (1601, 1024, "uniform"), (64, 600, "normal"), (3080, 1022, "normal"),
                       (1601, 448, "uniform")):
        X = gen_input({"seed": N + T, "dist": dist, "shape": [N, d, T]})
        Xd = nat.to_device(X)
        monkeypatch.setenv("FRUITS_HIP_STATIC", "1")
        got = nat.to_host(iss.transform_device(Xd))
        again = nat.to_host(iss.transform_device(Xd))
        np.testing.assert_array_equal(got, again)
        monkeypatch.setenv("FRUITS_HIP_STATIC", "0")
        interp = nat.to_host(iss.transform_device(Xd))
        rowwise_close(got, interp, rtol=1e-12)
        if N <= 1601:
            ref = corc.iss_transform(X, strs, mode)
            rowwise_close(got, ref)
            if dist == "uniform":
                np.testing.assert_allclose(got, ref, rtol=RTOL)


def test_static_program_is_what_runs(fr, monkeypatch):
    """A word list that compiles to the records of a standard set runs the static kernel
    whatever it was built from; an explicit group count the schedule was not generated for,
    a weighting or T <= 384 fall back to the interpreter - all with the same results."""
    from fruits_amd import _native as nat
    X = gen_input({"seed": 11, "dist": "normal", "shape": [520, 3, 1024]})
    Xd = nat.to_device(X)
    ws = [fr.words.SimpleWord(str(w)) for w in fr.words.of_weight(2, dim=3)]
    iss = fr.ISS(ws, mode=fr.ISSMode.EXTENDED)
    a = nat.to_host(iss.transform_device(Xd))
    for groups in (1, 2, 3, 9):
        b = nat.to_host(iss.transform_device(Xd, groups=groups))
        rowwise_close(a, b, rtol=1e-12)
    monkeypatch.setenv("FRUITS_HIP_STATIC", "0")
    c = nat.to_host(iss.transform_device(Xd))
    rowwise_close(a, c, rtol=1e-12)


def test_float_cuts_run_fused(fr, monkeypatch):
    """Coquantile (float) cuts are per-series boundaries: the fused launch reads them from a
    table (fr_pipeline_set_series_cuts) - same features as the materialising sieve kernels,
    NPI / MPI pairs with the same cuts still merged, a batch of another size re-arms the
    table, END range errors as in the unfused path."""
    rng = np.random.default_rng(77)
    T = 300
    X = rng.standard_normal((21, 2, T)).cumsum(axis=2)

    def build():
        fruit = fr.Fruit()
        fruit.add(fr.preparation.INC)
        fruit.add(fr.ISS(fr.words.of_weight(3, 2), mode=fr.ISSMode.EXTENDED,
                         weighting=fr.iss.weighting.Indices()))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), cut=[0.3, 100, -1]))
        fruit.add(fr.sieving.MPI(q=(0.5, 1.0), cut=[0.3, 100, -1]))
        fruit.add(fr.sieving.NPI(q=(0.25, 0.75), cut=0.7, inc=2))
        fruit.add(fr.sieving.END(cut=[0.5, 0.05, -1]))
        for slc in fruit:
            slc.fit_sample_size = 1.0
        return fruit
    fused = build()
    np.random.seed(3)
    fused.fit(X)
    slc = fused.get_slice()
    assert slc._fusable() and slc._fused(T) is not None and slc._fused(T)._cut_slots == 4 + 2 + 4
    a = fused.transform(X)
    b = fused.transform(X[:7])                 # another batch size: the table follows
    # (band means are sums of atomics: their last bit may depend on the launch shape)
    np.testing.assert_allclose(a[:7], b, rtol=1e-12, atol=1e-13)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build()
    np.random.seed(3)
    plain.fit(X)
    c = plain.transform(X)
    labels = [fused.label(i) for i in range(fused.nfeatures())]
    counts = np.array([("NPI" in s) for s in labels])
    np.testing.assert_array_equal(a[:, counts], c[:, counts])
    np.testing.assert_allclose(a[:, ~counts], c[:, ~counts], rtol=1e-9, atol=1e-12)


def _require_hiprtc(plan):
    """Skips a test of run-time compiled programs when hipRTC is not installed."""
    try:
        plan.jit(1, compile_only=True)
    except ValueError as e:
        if "not available" in str(e):
            pytest.skip("hipRTC is not installed")
        raise


def test_jit_static_program(fr, monkeypatch, tmp_path):
    """A plan outside the standard word sets: fr_plan_prepare compiles its static program at
    run time (hipRTC); results bit-identical to the interpreter's, equal to the oracle's;
    FRUITS_HIP_JIT=2 compiles at the first run instead; =0 leaves the interpreter."""
    from fruits_amd import _native as nat
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))
    strs = ["[1][2]", "[12][1]", "[2]", "[1][1][2]", "[3][1]", "[33]", "[2][3][1]"]
    words = [fr.words.SimpleWord(s) for s in strs]

    def fresh():
        iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
        return iss, iss._plan(0, len(words))
    for N, T, dist in ((53, 1024, "normal"), (1601, 1024, "uniform"), (3080, 1022, "normal")):
        X = gen_input({"seed": N + T, "dist": dist, "shape": [N, 3, T]})
        Xd = nat.to_device(X)
        monkeypatch.setenv("FRUITS_HIP_JIT", "0")
        iss, plan = fresh()
        plan.prepare(N, T)
        assert plan.jit_loaded() == 0
        interp = nat.to_host(iss.transform_device(Xd))
        monkeypatch.setenv("FRUITS_HIP_JIT", "1")
        iss, plan = fresh()
        _require_hiprtc(plan)
        plan.prepare(N, T)
        assert plan.jit_loaded() >= 1
        got = nat.to_host(iss.transform_device(Xd))
        np.testing.assert_array_equal(got, interp)
        np.testing.assert_array_equal(got, nat.to_host(iss.transform_device(Xd)))
        if N <= 1601:
            rowwise_close(got, corc.iss_transform(X, strs, "EXTENDED"))
        monkeypatch.setenv("FRUITS_HIP_JIT", "2")         # no prepare: compiled at the first run
        iss, plan = fresh()
        lazy = nat.to_host(iss.transform_device(Xd))
        assert plan.jit_loaded() >= 1
        np.testing.assert_array_equal(lazy, interp)
    assert len(os.listdir(tmp_path / "jit")) >= 2


def test_jit_repeated_words(fr, monkeypatch, tmp_path):
    """The metric's "48 weight-2 words" (words[i % 15], SINGLE): nodes with up to four output
    rows - immediates of the run-time compiled program; identical to the interpreter."""
    from fruits_amd import _native as nat
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))
    w15 = fr.words.of_weight(2, dim=3)
    w48 = [w15[i % 15] for i in range(48)]
    X = gen_input({"seed": 48, "dist": "normal", "shape": [1032, 3, 1024]})
    Xd = nat.to_device(X)
    monkeypatch.setenv("FRUITS_HIP_JIT", "0")
    interp = nat.to_host(fr.ISS(w48).transform_device(Xd))
    monkeypatch.setenv("FRUITS_HIP_JIT", "1")
    iss = fr.ISS(w48)
    plan = iss._plan(0, 48)
    _require_hiprtc(plan)
    plan.prepare(1032, 1024)
    assert plan.jit_loaded() == 2
    got = nat.to_host(iss.transform_device(Xd))
    np.testing.assert_array_equal(got, interp)
    for i in range(15, 48):
        np.testing.assert_array_equal(got[i], got[i % 15])


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "24"))))
def test_random_jit_differential(fr, seed, monkeypatch, tmp_path):
    """Random small word lists (positive exponents, repeated words, shared prefixes, both modes)
    through their run-time compiled static program (FRUITS_HIP_JIT=2: compiled at the first
    run): bit-identical to the interpreter, equal to the C oracle.  Lists the scheduler does
    not accept (more than 32 nodes, 4 rows or 4 open prefixes, letters of more than 4
    factors) simply stay on the interpreter - also checked."""
    from fruits_amd import _native as nat
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.integers(1, 5))
    N = int(rng.choice([1, 5, 8, 9, 24, 40, 777, 1600]))
    T = int(rng.choice([514, 600, 700, 1000, 1022, 1024]))
    words = [_random_word(rng, D).replace("-", "") for _ in range(int(rng.integers(1, 9)))]
    if rng.random() < 0.4:
        words += [words[0], words[-1], words[0]]            # repeated words: several output rows
    mode = "EXTENDED" if rng.random() < 0.6 else "SINGLE"
    dist = "uniform" if rng.random() < 0.5 else "normal"
    X = gen_input({"seed": seed, "dist": dist, "shape": [N, D, T]})
    if dist == "normal":
        X = X / 2.0
    Xd = nat.to_device(X)

    def run(jit):
        monkeypatch.setenv("FRUITS_HIP_JIT", jit)
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=getattr(fr.ISSMode, mode))
        plan = iss._plan(0, len(words))
        return nat.to_host(iss.transform_device(Xd)), plan
    interp, _ = run("0")
    got, plan = run("2")
    qualifies = plan.static_schedule(1) is not None and plan.static_program_index(1) == 0
    if qualifies and plan.jit_loaded() == 0:
        _require_hiprtc(plan)
    assert plan.jit_loaded() == (0 if not qualifies else min(2, 1 + (plan.info(nat.FR_INFO_GROUPS) > 1)))
    np.testing.assert_array_equal(got, interp)
    ref = corc.iss_transform(X, words, mode)
    rowwise_close(got, ref)


@pytest.mark.parametrize("words,D", [(["[4]"], 4), (["[3]"], 3), (["[2]", "[3]"], 3),
                                     (["[22]", "[23]", "[33]", "[2][2]", "[2][3]", "[3][2]", "[3][3]"], 3),
                                     (["[3]", "[1]"], 3)])
def test_static_program_other_dimensions(fr, monkeypatch, words, D, tmp_path):
    """Word lists with the SHAPE of a standard set on other input dimensions: they must not
    run that set's pre-compiled program (whose row sources are baked in) - the interpreter
    or their own run-time compiled program, equal to the oracle either way."""
    from fruits_amd import _native as nat
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))
    X = gen_input({"seed": 5 + D, "dist": "uniform", "shape": [530, D, 1024]})
    Xd = nat.to_device(X)
    ref = corc.iss_transform(X, words, "EXTENDED")
    for jit in ("0", "2"):
        monkeypatch.setenv("FRUITS_HIP_JIT", jit)
        iss = fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED)
        got = nat.to_host(iss.transform_device(Xd))
        rowwise_close(got, ref)
        np.testing.assert_allclose(got, ref, rtol=RTOL)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FRUITS_TEST_RANDOM_CASES", "12"))))
def test_select_ranks_random(fr, seed):
    """fr_select_ranks on random row blocks against a sort: sizes on both sides of the 2048
    candidates a workgroup settles, heavy ties (one value many times - settled in the gather
    pass; a few values many times - the histogram passes go on), plateaus of a running
    maximum, tiny / huge / negative values, neighbouring ranks (the successor search) and the
    extremes."""
    from fruits_amd import _native as nat
    rng = np.random.default_rng(4242 + seed)
    R = int(rng.integers(1, 4))
    N = int(rng.choice([1, 3, 17, 64, 300]))
    T = int(rng.choice([2, 33, 257, 1024, 1500]))
    A = rng.standard_normal((R, N, T))
    kind = seed % 6
    if kind == 1:
        A = np.round(A * 2) / 2                              # a handful of distinct values
    elif kind == 2:
        A[rng.random(A.shape) < 0.7] = 0.0                   # mostly ONE value
    elif kind == 3:
        A = np.maximum.accumulate(A, axis=2)                 # plateaus
    elif kind == 4:
        A *= 10.0 ** rng.integers(-300, 300, size=(R, 1, 1))
    elif kind == 5:
        A[:, :, ::3] = 1.5
        A[:, :, 1::3] = -2.25                                # two values many times + noise
    Ad = nat.to_device(A)
    rows, incs, ranks, want = [], [], [], []
    for r in range(R):
        for inc in (0, 1, 2):
            flat = np.sort(orc.pre_transform(A[r], inc).ravel())
            n = flat.size
            ks = {0, n - 1, n // 2, max(n // 2 - 1, 0), min(n // 2 + 1, n - 1), n // 4, (3 * n) // 4,
                  int(rng.integers(0, n)), int(rng.integers(0, n))}
            for k in sorted(ks):
                rows.append(r); incs.append(inc); ranks.append(k); want.append(flat[k])
    got = nat.select_ranks(Ad, rows, incs, ranks)
    np.testing.assert_array_equal(got, np.array(want))


@pytest.mark.parametrize("semiring", ["Reals", "Arctic", "Bayesian"])
@pytest.mark.parametrize("T", [300, 1024, 2100])
def test_total_weighting_increments_fused(fr, monkeypatch, semiring, T):
    """Increments (inc = 1, 2) of TOTALLY weighted sums in the fused epilogue: the sieves see
    c[t] (x) w[t], so the increment needs the weight one step to the left - same counts as the
    materialising sieve kernels on the stored rows, one and several time chunks."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((19, 2, T)).cumsum(axis=2) / np.sqrt(T)
    words = ["[1]", "[1][2]", "[12][1]", "[2][2][1]", "[2]"]

    def build():
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord(s) for s in words], mode=fr.ISSMode.EXTENDED,
                         semiring=getattr(fr.iss.semiring, semiring)(),
                         weighting=fr.iss.weighting.Indices(scale=2.0, total=True)))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=1, cut=[T // 3, -1]))
        fruit.add(fr.sieving.MPI(q=(0.25, 0.75), inc=2))
        fruit.add(fr.sieving.NPI(inc=0), fr.sieving.END)
        for slc in fruit:
            slc.fit_sample_size = 1.0
        return fruit
    fused = build()
    np.random.seed(1)
    fused.fit(X)
    assert fused.get_slice()._fused(T) is not None
    a = fused.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build()
    np.random.seed(1)
    plain.fit(X)
    b = plain.transform(X)
    labels = [fused.label(i) for i in range(fused.nfeatures())]
    counts = np.array([("NPI" in s) for s in labels])
    np.testing.assert_array_equal(a[:, counts], b[:, counts])
    np.testing.assert_allclose(a[:, ~counts], b[:, ~counts], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("semiring", ["Reals", "Arctic"])
@pytest.mark.parametrize("T", [900, 2100])
def test_total_weighting_high_orders_fused(fr, monkeypatch, semiring, T):
    """Both rare variants of the fused epilogue at once: differencing orders above 2 and cumulated
    rows of TOTALLY weighted sums, on one and on several time chunks (the instantiations that are
    TOTALINC and HIGHORD) - same counts as the materialising sieve kernels on the stored rows."""
    rng = np.random.default_rng(T + len(semiring))
    X = rng.standard_normal((17, 2, T)).cumsum(axis=2) / np.sqrt(T)

    def build():
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord(s) for s in ["[1]", "[1][2]", "[12][1]", "[2][2][1]"]],
                         mode=fr.ISSMode.EXTENDED, semiring=getattr(fr.iss.semiring, semiring)(),
                         weighting=fr.iss.weighting.Indices(scale=2.0, total=True)))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=3), fr.sieving.NPI(q=(0.4, 1.0), inc=4, cut=[T // 3, -1]))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=1), fr.sieving.NPI(q=(0.3, 1.0), inc=-1), fr.sieving.END)
        for slc in fruit:
            slc.fit_sample_size = 1.0
        return fruit
    fused = build()
    np.random.seed(1)
    fused.fit(X)
    assert fused.get_slice()._fused(T) is not None
    a = fused.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build()
    np.random.seed(1)
    plain.fit(X)
    b = plain.transform(X)
    labels = [fused.label(i) for i in range(fused.nfeatures())]
    compare_features(a, b, labels, count_frac=0.02)


@pytest.mark.parametrize("T", [200, 700, 1024, 1500, 2100, 4200])
def test_high_order_increments_fused(fr, monkeypatch, T):
    """NPI / MPI with inc = 3 ... 8 (IncrementSieve._pre_transform applies the increments inc
    times): fused - same counts as the materialising kernels - also on series of several time
    chunks (every order carries its last value of a chunk to the next one)."""
    rng = np.random.default_rng(T)
    X = rng.standard_normal((13, 2, T)).cumsum(axis=2)

    def build():
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord(s) for s in ["[1]", "[1][2]", "[12][1]", "[2][2][1]"]],
                         mode=fr.ISSMode.EXTENDED))
        for inc in (3, 4, 8):
            fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=inc, cut=[T // 2, -1]))
        fruit.add(fr.sieving.MPI(q=(0.25, 0.75), inc=5), fr.sieving.NPI(inc=6), fr.sieving.END)
        for slc in fruit:
            slc.fit_sample_size = 1.0
        return fruit
    fused = build()
    np.random.seed(2)
    fused.fit(X)
    assert fused.get_slice()._fused(T) is not None
    a = fused.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build()
    np.random.seed(2)
    plain.fit(X)
    b = plain.transform(X)
    labels = [fused.label(i) for i in range(fused.nfeatures())]
    counts = np.array([("NPI" in s) for s in labels])
    np.testing.assert_array_equal(a[:, counts], b[:, counts])
    # (band means and END values: the fused walk holds four consecutive elements per lane, the
    # materialising one two pieces of two - the same sums in another association)
    np.testing.assert_allclose(a[:, ~counts], b[:, ~counts], rtol=1e-9,
                               atol=1e-11 * max(1.0, float(np.abs(b[:, ~counts]).max())))
    monkeypatch.delenv("FRUITS_AMD_FUSED")


@pytest.mark.parametrize("semiring", ["Reals", "Arctic"])
@pytest.mark.parametrize("T", [200, 1024, 1500, 2300])
def test_cumulated_rows_fused(fr, monkeypatch, semiring, T):
    """NPI / MPI with inc < 0 (the row cumulated -inc times, fruits/sieving/increment.py:68-70):
    fused, also on series of several time chunks (every cumulation carries its running sum);
    against the materialising path, whose cumulation is the same parallel sum in another
    association (counts on exact ties may move by one)."""
    rng = np.random.default_rng(T + len(semiring))
    X = rng.standard_normal((11, 2, T)) / np.sqrt(T)

    def build():
        fruit = fr.Fruit()
        fruit.add(fr.ISS([fr.words.SimpleWord(s) for s in ["[1]", "[1][2]", "[2][1][1]"]],
                         mode=fr.ISSMode.EXTENDED, semiring=getattr(fr.iss.semiring, semiring)()))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0), inc=-1, cut=[T // 2, -1]))
        fruit.add(fr.sieving.MPI(q=(0.25, 0.75), inc=-2), fr.sieving.NPI(q=(0.3, 1.0), inc=-3), fr.sieving.END)
        for slc in fruit:
            slc.fit_sample_size = 1.0
        return fruit
    fused = build()
    np.random.seed(4)
    fused.fit(X)
    assert fused.get_slice()._fused(T) is not None
    a = fused.transform(X)
    monkeypatch.setenv("FRUITS_AMD_FUSED", "0")
    plain = build()
    np.random.seed(4)
    plain.fit(X)
    b = plain.transform(X)
    labels = [fused.label(i) for i in range(fused.nfeatures())]
    compare_features(a, b, labels, count_frac=0.02)


def test_zz_parity_bars():
    """The bars over ALL strict comparisons of this run (the oracle's thresholds in the GPU
    pipeline, tight exposure), by class of column.  Sums (Reals, CosWISS): SURVEY.md section 7 -
    counts differ by at most 1, on at most 0.1 % of the entries; and on batches of 64 series and
    more (the full-size fruits: below that, ONE data point on a threshold already exposes 1 / N
    of a column's entries) at most 5 % of the entries have an element near a threshold at all
    (observed 4.0 %: increments of a few ulp of the running sum next to the threshold 0 - the
    nearly absorbed summands of fruit_twi's L1-weighted 9-letter words and of fruit_general's
    deep Indices-weighted words; none of them differs).
    Max-plus columns (Arctic / Bayesian) tie by construction (plateaus of a running maximum,
    two thirds of the entries are exposed) - and are formed exactly like the reference's, from
    bit-equal inputs: no entry may differ at all."""
    from conftest import STRICT_REPORT
    sums = [r for r in STRICT_REPORT if r["cls"] == "sum"]
    if len(sums) < 20:
        pytest.skip("only part of the suite ran")
    entries = sum(r["entries"] for r in sums)
    differ = sum(r["differ"] for r in sums)
    assert max(r["max_d"] for r in sums) <= 1
    assert differ <= 1e-3 * entries, (differ, entries)
    big = [r for r in sums if r["series"] >= 64]
    if big:
        assert sum(r["exposed"] for r in big) <= 5e-2 * sum(r["entries"] for r in big), big
    maxplus = [r for r in STRICT_REPORT if r["cls"] == "max-plus"]
    assert maxplus and sum(r["differ"] for r in maxplus) == 0, maxplus


def test_bundled_kernels_serve_a_cold_cache(fr, tmp_path, monkeypatch):
    """The kernels shipped with the build (fruits_amd/jit_bundle, fruits_amd/gen_bundle.py): with
    an EMPTY user cache and without asking the compiler (fr_pipeline_prepare_cached never
    compiles) the pipelines of BASELINE configs 3 and 4 get their own kernels - the plan as
    straight-line code / in pieces - and compute what the generic kernel computes."""
    import fruits_amd.gen_bundle as gb
    if not gb.up_to_date():
        pytest.skip("fruits_amd/jit_bundle is not built (python -m fruits_amd.gen_bundle)")
    if "pieces=0" in os.environ.get("FRUITS_HIP_DEBUG", "") or os.environ.get("FRUITS_HIP_JIT", "1") == "0":
        pytest.skip("the knob sweep switched the bundled kernels' paths off")
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "empty"))
    monkeypatch.setenv("FRUITS_AMD_AUTO_PREPARE", "0")
    rng = np.random.default_rng(8)
    for weight, N in ((4, 1536), (6, 96)):
        X = rng.standard_normal((N, 3, 1024)).cumsum(axis=2) / 5.0
        fruit = fr.Fruit("bundled")
        fruit.add(fr.preparation.INC)
        fruit.add(fr.ISS(fr.words.of_weight(weight, dim=2), mode=fr.ISSMode.EXTENDED,
                         weighting=fr.iss.weighting.Indices()))
        fruit.add(fr.sieving.NPI(q=(0.5, 1.0)), fr.sieving.END)
        fruit.get_slice().fit_sample_size = 64 / N
        np.random.seed(2)
        fruit.fit(X)
        generic = fruit.transform(X)
        pipe = fruit.get_slice()._fused(1024)
        assert pipe.jit_loaded() == 0 and pipe.pieces_loaded() == 0
        pipe.prepare_cached(N)
        assert pipe.fully_compiled(), "the bundle did not serve this pipeline"
        assert not os.path.isdir(tmp_path / "empty") or os.listdir(tmp_path / "empty") == []
        np.testing.assert_array_equal(fruit.transform(X), generic)
