#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference package.

TEST INFRASTRUCTURE - runs only in the build container, never on the GPU box
(the reference tree at /root/reference does not travel).  The script imports
the reference (pure Python + numba) with two loader accommodations that change
no arithmetic:

  * ``numba`` is not installed here, so a stand-in module is registered whose
    ``njit`` returns the decorated function unchanged and whose ``prange`` is
    ``range`` - every njit body in the reference is plain numpy code and runs
    un-jitted;
  * numpy >= 2 dropped ``np.NINF`` (used by reference fruits/sieving/segment.py:72,83),
    so ``numpy.NINF = -numpy.inf`` is set before the import.

Only DATA is written: inputs (or the seed that makes them), word strings,
plans, and the arrays the reference returned.  No reference source is copied.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FRUITS_REFERENCE", "/root/reference")


def _install_loader_shims():
    if not hasattr(np, "NINF"):
        np.NINF = -np.inf
    try:
        import numba  # noqa: F401
        return
    except ImportError:
        pass
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.jit = njit
    nb.prange = range
    sys.modules["numba"] = nb


_install_loader_shims()
sys.path.insert(0, REF)
import fruits  # noqa: E402  (the reference)

arrays = {}
manifest = {"iss": [], "inc": [], "sieve": [], "fruit": [], "words": {},
            "cacheplan": [], "l1": []}


def put(name, arr):
    assert name not in arrays, name
    arrays[name] = np.ascontiguousarray(arr)
    return name


X_1 = np.array([
    [[-4, 0.8, 0, 5, -3], [2.0, 1, 0, 0, -7]],
    [[5.0, 8, 2, 6, 0], [-5, -1, -4, -0.5, -8]],
])
put("X_1", X_1)


# --------------------------------------------------------------------------
# words / cache plans
# --------------------------------------------------------------------------
def words_case(w, d):
    ws = fruits.words.of_weight(w, dim=d)
    strs = [str(x) for x in ws]
    plan = fruits.iss.CachePlan(ws)._plan
    manifest["words"][f"{w},{d}"] = {
        "words": strs,
        "plan": [int(p) for p in plan],
        "K": int(sum(plan)),
        "exps": [[list(map(int, el)) for el in x] for x in ws],
    }


for (w, d) in [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (9, 1),
               (1, 2), (2, 2), (3, 2), (4, 2), (6, 2), (2, 3), (3, 3)]:
    words_case(w, d)

cp_words = ["[1][11][3][11]", "[11][13][11][1][3]", "[1][13][1]",
            "[11][13][111][13][11]", "[3][11][111]", "[1][11][2]",
            "[11][2]", "[11][13][111][13][2]", "[3][11][1112][21]"]
cp = fruits.iss.CachePlan([fruits.words.SimpleWord(s) for s in cp_words])
manifest["cacheplan"].append({
    "words": cp_words, "plan": [int(p) for p in cp._plan],
    "labels": [cp.get_word_string(i) for i in range(cp.n_iterated_sums())],
})
# words whose strings differ but exponents agree are NOT merged by the plan
cp_words2 = ["[12][1]", "[21][2]", "[12]", "[21][1][1]", "[1]", "[1][2]"]
cp2 = fruits.iss.CachePlan([fruits.words.SimpleWord(s) for s in cp_words2])
manifest["cacheplan"].append({
    "words": cp_words2, "plan": [int(p) for p in cp2._plan],
    "labels": [cp2.get_word_string(i) for i in range(cp2.n_iterated_sums())],
})

parse_cases = ["[-12][-2-21]", "[(-11)(-11)(11)][25]", "[122(10)(62)][(24)5]",
               "[11][122]", "[1]", "[-1][-2]", "[(10)12345][9][23]",
               "[-1-12][(-11)3]"]
manifest["parse"] = {
    s: [list(map(int, el)) for el in fruits.words.SimpleWord(s)]
    for s in parse_cases
}
w_ = fruits.words.SimpleWord("[-12][-2-21]")
w_.multiply("[(-11)(-11)(11)][25]")
manifest["parse_multiply"] = {
    "first": "[-12][-2-21]", "second": "[(-11)(-11)(11)][25]",
    "name": str(w_), "exps": [list(map(int, el)) for el in w_],
}
manifest["alternate_sign"] = {
    "in": ["[1][1][1]", "[1][2][1][2]", "[11][2]"],
    "out": [str(w) for w in fruits.words.alternate_sign(
        [fruits.words.SimpleWord(s) for s in ["[1][1][1]", "[1][2][1][2]",
                                              "[11][2]"]])],
}


# --------------------------------------------------------------------------
# ISS cases
# --------------------------------------------------------------------------
def make_weighting(spec):
    if spec is None:
        return None
    kind = spec["kind"]
    kw = {k: v for k, v in spec.items() if k != "kind"}
    return getattr(fruits.iss.weighting, kind)(**kw)


def iss_case(name, x_key, words, mode="SINGLE", alphas=None, weighting=None,
             store_slice=None, semiring="Reals", argmax=False):
    X = arrays[x_key] if isinstance(x_key, str) else x_key[1]
    ws = [fruits.words.SimpleWord(s) for s in words]
    if alphas is not None:
        for w, a in zip(ws, alphas):
            if a is not None:
                w.alpha = a
    sr = fruits.semiring.Arctic(argmax=True) if argmax else getattr(fruits.semiring, semiring)()
    iss = fruits.ISS(ws, mode=getattr(fruits.ISSMode, mode), semiring=sr,
                     weighting=make_weighting(weighting))
    out = iss.fit_transform(X)
    entry = {
        "name": name, "words": list(words), "mode": mode, "semiring": semiring,
        "alphas": alphas, "weighting": weighting,
        "K": int(out.shape[0]),
    }
    if argmax:
        # (labels of an argmax ISS: CachePlan.get_word_string only knows the rows of the
        # plain EXTENDED plan - the reference raises beyond them; not pinned)
        entry["argmax"] = True
        assert out.shape[0] == iss.n_iterated_sums()
    else:
        entry["labels"] = [iss.label(i) for i in range(iss.n_iterated_sums())]
    if isinstance(x_key, str):
        entry["x"] = x_key
    else:
        entry["x_gen"] = x_key[0]
    if store_slice is not None:
        entry["series"] = list(store_slice)
        out = out[:, list(store_slice), :]
    entry["out"] = put(f"iss/{name}", out)
    manifest["iss"].append(entry)
    return out


def gen(spec):
    """Input described by a generator spec so large inputs need not be stored."""
    rng = np.random.default_rng(spec["seed"])
    if spec["dist"] == "uniform":
        X = rng.random(tuple(spec["shape"]))
    elif spec["dist"] == "normal":
        X = rng.standard_normal(tuple(spec["shape"]))
    else:
        raise ValueError
    return X


# hand-computed goldens of reference tests/signature/test_simple.py:11-34
iss_case("x1_six_words", "X_1", ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"])
put("iss/x1_six_words_expected", np.array([
    [[-4, -3.2, -3.2, 1.8, -1.2], [5, 13, 15, 21, 21]],
    [[2, 3, 3, 3, -4], [-5, -6, -10, -10.5, -18.5]],
    [[16, 16.64, 16.64, 41.64, 50.64], [25, 89, 93, 129, 129]],
    [[-8, -7.2, -7.2, -7.2, 13.8], [-25, -33, -41, -44, -44]],
    [[0, -3.2, -3.2, -19.2, -24.6], [0, 40, 66, 156, 156]],
    [[0., -4., -4., -4., -16.6], [0, -5, -57, -64.5, -232.5]],
]))

put("U_6_3_40", np.random.default_rng(11).random((6, 3, 40)))
put("G_5_3_37", np.random.default_rng(12).standard_normal((5, 3, 37)))
put("U_4_1_64", np.random.default_rng(13).random((4, 1, 64)))
put("U_3_10_50", np.random.default_rng(14).random((3, 10, 50)))
put("U_7_2_129", np.random.default_rng(15).random((7, 2, 129)))
put("P_4_2_33", np.random.default_rng(16).random((4, 2, 33)) + 0.5)

W23 = manifest["words"]["2,3"]["words"]
W32 = manifest["words"]["3,2"]["words"]
W42 = manifest["words"]["4,2"]["words"]
W41 = manifest["words"]["4,1"]["words"]

iss_case("w23_single_U", "U_6_3_40", W23, "SINGLE")
iss_case("w23_ext_U", "U_6_3_40", W23, "EXTENDED")
iss_case("w23_ext_G", "G_5_3_37", W23, "EXTENDED")
iss_case("w32_ext_U", "U_7_2_129", W32, "EXTENDED")
iss_case("w42_ext_U", "U_7_2_129", W42, "EXTENDED")
iss_case("w41_ext_U", "U_4_1_64", W41, "EXTENDED")
iss_case("cacheplan_ext", "U_6_3_40", cp_words, "EXTENDED")
iss_case("cacheplan2_ext", "U_6_3_40", cp_words2, "EXTENDED")
iss_case("ext_simple_1", "U_6_3_40", ["[11][21][331][22]"], "EXTENDED")
iss_case("ext_simple_2", "U_4_1_64", ["[1][11][111][1111]"], "EXTENDED")
iss_case("neg_words", "P_4_2_33", ["[-1][-2]", "[-12][-2-21]", "[1][-1]",
                                   "[-1-1][22]"], "SINGLE")
iss_case("neg_words_ext", "P_4_2_33", ["[-12][-2-21][2]", "[-12][1]"],
         "EXTENDED")
iss_case("many_dims", "U_3_10_50", ["[(10)12345][9][23]", "[(10)][(10)1]"],
         "EXTENDED")
iss_case("dup_words_single", "U_6_3_40", [W23[i % 15] for i in range(48)],
         "SINGLE")

# weighted (reference tests/signature/test_weighting.py:6-100)
iss_case("idx_total_alpha", "U_6_3_40", ["[12][2][33]"], "EXTENDED",
         alphas=[[.6, .2, .5]],
         weighting={"kind": "Indices", "scale": 1, "total": True})
iss_case("idx_nontotal_alpha", "U_3_10_50", ["[(10)12345][9][23]"], "EXTENDED",
         alphas=[[.45, 3.14, .3]],
         weighting={"kind": "Indices", "scale": 1, "total": False})
iss_case("idx_default_w32", "U_7_2_129", W32, "EXTENDED",
         weighting={"kind": "Indices"})
iss_case("idx_default_w42_single", "U_7_2_129", W42[:20], "SINGLE",
         weighting={"kind": "Indices"})
iss_case("idx_total_w32", "U_7_2_129", W32, "EXTENDED",
         weighting={"kind": "Indices", "total": True, "scale": 5.0})
iss_case("idx_notrelative", "U_6_3_40", ["[1][2][3]", "[11][2]"], "EXTENDED",
         weighting={"kind": "Indices", "relative": False, "scale": 2.0})
iss_case("idx_mixed_alpha", "U_6_3_40", ["[1][2][3]", "[1][2][1]", "[1][3]"],
         "EXTENDED", alphas=[[1., .5, .25], [1., .75, .25], None],
         weighting={"kind": "Indices", "scale": 3.0})
iss_case("l1_alpha", "U_6_3_40", ["[12][2][33]"], "SINGLE",
         alphas=[[.6, .2, .3]],
         weighting={"kind": "L1", "scale": 1, "total": False, "relative": True})
iss_case("l1_default_w41", "U_4_1_64", W41, "EXTENDED",
         weighting={"kind": "L1"})
iss_case("l1_total_G", "G_5_3_37", ["[1][2]", "[1][2][3]", "[3]"], "EXTENDED",
         weighting={"kind": "L1", "total": True, "scale": 4.0})
iss_case("l1_on_prepared", "G_5_3_37", ["[1][2]", "[2][2][1]"], "EXTENDED",
         weighting={"kind": "L1", "on_prepared": True, "scale": 10.0})

# config 1 (BASELINE.json configs[0]) - README identity
c1 = {"seed": 0, "dist": "uniform", "shape": [200, 3, 100]}
o = iss_case("config1", ("gen", gen(c1)), ["[11]"], "SINGLE")
manifest["iss"][-1]["x_gen"] = c1
assert np.allclose(o[0], np.cumsum(gen(c1)[:, 0, :] ** 2, axis=1))

# config 2 (BASELINE.json configs[1]) - a few series of the full-size run
c2 = {"seed": 0, "dist": "normal", "shape": [2048, 3, 1024]}
iss_case("config2_ext", ("gen", gen(c2)), W23, "EXTENDED",
         store_slice=[0, 1, 1023, 2047])
manifest["iss"][-1]["x_gen"] = c2
c2u = {"seed": 1, "dist": "uniform", "shape": [2048, 3, 1024]}
iss_case("config2_ext_uniform", ("gen", gen(c2u)), W23, "EXTENDED",
         store_slice=[0, 2047])
manifest["iss"][-1]["x_gen"] = c2u

# Arctic semiring (max, +): reference tests/signature/test_semiring.py:10-33 + random
iss_case("arctic_x1_six_words", "X_1", ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"],
         semiring="Arctic")
put("iss/arctic_x1_six_words_expected", np.array([
    [[-4, 0.8, 0.8, 5, 5], [5, 8, 8, 8, 8]],
    [[2, 2, 2, 2, 2], [-5, -1, -1, -0.5, -0.5]],
    [[-8, 1.6, 1.6, 10, 10], [10, 16, 16, 16, 16]],
    [[-2, 1.8, 1.8, 5, 5], [0, 7, 7, 7, 7]],
    [[-8, 1.6, 1.6, 10, 10], [10, 16, 16, 16, 16]],
    [[-2, 1.8, 1.8, 5., 5.], [0., 7., 7., 7.5, 7.5]],
]))
ALT = [str(w) for w in fruits.words.alternate_sign([
    fruits.words.SimpleWord(6 * "[1]"), fruits.words.SimpleWord(6 * "[2]"),
    fruits.words.SimpleWord(3 * "[1][2]"), fruits.words.SimpleWord(3 * "[2][1]")])]
manifest["alt_words"] = ALT
iss_case("arctic_w23_ext_G", "G_5_3_37", W23, "EXTENDED", semiring="Arctic")
iss_case("arctic_w32_single_U", "U_7_2_129", W32, "SINGLE", semiring="Arctic")
iss_case("arctic_alt_ext_G", "G_5_3_37", ALT, "EXTENDED", semiring="Arctic")
iss_case("arctic_neg_words", "P_4_2_33", ["[-1][-2]", "[-12][-2-21]", "[1][-1]", "[-1-1][22]",
                                           "[111][-2-2-2]"], "EXTENDED", semiring="Arctic")
iss_case("arctic_idx_nontotal", "U_6_3_40", ["[12][2][33]", "[1][3]"], "EXTENDED",
         alphas=[[.6, .2, .5], None], weighting={"kind": "Indices", "scale": 2.0},
         semiring="Arctic")
iss_case("arctic_idx_total", "U_6_3_40", ["[12][2][33]", "[1][3]"], "EXTENDED",
         alphas=[[.6, .2, .5], None], weighting={"kind": "Indices", "scale": 2.0, "total": True},
         semiring="Arctic")
iss_case("arctic_l1_G", "G_5_3_37", ["[1][2]", "[1][2][3]", "[3]"], "EXTENDED",
         weighting={"kind": "L1", "scale": 4.0}, semiring="Arctic")
c2a = {"seed": 0, "dist": "normal", "shape": [2048, 3, 1024]}
iss_case("arctic_config2_ext", ("gen", gen(c2a)), W23, "EXTENDED", store_slice=[0, 2047],
         semiring="Arctic")
manifest["iss"][-1]["x_gen"] = c2a

# Arctic with argmax=True (fruits/iss/semiring.py:239-284): every prefix of every word, its
# running maximum followed by the back-tracked positions of the maxima, EXTENDED mode only
manifest["iss_argmax"] = []
_n_plain = len(manifest["iss"])
iss_case("argmax_x1_six_words", "X_1", ["[1]", "[2]", "[11]", "[12]", "[1][1]", "[1][2]"],
         "EXTENDED", semiring="Arctic", argmax=True)
iss_case("argmax_w23_G", "G_5_3_37", W23, "EXTENDED", semiring="Arctic", argmax=True)
iss_case("argmax_alt_G", "G_5_3_37", ALT, "EXTENDED", semiring="Arctic", argmax=True)
iss_case("argmax_neg_words_P", "P_4_2_33", ["[-1][-2]", "[-12][-2-21]", "[111][-2-2-2][1]"],
         "EXTENDED", semiring="Arctic", argmax=True)
iss_case("argmax_idx_U", "U_6_3_40", ["[12][2][33]", "[1][3]"], "EXTENDED",
         alphas=[[.6, .2, .5], None], weighting={"kind": "Indices", "scale": 2.0},
         semiring="Arctic", argmax=True)
iss_case("argmax_l1_G", "G_5_3_37", ["[1][2]", "[1][2][3]", "[3]"], "EXTENDED",
         weighting={"kind": "L1", "scale": 4.0}, semiring="Arctic", argmax=True)
iss_case("argmax_long_U", "U_7_2_129", ["[1][2][1][2][1]", "[2][2][2]"], "EXTENDED",
         semiring="Arctic", argmax=True)
manifest["iss_argmax"] = manifest["iss"][_n_plain:]
del manifest["iss"][_n_plain:]

# Bayesian semiring (max, x): reference tests/signature/test_weighting.py:220-278 word sets + random
iss_case("bayes_w23_ext_U", "U_6_3_40", W23, "EXTENDED", semiring="Bayesian")
iss_case("bayes_w32_single_U", "U_7_2_129", W32, "SINGLE", semiring="Bayesian")
iss_case("bayes_neg_words", "P_4_2_33", ["[-1][-2]", "[-12][-2-21]", "[1][-1]", "[-1-1][22]",
                                          "[111][-2-2-2]"], "EXTENDED", semiring="Bayesian")
iss_case("bayes_l2_nontotal", "U_6_3_40", ["[12][3][2213]", "[1][3]"], "EXTENDED",
         alphas=[[.6, .2, .3], None], weighting={"kind": "L2", "scale": 1.0},
         semiring="Bayesian")
iss_case("bayes_l2_total", "U_6_3_40", ["[12][3][2213]", "[1][3]"], "EXTENDED",
         alphas=[[.6, .2, .3], None], weighting={"kind": "L2", "scale": 1.0, "total": True},
         semiring="Bayesian")
iss_case("bayes_idx_G", "G_5_3_37", ["[1][2]", "[1][2][3]", "[3]"], "EXTENDED",
         weighting={"kind": "Indices", "scale": 3.0}, semiring="Bayesian")

# CosWISS ("next" row): frequencies exactly representable in float32 AND with an exact
# float32 product freq*(T-1), so the un-jitted run agrees with numba's f4->f8 promotion
manifest["coswiss"] = []


def cos_case(name, x_key, words, freqs, **kw):
    X = arrays[x_key]
    cw = fruits.CosWISS([fruits.words.SimpleWord(s) for s in words], freqs, **kw)
    out = cw.fit_transform(X)
    manifest["coswiss"].append({
        "name": name, "x": x_key, "words": list(words), "freqs": list(freqs), "kw": kw,
        "labels": [cw.label(i) for i in range(cw.n_iterated_sums())],
        "weightings": {s: fruits.CosWISS([fruits.words.SimpleWord(s)], [0.5], **kw)
                       ._get_weightings(fruits.words.SimpleWord(s)).tolist() for s in words},
        "out": put(f"cos/{name}", out)})


put("U_5_3_65", np.random.default_rng(41).random((5, 3, 65)))   # T-1 = 64
cos_case("e1", "U_5_3_65", ["[1][23]", "[12][2][33]"], [0.5], exponent=1)
cos_case("e1_total", "U_5_3_65", ["[1][23]", "[12][2][33]"], [0.5], exponent=1,
         total_weighting=True)
cos_case("e2_multi", "U_5_3_65", ["[1]", "[12]", "[1][2]", "[2][13]", "[1][2][3]"],
         [0.5, 0.25, 2.0], exponent=2)
cos_case("e2_total_multi", "U_5_3_65", ["[1]", "[1][2]", "[11][2][3]"], [0.25, 1.0],
         exponent=2, total_weighting=True)
cos_case("e3", "U_5_3_65", ["[1][-2]", "[3][1][1]"], [0.5], exponent=3)

# the randomised variants (fruits/iss/cos.py:51-164,248-263): the weights / dropped indices
# drawn by CosWISS._fit from numpy's global generator are stored with the outputs
manifest["coswiss_random"] = []


def cos_random_case(name, x_key, words, freqs, np_seed, **kw):
    X = arrays[x_key]
    cw = fruits.CosWISS([fruits.words.SimpleWord(s) for s in words], freqs, **kw)
    np.random.seed(np_seed)
    cw.fit(X)
    out = cw.transform(X)
    entry = {"name": name, "x": x_key, "words": list(words), "freqs": list(freqs), "kw": kw,
             "np_seed": np_seed, "out": put(f"cosr/{name}", out)}
    if kw.get("ffn_size") is not None:
        entry["A"], entry["b"], entry["C"] = (put(f"cosr/{name}_A", cw._A),
                                              put(f"cosr/{name}_b", cw._b),
                                              put(f"cosr/{name}_C", cw._C))
    if kw.get("dropout") is not None:
        entry["dropout_indices"] = put(f"cosr/{name}_drop", cw._dropout_indices)
    manifest["coswiss_random"].append(entry)


cos_random_case("drop_e2", "U_5_3_65", ["[1][2]", "[2][13]", "[1][2][3]"], [0.5, 0.25], 11,
                exponent=2, dropout=0.1)
cos_random_case("drop_e1_total", "U_5_3_65", ["[1]", "[12][2][33]"], [0.5], 12, exponent=1,
                total_weighting=True, dropout=0.2)
cos_random_case("ffn3_e2", "U_5_3_65", ["[1][2]", "[2][13]"], [0.5, 2.0], 13, exponent=2,
                ffn_size=3)
cos_random_case("ffn9_e1_total", "U_5_3_65", ["[1][23]", "[3]"], [0.25], 14, exponent=1,
                total_weighting=True, ffn_size=9)

# per-word operator (iterated_sum_fast, fruits/iss/semiring.py:203-219)
Z = arrays["U_6_3_40"]
word = fruits.words.SimpleWord("[12][2][33]")
res = fruits.semiring.Reals().iterated_sum_fast(
    Z, np.array(list(word), dtype=np.int32),
    np.array([.6, .2, .5], dtype=np.float32),
    fruits.iss.weighting.Indices(scale=2.0).get_lookup(Z), 2, False)
put("op/fast_nontotal_E2", res)
res = fruits.semiring.Reals().iterated_sum_fast(
    Z, np.array(list(word), dtype=np.int32),
    np.array([.6, .2, .5], dtype=np.float32),
    fruits.iss.weighting.Indices(scale=2.0).get_lookup(Z), 3, True)
put("op/fast_total_E3", res)
put("op/lookup", fruits.iss.weighting.Indices(scale=2.0).get_lookup(Z))


# --------------------------------------------------------------------------
# L1 lookup
# --------------------------------------------------------------------------
def l1_case(name, x_key, **kw):
    X = arrays[x_key]
    wt = fruits.iss.weighting.L1(**kw)
    wt._cache = fruits.cache.SharedSeedCache(X)
    manifest["l1"].append({"name": name, "x": x_key, "kw": kw,
                           "out": put(f"l1/{name}", wt.get_lookup(X))})


l1_case("default", "U_6_3_40")
l1_case("relative", "G_5_3_37", relative=True, scale=1)
Xc = arrays["U_4_1_64"].copy()
Xc[1] = 0.25  # a constant series -> all-zero lookup row
put("U_4_1_64_const", Xc)
l1_case("const_row", "U_4_1_64_const", scale=7.0)


# L2 path length and Plateaus staircases (same kernel consumes them; host / device lookups)
manifest["lookups"] = []


def lookup_case(name, x_key, kind, **kw):
    X = arrays[x_key]
    wt = getattr(fruits.iss.weighting, kind)(**kw)
    wt._cache = fruits.cache.SharedSeedCache(X)
    manifest["lookups"].append({"name": name, "x": x_key, "kind": kind, "kw": kw,
                                "out": put(f"lookup/{name}", wt.get_lookup(X))})


lookup_case("l2_default", "U_6_3_40", "L2")
lookup_case("l2_relative", "G_5_3_37", "L2", relative=True, scale=1)
lookup_case("plateaus4", "U_6_3_40", "Plateaus", n=4)
lookup_case("plateaus3_rev", "G_5_3_37", "Plateaus", n=3, reverse=True, scale=2.0)
lookup_case("plateaus7", "U_7_2_129", "Plateaus", n=7, scale=1.0)

# --------------------------------------------------------------------------
# INC
# --------------------------------------------------------------------------
def inc_case(name, x_key, **kw):
    out = fruits.preparation.INC(**kw).fit_transform(arrays[x_key])
    manifest["inc"].append({"name": name, "x": x_key, "kw": kw,
                            "out": put(f"inc/{name}", out)})


inc_case("x1_default", "X_1")
inc_case("x1_nopad", "X_1", zero_padding=False)
inc_case("u_default", "U_6_3_40")
inc_case("u_shift3", "U_6_3_40", shift=3)
inc_case("u_depth2", "U_6_3_40", depth=2)
inc_case("u_shift2_depth2_nopad", "U_7_2_129", shift=2, depth=2,
         zero_padding=False)


# --------------------------------------------------------------------------
# sieves on (N, T) arrays
# --------------------------------------------------------------------------
def sieve_case(name, kind, x_key, fit_key=None, **kw):
    X = arrays[x_key]
    sv = getattr(fruits.sieving, kind)(**kw)
    if fit_key is None:
        out = sv.fit_transform(X)
    else:
        sv.fit(arrays[fit_key])
        out = sv.transform(X)
    kwj = {k: (list(v) if isinstance(v, (tuple, list)) else v)
           for k, v in kw.items()}
    manifest["sieve"].append({
        "name": name, "kind": kind, "x": x_key, "fit": fit_key, "kw": kwj,
        "quantiles": [float(q) if np.isfinite(q) else str(q)
                      for q in sv._quantiles] if hasattr(sv, "_quantiles") else None,
        "labels": [sv.label(i) for i in range(sv.nfeatures())],
        "out": put(f"sieve/{name}", out)})


put("X_1_0", X_1[0])
put("X_1_1", X_1[1])
put("S_8_50", np.random.default_rng(21).standard_normal((8, 50)).cumsum(axis=1))
put("S_5_33", np.random.default_rng(22).standard_normal((5, 33)))
sieve_case("end_x10", "END", "X_1_0")
sieve_case("end_cut_group_int", "END", "X_1_0", cut=[1, 4, -1])
sieve_case("end_s", "END", "S_8_50", cut=[10, -1, 25])
sieve_case("npi_x10", "NPI", "X_1_0")
sieve_case("npi_cut3", "NPI", "X_1_0", cut=3)
sieve_case("npi_cut_group", "NPI", "X_1_1", cut=[-1, 3, 1])
sieve_case("npi_s_default", "NPI", "S_8_50")
sieve_case("npi_s_q", "NPI", "S_8_50", q=(0.5, 1.0))
sieve_case("npi_s_q_multi", "NPI", "S_8_50", q=(-1.0, 0.25, 0.0, 0.75, 1.0),
           cut=[20, -1])
sieve_case("npi_s_inc0", "NPI", "S_8_50", q=(0.5, 1.0), inc=0)
sieve_case("npi_s_inc2", "NPI", "S_8_50", q=(0.5, 1.0), inc=2)
sieve_case("npi_s_incneg", "NPI", "S_5_33", q=(0.3, 1.0), inc=-1)
sieve_case("npi_fit_other", "NPI", "S_8_50", fit_key="S_8_50", q=(0.1, 0.9),
           cut=[5, 17, -1])
sieve_case("mpi_s_default", "MPI", "S_8_50")
sieve_case("mpi_s_q", "MPI", "S_8_50", q=(0.5, 1.0), inc=2, cut=[20, -1])
sieve_case("mpi_x11", "MPI", "X_1_1", cut=[-1, 3, 1])
# float ("coquantile") cuts: per-series boundaries from the path length of the input
# (fruits/sieving/segment.py:51-64, fruits/cache.py:16-40)
sieve_case("end_s_coq", "END", "S_8_50", cut=[0.5, 0.25, -1])
sieve_case("npi_s_coq", "NPI", "S_8_50", q=(0.5, 1.0), cut=[0.3, -1])
sieve_case("npi_s_coq_l1", "NPI", "S_8_50", q=(0.25, 0.75, 1.0), cut=[10, 0.6, -1], inc=0,
           coquantile_norm="L1")
sieve_case("mpi_s_coq", "MPI", "S_8_50", q=(0.5, 1.0), cut=[0.2, 0.8], inc=2)


# --------------------------------------------------------------------------
# whole-fruit pipelines
# --------------------------------------------------------------------------
def build_fruit(spec):
    fr = fruits.Fruit(spec.get("name", ""))
    for sl in spec["slices"]:
        fr.cut()
        for p in sl.get("preps", []):
            kind = p["kind"]
            kw = {k: v for k, v in p.items() if k not in ("kind", "inner")}
            if kind == "NEW":
                inner = p.get("inner")
                inner_obj = None if inner is None else getattr(
                    fruits.preparation, inner["kind"])(
                        **{k: v for k, v in inner.items() if k != "kind"})
                fr.add(fruits.preparation.NEW(inner_obj))
            else:
                fr.add(getattr(fruits.preparation, kind)(**kw))
        for i in sl["iss"]:
            ws = [fruits.words.SimpleWord(s) for s in i["words"]]
            if i.get("kind") == "CosWISS":
                fr.add(fruits.CosWISS(freqs=i["freqs"], words=ws, exponent=i.get("exponent", 2),
                                      total_weighting=i.get("total_weighting", False)))
                continue
            fr.add(fruits.ISS(ws, mode=getattr(fruits.ISSMode, i["mode"]),
                              semiring=getattr(fruits.semiring, i.get("semiring", "Reals"))(),
                              weighting=make_weighting(i.get("weighting"))))
        for s in sl["sieves"]:
            kw = {k: (tuple(v) if isinstance(v, list) and k == "q" else v)
                  for k, v in s.items() if k != "kind"}
            fr.add(getattr(fruits.sieving, s["kind"])(**kw))
        if "fit_sample_size" in sl:
            fr.get_slice().fit_sample_size = sl["fit_sample_size"]
    return fr


def fruit_case(name, x_key, spec, np_seed=None, x_test_key=None):
    X = arrays[x_key]
    fr = build_fruit(spec)
    if np_seed is not None:
        np.random.seed(np_seed)
    fr.fit(X)
    out = fr.transform(X)
    entry = {"name": name, "x": x_key, "spec": spec, "np_seed": np_seed,
             "nfeatures": int(fr.nfeatures()),
             "labels": [fr.label(i) for i in range(fr.nfeatures())],
             "labels_v2": [fr.label(i, verbose=2)
                           for i in range(min(fr.nfeatures(), 12))],
             "summary": fr.summary(),
             "out": put(f"fruit/{name}", out)}
    if x_test_key is not None:
        entry["x_test"] = x_test_key
        entry["out_test"] = put(f"fruit/{name}_test",
                                fr.transform(arrays[x_test_key]))
    manifest["fruit"].append(entry)


put("U_20_3_60", np.random.default_rng(31).random((20, 3, 60)))
put("U_9_3_60", np.random.default_rng(32).random((9, 3, 60)))
put("G_16_3_96", np.random.default_rng(33).standard_normal((16, 3, 96)))
put("G_12_2_80", np.random.default_rng(34).standard_normal((12, 2, 80)))
put("G_10_1_128", np.random.default_rng(35).standard_normal((10, 1, 128)))

readme_spec = {"name": "My Fruit", "slices": [
    {"preps": [{"kind": "INC"}],
     "iss": [{"words": W23, "mode": "EXTENDED"}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "END"}]},
    {"iss": [{"words": W23, "mode": "EXTENDED"}],
     "sieves": [{"kind": "NPI"}, {"kind": "END"}]},
]}
fruit_case("readme", "U_20_3_60", readme_spec, np_seed=1234,
           x_test_key="U_9_3_60")
readme_full = json.loads(json.dumps(readme_spec))
for sl in readme_full["slices"]:
    sl["fit_sample_size"] = 1.0
fruit_case("readme_fullfit", "G_16_3_96", readme_full, np_seed=7)

# reference tests/core/test_branches.py:61-86 uses MAX/MIN; END variant here
fruit_case("x1_two_slices_end", "X_1", {"slices": [
    {"iss": [{"words": ["[1]", "[2]", "[11]"], "mode": "SINGLE"}],
     "sieves": [{"kind": "END"}]},
    {"iss": [{"words": ["[12]", "[1][1]", "[1][2]"], "mode": "SINGLE"}],
     "sieves": [{"kind": "END"}, {"kind": "NPI"}]},
]}, np_seed=14)

# BASELINE config 3 shape of pipeline (fruit_reduced slice 1 on the hot path)
cfg3 = {"name": "cfg3", "slices": [
    {"preps": [{"kind": "INC"}],
     "iss": [{"words": W42, "mode": "EXTENDED",
              "weighting": {"kind": "Indices"}}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0]}, {"kind": "END"}],
     "fit_sample_size": 1.0}]}
fruit_case("cfg3_small", "G_12_2_80", cfg3, np_seed=8)
cfg3u = json.loads(json.dumps(cfg3))
cfg3u["slices"][0]["iss"][0].pop("weighting")
fruit_case("cfg3_small_unweighted", "G_12_2_80", cfg3u, np_seed=9)

# fruit_reduced slice 1 verbatim shape (NEW(INC), STD, NPI/MPI inc 0..2, END)
red = {"name": "reduced1", "slices": [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
     "iss": [{"words": W42, "mode": "EXTENDED",
              "weighting": {"kind": "Indices"}}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0},
                {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
                {"kind": "NPI", "q": [0.5, 1.0], "inc": 2},
                {"kind": "MPI", "q": [0.5, 1.0], "inc": 0},
                {"kind": "MPI", "q": [0.5, 1.0], "inc": 1},
                {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
                {"kind": "END"}],
     "fit_sample_size": 1.0}]}
fruit_case("reduced_slice1_small", "G_10_1_128", red, np_seed=10)

# BASELINE config 5 shape (fruit_twi slice 1): INC -> ISS(of_weight(w,1), L1) -> NPI, MPI, END
twi = {"name": "twi", "slices": [
    {"preps": [{"kind": "INC"}],
     "iss": [{"words": manifest["words"]["5,1"]["words"], "mode": "EXTENDED",
              "weighting": {"kind": "L1"}}],
     "sieves": [{"kind": "NPI"}, {"kind": "MPI"}, {"kind": "END"}],
     "fit_sample_size": 1.0}]}
fruit_case("twi_small", "G_10_1_128", twi, np_seed=11)
twi_hot = json.loads(json.dumps(twi))
twi_hot["slices"][0]["sieves"] = [{"kind": "NPI"}, {"kind": "END"}]
fruit_case("twi_small_hot", "G_16_3_96", twi_hot, np_seed=12)

# fruit_twi slice 2 shape: Arctic ISS over alternate_sign chains -> NPI, END
fruit_case("twi_arctic_small", "G_10_1_128", {"name": "twi2", "slices": [
    {"iss": [{"words": [str(w) for w in fruits.words.alternate_sign(
        [fruits.words.SimpleWord(12 * "[1]")])], "mode": "EXTENDED", "semiring": "Arctic"}],
     "sieves": [{"kind": "NPI"}, {"kind": "END"}], "fit_sample_size": 1.0}]}, np_seed=15)
# fruit_reduced slice 2 shape: NEW(INC) -> Arctic ISS -> NPI/MPI inc 0..2, END
fruit_case("reduced_arctic_small", "G_10_1_128", {"name": "red2", "slices": [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}],
     "iss": [{"words": ALT, "mode": "EXTENDED", "semiring": "Arctic"}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0},
                {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
                {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
                {"kind": "END"}],
     "fit_sample_size": 1.0}]}, np_seed=16)

# fruit_reduced slice 3 shape: NEW(INC) -> STD -> CosWISS -> NPI/MPI, END
put("G_8_1_65", np.random.default_rng(42).standard_normal((8, 1, 65)))
fruit_case("reduced_coswiss_small", "G_8_1_65", {"name": "red3", "slices": [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
     "iss": [{"kind": "CosWISS", "words": manifest["words"]["1,2"]["words"]
              + manifest["words"]["2,2"]["words"], "freqs": [0.5, 0.25], "exponent": 1,
              "total_weighting": True, "mode": "SINGLE"}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0},
                {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
                {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
                {"kind": "END"}],
     "fit_sample_size": 1.0}]}, np_seed=17)

# experiments/fruit_reduced.py VERBATIM (all four slices) on a small batch; T-1 = 64 keeps the
# float32 product freq*(T-1) exact for the experiment's frequencies i/20
SIEVES7 = [{"kind": "NPI", "q": [0.5, 1.0], "inc": 0}, {"kind": "NPI", "q": [0.5, 1.0], "inc": 1},
           {"kind": "NPI", "q": [0.5, 1.0], "inc": 2}, {"kind": "MPI", "q": [0.5, 1.0], "inc": 0},
           {"kind": "MPI", "q": [0.5, 1.0], "inc": 1}, {"kind": "MPI", "q": [0.5, 1.0], "inc": 2},
           {"kind": "END"}]
ALT24 = [str(w) for w in fruits.words.alternate_sign([
    fruits.words.SimpleWord(24 * "[1]"), fruits.words.SimpleWord(24 * "[2]"),
    fruits.words.SimpleWord(12 * "[1][2]"), fruits.words.SimpleWord(12 * "[2][1]")])]
COSW = (manifest["words"]["1,2"]["words"] + manifest["words"]["2,2"]["words"]
        + manifest["words"]["3,2"]["words"])
reduced = {"name": "Reduced Fruit", "slices": [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
     "iss": [{"words": W42, "mode": "EXTENDED", "weighting": {"kind": "Indices"}}],
     "sieves": SIEVES7, "fit_sample_size": 1.0},
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}],
     "iss": [{"words": ALT24, "mode": "EXTENDED", "semiring": "Arctic"}],
     "sieves": SIEVES7, "fit_sample_size": 1.0}] + [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}, {"kind": "STD"}],
     "iss": [{"kind": "CosWISS", "words": COSW, "freqs": [i / 20 for i in range(1, 11, 2)],
              "exponent": e, "total_weighting": True, "mode": "SINGLE"}],
     "sieves": SIEVES7, "fit_sample_size": 1.0} for e in (1, 2)]}
put("G_6_1_65", np.random.default_rng(43).standard_normal((6, 1, 65)).cumsum(axis=2))
fruit_case("fruit_reduced_verbatim", "G_6_1_65", reduced, np_seed=18)
# ... and the experiment file itself gives the same features as the spec above
import importlib.util
_sp = importlib.util.spec_from_file_location(
    "ref_fruit_reduced", os.path.join(REF, "experiments", "fruit_reduced.py"))
_mod = importlib.util.module_from_spec(_sp)
_sp.loader.exec_module(_mod)
np.random.seed(18)
_mod.fruit.fit(arrays["G_6_1_65"])
_out = _mod.fruit.transform(arrays["G_6_1_65"])
_case = [c for c in manifest["fruit"] if c["name"] == "fruit_reduced_verbatim"][0]
assert np.array_equal(_out, arrays[_case["out"]], equal_nan=True), "spec != experiments/fruit_reduced.py"
assert _mod.fruit.nfeatures() == _case["nfeatures"]

# chained ISS (reference tests/signature/test_consecutive.py) with END
# float cuts through a whole fruit (the fused epilogue reads them from a per-series table)
fruit_case("coquantile_cuts", "G_16_3_96", {"name": "coq", "slices": [
    {"preps": [{"kind": "INC"}],
     "iss": [{"words": manifest["words"]["2,3"]["words"], "mode": "EXTENDED",
              "weighting": {"kind": "Indices"}}],
     "sieves": [{"kind": "NPI", "q": [0.5, 1.0], "cut": [0.3, -1]},
                {"kind": "MPI", "q": [0.5, 1.0], "cut": [0.3, -1]},
                {"kind": "NPI", "q": [0.25, 0.75, 1.0], "cut": [0.5, 20, 0.9], "inc": 2},
                {"kind": "END", "cut": [0.25, 0.5, -1]}],
     "fit_sample_size": 1.0}]}, np_seed=19)
fruit_case("coquantile_cuts_arctic", "G_10_1_128", {"name": "coq2", "slices": [
    {"preps": [{"kind": "NEW", "inner": {"kind": "INC"}}],
     "iss": [{"words": ALT, "mode": "EXTENDED", "semiring": "Arctic"}],
     "sieves": [{"kind": "NPI", "cut": [0.4, 0.7]}, {"kind": "END", "cut": [0.6, -1]}],
     "fit_sample_size": 1.0}]}, np_seed=20)

fruit_case("consecutive_end", "U_9_3_60", {"slices": [
    {"iss": [{"words": ["[12][1]", "[1][32]", "[11][121][3]"], "mode": "EXTENDED"},
             {"words": ["[11]", "[111]", "[111][1][11]", "[1][1][11]"],
              "mode": "EXTENDED"}],
     "sieves": [{"kind": "END"}, {"kind": "NPI"}]}]}, np_seed=13)

np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(manifest, f, indent=1)
tot = sum(a.nbytes for a in arrays.values())
print(f"wrote {len(arrays)} arrays ({tot/1e6:.2f} MB raw) and manifest")
