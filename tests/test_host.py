"""CPU-only tests: host logic of fruits_amd (no compute) and the C ABI surface."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
import fruits_amd as fr
from fruits_amd import _native as nat
from oracle import ref_numpy as orc

G = load_golden()
M = G.manifest


# ---------------------------------------------------------------- C ABI surface
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fruits_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = nat.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libfruits_hip.so does not export {name}"
    assert set(nat.EXPORTS) == set(names)
    assert lib.fr_version() >= 100


def test_no_cpu_fallback():
    if nat.device_count() > 0:
        pytest.skip("a HIP device is present")
    X = np.zeros((2, 1, 8))
    with pytest.raises(nat.NativeError):
        fr.ISS([fr.words.SimpleWord("[1]")]).fit_transform(X)
    with pytest.raises(nat.NativeError):
        fr.preparation.INC().fit_transform(X)
    with pytest.raises(nat.NativeError):
        fr.sieving.NPI().fit_transform(X[:, 0, :])
    with pytest.raises(nat.NativeError):
        fr.semiring.Reals().iterated_sum_fast(X, np.array([[1]], np.int32), None, None, 1, True)


def test_end_cut_validation_is_host_side():
    """Integer END cuts are validated before anything touches a device (the reference
    raises IndexError from np.take_along_axis, fruits/sieving/segment.py:213-218)."""
    for cut, T, ok in [(-1, 10, True), (10, 10, True), (1, 10, True), (-10, 10, True),
                       (11, 10, False), (-12, 10, True), (0, 10, True), ([-30, -25], 10, False)]:
        sv = fr.sieving.END(cut=cut)
        if ok:
            sv._int_cut_row(T)
        else:
            with pytest.raises(IndexError):
                sv._int_cut_row(T)


def test_no_oracle_import_in_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fruits_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), (dirpath, f)


# ---------------------------------------------------------------- words
def test_simpleword_parse_golden():
    for s, rows in M["parse"].items():
        w = fr.words.SimpleWord(s)
        assert [list(r) for r in w] == rows
        assert str(w) == s and len(w) == len(rows)
        assert w.table().shape == (len(rows), len(rows[0]))
    pm = M["parse_multiply"]
    w = fr.words.SimpleWord(pm["first"])
    assert w._extended_letters == [[-1, 1], [1, -2]]     # reference test_simple.py:62
    w.multiply(pm["second"])
    assert w._extended_letters == pm["exps"] and str(w) == pm["name"]


def test_simpleword_errors_and_alpha():
    for bad in ["", "[1", "1]", "[a]", "[1][]", "[()]", "[1] [2]"]:
        with pytest.raises(ValueError):
            fr.words.SimpleWord(bad)
    w = fr.words.SimpleWord("[1][2]")
    with pytest.raises(NotImplementedError):
        w.multiply(3)
    np.testing.assert_array_equal(w.alpha, np.ones(2, np.float32))
    assert w.alpha.dtype == np.float32
    with pytest.raises(ValueError):
        w.alpha = [1.0]
    w.alpha = [0.5, 2]
    np.testing.assert_array_equal(w.alpha, np.array([0.5, 2], np.float32))
    c = w.copy()
    assert c == w and c is not w and str(c) == "[1][2]"
    assert fr.words.SimpleWord("[12][122]") == fr.words.SimpleWord("[21][212]")


@pytest.mark.parametrize("key", sorted(M["words"]))
def test_of_weight_and_cacheplan_golden(key):
    w, d = map(int, key.split(","))
    ent = M["words"][key]
    words = fr.words.of_weight(w, dim=d)
    assert [str(x) for x in words] == ent["words"]
    assert [[list(r) for r in x] for x in words] == ent["exps"]
    cp = fr.iss.CachePlan(words)
    assert cp._plan == ent["plan"] and cp.n_iterated_sums() == ent["K"]


def test_word_counts():
    # reference tests/signature/test_simple.py:54-57
    for n in range(1, 7):
        assert len(fr.words.of_weight(n, dim=1)) == 2 ** (n - 1)
    assert len(fr.words.of_weight(4, dim=2)) == 82


def test_alternate_sign_golden():
    a = M["alternate_sign"]
    out = fr.words.alternate_sign([fr.words.SimpleWord(s) for s in a["in"]])
    assert [str(w) for w in out] == a["out"]


def test_cacheplan_goldens_and_labels():
    for ent in M["cacheplan"]:
        cp = fr.iss.CachePlan([fr.words.SimpleWord(s) for s in ent["words"]])
        assert cp._plan == ent["plan"]
        assert [cp.get_word_string(i) for i in range(cp.n_iterated_sums())] == ent["labels"]
        with pytest.raises(IndexError):
            cp.get_word_string(cp.n_iterated_sums())
    assert M["cacheplan"][0]["plan"] == [4, 5, 2, 3, 3, 1, 1, 1, 2]   # test_cache.py:26
    cp = fr.iss.CachePlan([fr.words.SimpleWord(s) for s in M["cacheplan"][0]["words"]])
    assert cp.get_word_index(0) == 0 and cp.get_word_index(4) == 1
    assert cp.n_iterated_sums(range(2)) == 9


# ---------------------------------------------------------------- plan compiler (host C++)
def plan_of(words, mode="EXTENDED", share=True, weighting=nat.FR_W_NONE, alphas=None):
    ws = [fr.words.SimpleWord(s) for s in words]
    depths = orc.cache_plan(words) if mode == "EXTENDED" else [1] * len(words)
    if weighting != nat.FR_W_NONE and alphas is None:
        alphas = [w.alpha for w in ws]
    return nat.Plan([w.table() for w in ws], depths, alphas, weighting, share), depths


@pytest.mark.parametrize("key", ["2,3", "4,2", "6,2", "9,1", "3,3"])
def test_plan_is_a_trie_walk(key):
    ent = M["words"][key]
    plan, depths = plan_of(ent["words"])
    K = sum(depths)
    assert plan.rows == K
    # of_weight sets contain every prefix they need exactly once: K scans, not sum(L)
    assert plan.nodes == K
    assert plan.info(nat.FR_INFO_SHARED) == 1
    d = plan.dump()
    assert d.shape == (K, 8)
    # DFS preorder: a node's level is at most one deeper than its predecessor's
    lv = d[:, 0]
    assert lv[0] == 0 and np.all(np.diff(lv) <= 1) and lv.max() + 1 == plan.info(nat.FR_INFO_LEVELS)
    # every output row is written by exactly one node
    assert sorted(d[:, 4].tolist()) == list(range(K)) and np.all(d[:, 3] == 1)
    unshared, _ = plan_of(ent["words"], share=False)
    assert unshared.nodes == sum(len(e) for e in ent["exps"]) and unshared.rows == K
    assert unshared.info(nat.FR_INFO_LEVELS) == 1          # chains run in place


def test_plan_details():
    plan, _ = plan_of(["[11]", "[1][2]", "[1][2][3]", "[2]"])
    d = plan.dump()
    # nodes: [11] | [1] -> [2] -> [3] (chain, in place) | [2]
    assert plan.nodes == 5 and plan.rows == 6 - 1
    assert d[:, 0].tolist() == [0, 0, 0, 0, 0]
    assert (d[:, 1] & 1).tolist() == [0, 0, 1, 1, 0]        # F_CHAIN
    assert (d[:, 1] & 2).tolist() == [0, 2, 2, 0, 0]        # F_CHILDREN
    assert d[:, 2].tolist() == [2, 1, 1, 1, 1]              # factors
    assert plan.max_dim == 3 and plan.dims_used == 3
    assert plan.info(nat.FR_INFO_GROUPS) == 3
    # duplicated words (SINGLE mode): one node, several output rows
    dup, _ = plan_of(["[12]", "[1]", "[12]", "[21]"], mode="SINGLE")
    dd = dup.dump()
    assert dup.rows == 4 and dup.nodes == 2 and sorted(dd[:, 3].tolist()) == [1, 3]
    # weighted: distinct alphas make distinct nodes and exp tables
    ws = [fr.words.SimpleWord(s) for s in ["[1][2]", "[1][3]"]]
    ws[1].alpha = [0.5, 1.0]
    wp = nat.Plan([w.table() for w in ws], [2, 2], [w.alpha for w in ws], nat.FR_W_NONTOTAL)
    assert wp.nodes == 4 and wp.info(nat.FR_INFO_ALPHAS) == 2
    wp2 = nat.Plan([w.table() for w in ws], [2, 1], [np.ones(2, np.float32)] * 2, nat.FR_W_TOTAL)
    assert wp2.nodes == 3 and wp2.info(nat.FR_INFO_ALPHAS) == 1 and wp2.rows == 3
    assert plan.workspace_bytes(100, 1024, 0) == 0
    assert plan.workspace_bytes(100, 4096, 0) >= 100 * 2 * plan.nodes * 8
    assert wp.workspace_bytes(10, 64, 1) >= 2 * 2 * 64 * 8


def test_plan_rejects_bad_input():
    t = np.array([[1]], np.int32)
    with pytest.raises(ValueError):
        nat.Plan([t], [2])                         # depth > L
    with pytest.raises(ValueError):
        nat.Plan([t], [1], None, nat.FR_W_TOTAL)   # weighted without alphas
    with pytest.raises(ValueError):
        nat.Plan([t], [1], [np.ones(3, np.float32)], nat.FR_W_TOTAL)
    with pytest.raises(ValueError):
        nat.Plan([np.array([1], np.int32)], [1])   # not (L, Dw)
    deep = nat.Plan([fr.words.SimpleWord("[1]" * 30 + "[2]").table(),
                     fr.words.SimpleWord("[1]" * 30 + "[3]").table()], [31, 1])
    assert deep.info(nat.FR_INFO_LEVELS) <= 8 and deep.rows == 32


# ---------------------------------------------------------------- stages (no compute)
def test_iss_bookkeeping():
    words = fr.words.of_weight(2, dim=3)
    ext = fr.ISS(words, mode=fr.ISSMode.EXTENDED)
    assert ext.n_iterated_sums() == 18 and not ext.requires_fitting
    assert fr.ISS(words).n_iterated_sums() == 15
    case = [c for c in G.cases("iss") if c["name"] == "w23_ext_U"][0]
    assert [ext.label(i) for i in range(18)] == case["labels"]
    wl = fr.ISS(words, weighting=fr.iss.weighting.Indices())
    assert wl.label(0) == "[11] : Indices"
    cp = ext.copy()
    assert cp is not ext and cp.words is ext.words and cp.mode == ext.mode
    with pytest.raises(ValueError):
        next(fr.ISS(words).batch_transform(np.zeros((1, 3, 4)), batch_size=16))
    with pytest.raises(TypeError):
        fr.ISS(words).fit_transform(np.zeros((1, 3, 4), dtype=np.float32))
    fr.ISS(words, semiring=fr.semiring.Arctic())._check_supported()
    fr.ISS(words, semiring=fr.semiring.Bayesian())._check_supported()
    assert fr.ISS(words, semiring=fr.semiring.Bayesian()).label(0) == "[11] : Bayesian"
    am = fr.ISS(words, semiring=fr.semiring.Arctic(argmax=True), mode=fr.ISSMode.EXTENDED)
    am._check_supported()
    # fruits/iss/iss.py:140-145: L + L(L+1)/2 rows per word
    assert am.n_iterated_sums() == sum(len(w) + len(w) * (len(w) + 1) // 2 for w in words)
    with pytest.raises(NotImplementedError):   # iss.py:37-40,146-149
        fr.ISS(words, semiring=fr.semiring.Arctic(argmax=True)).n_iterated_sums()
    with pytest.raises(NotImplementedError):
        fr.ISS(words, semiring=fr.semiring.Arctic(argmax=True))._check_supported()
    assert fr.ISS(words, semiring=fr.semiring.Arctic()).label(0) == "[11] : Arctic"
    assert ext.word_batches(4, 8, 1) == [(i, i + 1) for i in range(15)]
    assert ext.word_batches(4, 8) == [(0, 15)]


def test_weighting_rows_match_oracle():
    for kw in [{}, {"scale": 2.0}, {"relative": False, "scale": 3.0}]:
        row = fr.iss.weighting.Indices(**kw)._row(37)
        ref = orc.lookup_indices(1, 37, kw.get("relative", True), kw.get("scale", 50))[0]
        np.testing.assert_array_equal(np.asarray(row, dtype=np.float64), ref)
    with pytest.raises(ValueError):
        fr.iss.weighting.Plateaus(1)
    p = fr.iss.weighting.Plateaus(4, scale=1.0)._row(16)
    assert p[0] == 0 and p[-1] == 1 and len(set(np.round(p, 12))) == 4
    for case in M.get("lookups", []):
        if case["kind"] == "Plateaus":         # pinned against the reference's staircases
            row = fr.iss.weighting.Plateaus(**case["kw"])._row(G[case["x"]].shape[2])
            np.testing.assert_allclose(row, G[case["out"]][0], rtol=1e-15, atol=0)


def test_preparateurs_and_sieves_bookkeeping():
    with pytest.raises(ValueError):
        fr.preparation.INC(depth=0)
    inc = fr.preparation.INC(2, 3, False)
    assert str(inc) == "INC(2, 3, False)" and inc == inc.copy() and not inc.requires_fitting
    assert inc._lag(100) == 2 and fr.preparation.INC(0.1)._lag(95) == 10
    assert fr.preparation.INC(lambda T: T // 4)._lag(100) == 25
    assert str(fr.preparation.NEW(fr.preparation.INC())) == "NEW(INC(1, 1, True))"
    assert str(fr.preparation.STD()) == "STD(True, True)"
    npi = fr.sieving.NPI(q=(0.5, 1.0), cut=[10, -1])
    assert npi.requires_fitting and npi.nfeatures() == 2
    assert npi.label(1) == "NPI[inc=1]!-1![0.5, 1.0]"
    assert str(npi) == "NPI([10, -1], (0.5, 1.0), 1)"
    assert not fr.sieving.NPI().requires_fitting and not fr.sieving.END().requires_fitting
    assert fr.sieving.END().label(0) == "END!-1![-1.0, 1.0]"
    np.testing.assert_array_equal(fr.sieving.NPI(cut=[-1, 3, 1])._int_cut_row(5), [0, 1, 3, 5])
    cuts = fr.sieving.END(cut=[1, 4, -1])._get_transformed_cuts(np.zeros((2, 5)))
    np.testing.assert_array_equal(cuts, orc.transformed_cuts(2, 5, [1, 4, -1]))
    with pytest.raises(RuntimeError):
        fr.sieving.NPI(q=(0.3, 1.0))._get_unfitted_quantiles()
    s = fr.sieving.END(q=(-1.0, 0.25, 0.0, 1.0))   # SegmentSieve._fit: host np.quantile
    s._fit(np.arange(8.0).reshape(2, 4))
    np.testing.assert_array_equal(s._quantiles, orc.fit_quantiles((-1.0, 0.25, 0.0, 1.0),
                                                                   np.arange(8.0)))


def build_fruit(spec):
    from test_hip_parity import build_fruit as bf
    return bf(fr, spec)


@pytest.mark.parametrize("case", G.cases("fruit"), ids=lambda c: c["name"])
def test_fruit_structure_golden(case):
    fruit = build_fruit(case["spec"])
    assert fruit.nfeatures() == case["nfeatures"]
    assert [fruit.label(i) for i in range(fruit.nfeatures())] == case["labels"]
    assert [fruit.label(i, verbose=2) for i in range(len(case["labels_v2"]))] == case["labels_v2"]
    assert fruit.summary() == case["summary"]
    with pytest.raises(RuntimeError, match="Missing call of self.fit"):
        fruit.transform(G[case["x"]])
    dc = fruit.deepcopy()
    assert dc.nfeatures() == fruit.nfeatures() and len(dc) == len(fruit)
    assert dc.name.endswith("(Deepcopy)") and fruit.copy().name.endswith("(Copy)")


def test_fruit_api_errors():
    fruit = fr.Fruit("x")
    with pytest.raises(TypeError):
        fruit.add(3)
    fruit.add(fr.preparation.INC)          # classes are instantiated
    assert isinstance(fruit.get_slice().get_preparateurs()[0], fr.preparation.INC)
    with pytest.raises(RuntimeError, match="No ISS given"):
        fruit.fit(np.zeros((2, 1, 8)))
    fruit.add(fr.ISS(fr.words.of_weight(4, dim=2)))
    with pytest.raises(RuntimeError, match="No feature sieves given"):
        fruit.fit(np.zeros((2, 2, 8)))
    fruit.add(fr.sieving.NPI, fr.sieving.END)
    assert fruit.nfeatures() == 164
    with pytest.raises(TypeError):
        fruit.fit(np.zeros((2, 2, 8), dtype=np.float32))
    with pytest.raises(IndexError):
        fruit.switch_slice(3)
    fruit.cut()
    assert len(fruit) == 2 and fruit[1] is fruit.get_slice()
    assert [s for s in fruit] == [fruit[0], fruit[1]]
    fruit[0].clear()
    assert fruit[0].nfeatures() == 0 and fruit[0].fit_sample_size == 1
    # reference tests/core/test_fruit.py:11-28 (counts only; PPV/MAX/MIN are out of scope)
    f2 = fr.Fruit()
    f2.add(fr.preparation.INC(zero_padding=False))
    f2.add(fr.ISS(fr.words.of_weight(4, dim=2)))
    assert len(f2.get_slice().get_iss()[0].words) == 82


@pytest.mark.parametrize("case", G.manifest.get("coswiss", []), ids=lambda c: c["name"])
def test_coswiss_host_tables(case):
    # expansion table, labels and term programs of CosWISS (fruits/iss/cos.py:230-287,345-351)
    cw = fr.CosWISS([fr.words.SimpleWord(s) for s in case["words"]], case["freqs"], **case["kw"])
    assert cw.n_iterated_sums() == len(case["labels"])
    assert [cw.label(i) for i in range(cw.n_iterated_sums())] == case["labels"]
    assert not cw.requires_fitting
    for s, w in case["weightings"].items():
        assert cw._get_weightings(fr.words.SimpleWord(s)).tolist() == w
    T = G[case["x"]].shape[2]
    trig = cw._trig(T)
    for f, freq in enumerate(case["freqs"]):
        sn, cs = orc.coswiss_trig(T, freq)
        np.testing.assert_array_equal(trig[f, 0], sn)
        np.testing.assert_array_equal(trig[f, 1], cs)
    c2 = cw._copy()
    assert c2._freqs == cw._freqs and c2._exponent == cw._exponent


def test_plan_compiler_sanitized(tmp_path):
    """The host plan compiler under AddressSanitizer + UBSan (CPU build; the GPU pool has
    no sanitizer runs): word lists of the BASELINE configs, long chains, duplicated and
    invalid words, weighted / Arctic / Bayesian / unshared plans, CosWISS programs."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "plan_sanitize")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", os.path.join(ROOT, "tests", "native", "plan_sanitize.cpp"),
           os.path.join(ROOT, "fruits_amd", "csrc", "plan.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr

    def case(words, mode="EXTENDED", weighting=0, flags=1, kind=0, depths=None, alphas=None):
        ws = [fr.words.SimpleWord(s) for s in words]
        if depths is None:
            depths = orc.cache_plan(words) if mode == "EXTENDED" else [1] * len(words)
        lines = [f"{kind} {len(ws)} {weighting} {flags}"]
        for i, w in enumerate(ws):
            t = w.table()
            a = alphas[i] if alphas else [1.0] * t.shape[0]
            lines.append(f"{t.shape[0]} {t.shape[1]} {depths[i]} "
                         + " ".join(str(int(v)) for v in t.ravel()) + " "
                         + " ".join(str(float(v)) for v in a))
        return "\n".join(lines)

    alt = [str(w) for w in fr.words.alternate_sign([fr.words.SimpleWord(48 * "[1]"),
                                                    fr.words.SimpleWord(24 * "[1][2]")])]
    cases = [
        case(M["words"]["2,3"]["words"]),
        case(M["words"]["4,2"]["words"], weighting=1),
        case(M["words"]["6,2"]["words"], weighting=2),
        case(M["words"]["9,1"]["words"], weighting=1, flags=0),
        case(alt, flags=1 | 2),                                  # Arctic, 48 levels -> unshared
        case(M["words"]["3,2"]["words"], flags=1 | 4, weighting=2),   # Bayesian
        case(["[12]", "[1]", "[12]", "[21]", "[1][-2]"], mode="SINGLE"),
        case(["[1][2]", "[1][3]"], weighting=1, alphas=[[1.0, 1.0], [0.5, 1.0]]),
        case(["[1][2][3]"], depths=[0]),                         # nothing to output
        case(["[1][2]"], depths=[5]),                            # invalid depth: rejected
        case(M["words"]["3,2"]["words"], kind=1, flags=1),       # CosWISS program
    ]
    text = f"{len(cases)}\n" + "\n".join(cases) + "\n"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], input=text, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "checksum" in r.stdout and "rejected" in r.stdout
    assert "K=18 nodes=18" in r.stdout.splitlines()[0]


def test_numpy_sum_model(tmp_path):
    """STD's statistics follow numpy's summation order (csrc/pairwise.h): the order logic of the
    device routine, compiled for the host (tests/native/pairwise_host.cpp, the lanes' shuffles
    emulated), equals np.add.reduce bit for bit - lengths around every boundary of the
    recursion (8, 128, the halving to multiples of 8) and of the 8192-element buffers."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    so = str(tmp_path / "pairwise_host.so")
    subprocess.check_call([gxx, "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "native", "pairwise_host.cpp"), "-o", so])
    L = C.CDLL(so)
    L.pw_host_sum.restype = C.c_double
    L.pw_host_sum.argtypes = [C.c_void_p, C.c_int64]
    rng = np.random.default_rng(5)
    lengths = (list(range(1, 300)) + [511, 512, 1000, 1023, 1024, 1025, 2047, 4096, 4097, 5000,
               8191, 8192, 8193, 8199, 8200, 12345, 16384, 16385, 20000, 70001, 200000])
    for T in lengths:
        a = np.ascontiguousarray(rng.standard_normal(T) * np.exp(rng.uniform(-5, 5, T)))
        got = L.pw_host_sum(a.ctypes.data, T)
        assert got == np.add.reduce(a), (T, got, np.add.reduce(a))
        X = a.reshape(1, 1, T)
        assert got / T == np.mean(X, axis=2)[0, 0]
    z = np.full(9, -0.0)
    assert np.signbit(L.pw_host_sum(z.ctypes.data, 9)) == np.signbit(np.add.reduce(z))


@pytest.mark.parametrize("key,piece,types", [("6,2", 64, 5), ("6,2", 128, 4), ("9,1", 64, 5), ("4,2", 16, None),
                                             ("3,3", 8, None)])
def test_plan_in_pieces_is_a_cover(key, piece, types):
    """A large plan in pieces (csrc/plan.h PiecedProgram, what the fused walk of config 4 / 5
    runs): walking every unit of every piece type - its items' chains, then their bodies - emits
    every output row exactly once, in the walk order the tables claim; equal sub-tries are ONE
    type (of_weight(6,2): 1351 nodes, five bodies of at most 62 nodes to compile), the recomputed
    chains cost a few per cent of the node executions."""
    ent = M["words"][key]
    words = [fr.words.SimpleWord(s) for s in ent["words"]]
    for kw in ({}, {"weighting": fr.iss.weighting.Indices()}):
        plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED, **kw)._plan(0, len(words))
        pc = plan.pieces(piece)
        assert pc is not None and pc["K"] == ent["K"] and pc["nodes"] == plan.nodes
        if types is not None:
            assert len(pc["types"]) == types
        seen = np.zeros(pc["K"], dtype=int)
        execs = 0
        for t in pc["types"]:
            recs, nb = t["recs"], t["body_nodes"]
            assert nb <= piece and (recs[nb, 0] & 0xff) == 0xff            # (sentinel behind the body)
            assert int(recs[:nb, 6].sum()) == t["body_rows"] and t["levels"] <= 8
            assert (recs[:nb, 0] & 0xff).max() < t["levels"]
            for u in range(t["units"]):
                q, nodes = int(t["unit_row0"][u]), 0
                for chain_off, row_base, node_base, _ in t["items"][t["unit_begin"][u]:t["unit_begin"][u + 1]]:
                    assert node_base == nodes and chain_off % 64 == 0
                    r = chain_off // 64
                    while (recs[r, 0] & 0xff) != 0xff:                      # the chain: frame 0, in place
                        assert (recs[r, 0] & 0xff) == 0 and (recs[r, 0] >> 8) & 1
                        ne = int(recs[r, 6])
                        assert ne == 0 or recs[r, 7] == q
                        seen[pc["row_of_walk"][q:q + ne]] += 1
                        q, r, nodes = q + ne, r + 1, nodes + 1
                    assert row_base == q
                    seen[pc["row_of_walk"][q:q + t["body_rows"]]] += 1
                    q, nodes = q + t["body_rows"], nodes + nb
                assert nodes <= t["max_unit_nodes"] and q - t["unit_row0"][u] <= t["max_unit_rows"]
                execs += nodes
        assert (seen == 1).all()
        assert execs == pc["nodes"] + pc["chain_nodes"] - sum(
            1 for t in pc["types"] for r in t["recs"][t["body_nodes"] + 1:] if (r[0] & 0xff) != 0xff and r[6] > 0)
        if plan.nodes > 500:
            assert pc["chain_nodes"] <= 0.06 * plan.nodes


def test_static_schedules_are_valid_walks():
    """plan.cpp static_schedule: every node once, letters only on completed rows, children
    behind their parents - for the standard word sets and 1-3 groups per series."""
    for w, d in ((1, 3), (2, 2), (2, 3), (3, 1)):
        for mode in (fr.ISSMode.EXTENDED, fr.ISSMode.SINGLE):
            words = fr.words.of_weight(w, dim=d)
            plan = fr.ISS(words, mode=mode)._plan(0, len(words))
            for G in (1, 2, 3):
                got = plan.static_schedule(G)
                assert got is not None
                head, sched = got
                recs = plan.records(1)
                _check_schedule_with_children(recs, head, sched)
    # repeated words (SINGLE mode): nodes with more than the two output rows a record holds
    w15 = fr.words.of_weight(2, dim=3)
    w48 = [w15[i % 15] for i in range(48)]
    plan = fr.ISS(w48)._plan(0, 48)
    for G in (1, 3):
        head, sched = plan.static_schedule(G)
        assert any((int(e[0]) & 0xff) == 0xfc for e in sched)
        _check_schedule_with_children(plan.records(1), head, sched)
    # plans that do not qualify: weighted, too many nodes, Arctic letter sums
    big = fr.words.of_weight(4, dim=2)
    assert fr.ISS(big, mode=fr.ISSMode.EXTENDED)._plan(0, len(big)).static_schedule(1) is None
    wd = fr.ISS(fr.words.of_weight(2, dim=2), weighting=fr.iss.weighting.Indices())
    assert wd._plan(0, 7).static_schedule(1) is None


def _check_schedule_with_children(recs, head, sched):
    # parents from the interpreter records (level / chain structure, as walk() reads them)
    parent, last_at = {}, {}
    for r in recs[:-1]:
        lv, fl, nid = int(r[0]) & 0xff, int(r[0]) >> 8, int(r[9])
        parent[nid] = last_at.get(lv) if fl & 1 else (last_at.get(lv - 1) if lv > 0 else None)
        last_at[lv] = nid
    children = {}
    for c, p in parent.items():
        if p is not None:
            children[p] = children.get(p, 0) + 1
    seen, emitted = [], []
    for g in range(head["groups"]):
        staged, frame_of, left = set(), {}, {}
        i = head["group_begin"][g]
        while True:
            e = sched[i]
            i += 1
            kind = int(e[0]) & 0xff
            if kind == 0xff:
                break
            if kind == 0xfe:
                staged.add(int(e[1]))
                continue
            if kind == 0xfd:
                assert head["groups"] == 1
                continue
            if kind == 0xfc:     # further output rows of the node entry before it
                assert seen and 1 <= int(e[1]) <= 14
                emitted[-1] += int(e[1])
                continue
            nid, fin, fout = int(e[9]), int(e[14]), int(e[15])
            emitted.append(int(e[6]))
            for j in range(int(e[1])):
                assert (int(e[2 + j]) & 0x7f) in staged, "letter on a row that is still in flight"
            p = parent[nid]
            if p is None:
                assert fin == -1
            else:
                assert frame_of.get(p) == fin, "reads the frame its parent wrote"
                left[p] -= 1
                if left[p] == 0:
                    del frame_of[p]
            if children.get(nid, 0) > 0:
                assert 0 <= fout < head["frames"] and fout not in frame_of.values()
                frame_of[nid] = fout
                left[nid] = children[nid]
            else:
                assert fout == -1
            seen.append(nid)
        assert not frame_of, "a frame was left open"
    assert sorted(seen) == sorted(parent), "every node exactly once"
    total = {int(r[9]): int(r[6]) for r in recs[:-1]}
    assert sorted(emitted) == sorted(total.values()), "every output row of every node"


def test_static_program_header_is_current():
    """csrc/static_programs.h is generated data: regenerate and compare."""
    from fruits_amd import gen_static
    with open(gen_static.HEADER) as f:
        assert f.read() == gen_static.render(), "run `python -m fruits_amd.gen_static` and rebuild"


def test_jit_compiles_static_programs(tmp_path, monkeypatch):
    """jit.cpp: the static program of a plan outside the standard word sets compiles with
    hipRTC (no GPU needed) and is cached on disk; plans without a schedule compile nothing."""
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(tmp_path / "jit"))
    words = [fr.words.SimpleWord(s) for s in ["[1][2]", "[12][1]", "[2]", "[1][1][2]", "[3][1]", "[33]"]]
    plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
    assert plan.static_schedule(1) is not None
    try:
        size, msg = plan.jit(1, compile_only=True)
    except ValueError as e:
        if "not available" in str(e):
            pytest.skip("hipRTC is not installed")
        raise
    assert size > 4096 and msg == ""
    files = os.listdir(tmp_path / "jit")
    assert len(files) == 1 and files[0].endswith(".gfx950.co")
    assert plan.jit(1, compile_only=True)[0] == size          # served from the cache
    assert plan.jit(3, compile_only=True)[0] > 4096 and len(os.listdir(tmp_path / "jit")) == 2
    big = fr.words.of_weight(4, dim=2)
    assert fr.ISS(big, mode=fr.ISSMode.EXTENDED)._plan(0, len(big)).jit(1, compile_only=True)[0] == 0


def test_jit_cache_is_validated(tmp_path, monkeypatch):
    """The disk cache of compiled programs: a damaged or foreign file is not trusted (header with
    sizes and a hash of the payload) but compiled again and replaced; a cache directory that
    cannot be created or is open to others is simply not used."""
    words = [fr.words.SimpleWord(s) for s in ["[1][2]", "[12][1]", "[2]", "[1][1][2]"]]
    plan = fr.ISS(words, mode=fr.ISSMode.EXTENDED)._plan(0, len(words))
    cache = tmp_path / "jit"
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(cache))
    try:
        size, _ = plan.jit(1, compile_only=True)
    except ValueError as e:
        if "not available" in str(e):
            pytest.skip("hipRTC is not installed")
        raise
    (name,) = os.listdir(cache)
    good = (cache / name).read_bytes()
    assert good[:7] == b"FRJITCO" and len(good) > size
    assert (os.stat(cache).st_mode & 0o077) == 0                # private directory
    for damaged in (good[:len(good) // 2], good[:-1] + bytes([good[-1] ^ 1]), b"", b"x" * 4096):
        (cache / name).write_bytes(damaged)
        assert plan.jit(1, compile_only=True)[0] == size        # rejected, compiled again
        assert (cache / name).read_bytes() == good              # and replaced
    # a directory that cannot exist, and one others may write to: compile without a cache
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", "/dev/null/jit")
    assert plan.jit(1, compile_only=True)[0] == size
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", str(shared))
    assert plan.jit(1, compile_only=True)[0] == size
    assert os.listdir(shared) == []


def test_jit_without_hiprtc_falls_back(tmp_path):
    """No hipRTC: fr_plan_jit says so (FR_E_LIMIT) and nothing else changes - the plan keeps
    its interpreter program (checked in a fresh process: the library handle is cached)."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import fruits_amd as fr\n"
        "ws = [fr.words.SimpleWord(s) for s in ['[1][2]', '[12][1]', '[2]']]\n"
        "plan = fr.ISS(ws, mode=fr.ISSMode.EXTENDED)._plan(0, 3)\n"
        "try:\n"
        "    plan.jit(1, compile_only=True)\n"
        "    print('compiled')\n"
        "except ValueError as e:\n"
        "    print('refused:', 'not available' in str(e))\n"
        "print(len(plan.records(1)))\n" % ROOT)
    env = dict(os.environ, FRUITS_HIP_RTC_LIB="/nonexistent/libhiprtc.so",
               FRUITS_HIP_JIT_CACHE=str(tmp_path / "jit"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert "refused: True" in r.stdout and r.stdout.strip().endswith("6")   # 5 nodes + sentinel


def test_static_program_identity_includes_dimensions():
    """The records of a plan name LDS rows, not input dimensions: `[4]` alone compiles to the
    records of `[1]` alone.  Only a plan whose staged rows come from the SAME dimensions may
    run a pre-compiled program (found by the soak of the run-time compiled programs)."""
    def index(words, mode="EXTENDED"):
        ws = [fr.words.SimpleWord(s) for s in words]
        return fr.ISS(ws, mode=getattr(fr.ISSMode, mode))._plan(0, len(ws)).static_program_index()
    assert index(["[1]"]) > 0
    assert index(["[4]"]) == 0 and index(["[2]"]) == 0
    assert index(["[1]", "[2]"]) > 0 and index(["[2]", "[3]"]) == 0
    w22 = [str(w) for w in fr.words.of_weight(2, dim=2)]
    assert index(w22) > 0
    assert index([s.replace("2", "3") for s in w22]) == 0          # dimensions (1, 3)
    assert index([s.replace("2", "3").replace("1", "2") for s in w22]) == 0   # dimensions (2, 3)


def test_bundle_compiles_without_a_device(tmp_path, monkeypatch):
    """fr_pipeline_bundle (what fruits_amd/gen_bundle.py builds fruits_amd/jit_bundle with): the
    kernels a pipeline would compile at run time - the sieves as immediates, and the plan as
    straight-line code or in pieces - compiled into a directory WITHOUT a GPU; a second call finds
    them there; placeholder thresholds carry the infinities only."""
    from fruits_amd import gen_bundle as gb
    from fruits_amd.fruit import FruitSlice
    monkeypatch.setenv("FRUITS_HIP_JIT_CACHE", "")
    out = tmp_path / "bundle"
    out.mkdir(mode=0o700)
    S = fr.sieving
    cases = [
        # 9 nodes: straight-line plan; 33 nodes with pieces of <= 12: a plan in pieces
        (fr.words.of_weight(2, 2), {}, 2),
        (fr.words.of_weight(3, 2), {"FRUITS_HIP_DEBUG": "piece_min=20,piece_nodes=12"}, None),
    ]
    for words, env, want in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        iss = fr.ISS(words, mode=fr.ISSMode.EXTENDED, weighting=fr.iss.weighting.Indices())
        plan = iss._plan(0, len(words))
        sieves = [S.NPI(q=(0.5, 1.0)), S.END()]
        specs, _, _ = FruitSlice._pipeline_specs(sieves, 1024)
        pipe = nat.Pipeline(plan, specs, 1024)
        quant = gb.placeholder_quantiles(sieves, plan.rows, pipe.q_stride)
        assert np.isinf(quant[:, 1]).all() and np.isfinite(quant[:, 0]).all()
        try:
            n = pipe.bundle(quant, str(out))
        except nat.NativeError as e:
            if "not available" in str(e):
                pytest.skip("hipRTC is not installed")
            raise
        files = sorted(os.listdir(out))
        if want is not None:
            assert n == want and len(files) == want
        else:
            cover = plan.pieces(12)
            assert n == 1 + len(cover["types"]) and len(files) == 2 + n
        assert all(f.endswith(".gfx950.co") for f in files)
        stamp = {f: os.path.getmtime(out / f) for f in files}
        assert pipe.bundle(quant, str(out)) == n and sorted(os.listdir(out)) == files   # found, not rebuilt
        assert all(os.path.getsize(out / f) > 4096 for f in files)


def test_fitted_rows_table():
    """What a device-side fit leaves in FruitSlice._sieves_extended (fruit.py, _FittedRows): the
    thresholds of all rows as arrays, formed like SegmentSieve._set_quantiles_from_stats forms one
    row's (numpy's _lerp, both ends; NaN-last sort), the list-of-lists of fruits/fruit.py:462-476
    made on demand - changed copies are seen by later reads and survive pickling."""
    import pickle
    from fruits_amd.fruit import _FittedRows, _interpolated_quantiles
    rng = np.random.default_rng(0)
    npi = fr.sieving.NPI(q=(0.25, 0.7, 1.0), inc=1)
    mpi = fr.sieving.MPI(q=(-1.0, 0.5))
    end = fr.sieving.END()
    n = 1001
    rows = 7
    tables = []
    for sv in (npi, mpi):
        reqs = sv._quantile_requests(n)
        lo = rng.standard_normal((rows, len(reqs)))
        hi = lo + rng.random((rows, len(reqs)))
        lo[2, 0] = np.nan
        got = _interpolated_quantiles(sv, reqs, lo, hi)
        for k in range(rows):
            one = sv.copy()
            one._set_quantiles_from_stats(reqs, lo[k].tolist(), hi[k].tolist())
            np.testing.assert_array_equal(got[k], one._quantiles)
        tables.append(got)
    ext = _FittedRows([npi, mpi, end], cache=None)
    first = ext.add_rows(3)
    second = ext.add_rows(rows - 3)
    for i in range(2):
        first[i], second[i] = tables[i][:3], tables[i][3:]
    assert len(ext) == rows and bool(ext) and not _FittedRows([npi], None)
    assert ext.thresholds(2, range(rows)) is None          # END is not fitted on data
    np.testing.assert_array_equal(ext.thresholds(0, [5, 1]), tables[0][[5, 1]])
    row = ext[-1]
    assert [type(s) for s in row] == [fr.sieving.NPI, fr.sieving.MPI, fr.sieving.END] and ext[rows - 1] is row
    np.testing.assert_array_equal(row[1]._quantiles, tables[1][-1])
    assert row[0]._inc == 1 and len(list(ext)) == rows and len(ext[1:4]) == 3
    with pytest.raises(IndexError):
        ext[rows]
    row[0]._quantiles = np.array([1.0, 2.0, 3.0])       # (what transplant_thresholds does)
    np.testing.assert_array_equal(ext.thresholds(0, [rows - 1, 0]), [[1.0, 2.0, 3.0], tables[0][0]])
    back = pickle.loads(pickle.dumps(ext))
    assert len(back) == rows and back._cache is None
    np.testing.assert_array_equal(back[rows - 1][0]._quantiles, [1.0, 2.0, 3.0])
    np.testing.assert_array_equal(back.thresholds(1, range(rows)), tables[1])
