"""CPU oracle: a numpy restatement of the reference's ISS hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fruits_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.  Parity is PINNED: every
function here is checked by ``tests/test_oracle.py`` against golden vectors
that ``tests/golden/make_golden.py`` produced by running the reference itself
(irkri/fruits @ 2025-09-05), and against the hand-computed vectors held by the
reference's own tests.

Each function cites the reference file:line it restates (paths relative to the
reference root).  The arithmetic is the reference's: same multiply / divide
order per letter, strictly sequential ``cumsum`` along time, the same shifts.
The only liberty is vectorisation over the series axis N (the reference loops
``prange`` over N), which changes no floating-point result.
"""
from __future__ import annotations

import itertools
import re

import numpy as np

# --------------------------------------------------------------------------
# words
# --------------------------------------------------------------------------

_WORD_RE = re.compile(r"(\[(-?\d|\(-?\d+\))+\])+")


def parse_word(string: str) -> list[list[int]]:
    """fruits/iss/words/word.py:189-245 (SimpleWord.__init__/multiply).

    ``"[-12][(10)1]"`` -> one row per extended letter; entry d = signed number
    of occurrences of dimension d+1 (negative = reciprocal power).
    """
    if not _WORD_RE.fullmatch(string):
        raise ValueError("not a simple word: %r" % (string,))
    rows_raw = []
    for el in string.split("]")[:-1]:
        el = el[1:]
        letters = []
        j = 0
        while j < len(el):
            if el[j] == "(":
                k = el.index(")", j)
                letters.append(int(el[j + 1:k]))
                j = k
            elif el[j] == "-":
                letters.append(int(el[j:j + 2]))
                j += 1
            else:
                letters.append(int(el[j]))
            j += 1
        rows_raw.append(letters)
    max_dim = max(abs(x) for r in rows_raw for x in r)
    rows = []
    for letters in rows_raw:
        row = [0] * max_dim
        for x in set(letters):
            c = letters.count(x)
            row[abs(x) - 1] += c if x > 0 else -c
        rows.append(row)
    return rows


def _partitions_of(n, start=1):
    # fruits/iss/words/creation.py:8-12
    yield (n,)
    for i in range(start, n // 2 + 1):
        for p in _partitions_of(n - i, i):
            yield (i,) + p


def of_weight_strings(w: int, dim: int = 1, perm_order=None) -> list[str]:
    """fruits/iss/words/creation.py:26-50.

    The reference iterates ``set(itertools.permutations(partition))`` whose
    order is an implementation detail of CPython's set; the golden manifest
    pins the order the reference produced, and the tests compare against it.
    """
    letters = []
    for i in range(1, w + 1):
        letters.append([
            "[" + "".join("(%d)" % x if x > 9 else str(x) for x in el) + "]"
            for el in itertools.combinations_with_replacement(
                range(1, dim + 1), i)
        ])
    out = []
    for partition in _partitions_of(w):
        for mixed in set(itertools.permutations(partition)):
            for raw in itertools.product(*[letters[k - 1] for k in mixed]):
                out.append("".join(raw))
    return out


def cache_plan(word_strings: list[str]) -> list[int]:
    """fruits/iss/cache.py:17-37 (CachePlan._create_plan)."""
    plan = []
    for i, wstr in enumerate(word_strings):
        els = wstr.split("[")[1:]
        start = 0
        depth = len(els)
        for j in range(len(els)):
            prefix = "[" + "[".join(els[:j + 1])
            for k in range(start, i):
                if word_strings[k].startswith(prefix):
                    start = k
                    depth -= 1
                    break
            else:
                break
        plan.append(depth)
    return plan


def plan_labels(word_strings: list[str], plan: list[int]) -> list[str]:
    """fruits/iss/cache.py:55-65 (CachePlan.get_word_string) for every index."""
    labels = []
    for wstr, depth in zip(word_strings, plan):
        parts = wstr.split("]")
        n = len(parts) - 1
        for e in range(depth):
            # is_index runs -depth .. -1 ; slice [:is_index] of the split
            labels.append("]".join(parts[:e - depth]) + "]")
        del n
    return labels


# --------------------------------------------------------------------------
# preparateurs
# --------------------------------------------------------------------------

def increments(X: np.ndarray, k: int) -> np.ndarray:
    """fruits/cache.py:8-13 (_increments): out[..., k:] = X[..., k:] - X[..., :-k]."""
    out = np.zeros(X.shape)
    out[:, :, k:] = X[:, :, k:] - X[:, :, :-k]
    return out


def inc_transform(X, shift=1, depth=1, zero_padding=True):
    """fruits/preparation/transform.py:60-75 (INC._transform)."""
    if isinstance(shift, int):
        k = shift
    elif isinstance(shift, float):
        k = int(np.ceil(shift * X.shape[2]))
    else:
        k = int(shift(X.shape[2]))
    out = X
    for _ in range(depth):
        out = increments(out, k)
        if not zero_padding:
            out[:, :, :shift] = X[:, :, :shift]
    return out


def std_transform(X, var=True, eps=1e-5):
    """fruits/preparation/transform.py:141-147 (STD._transform, separately=True)."""
    mean_ = np.mean(X, axis=2)[:, :, None]
    std_ = np.ones((X.shape[0], X.shape[1], 1))
    if var:
        std_ = np.std(X, axis=2)[:, :, None]
    return (X - mean_) / (std_ + eps)


def new_transform(X, inner):
    """fruits/preparation/wrapper.py:78-93 (NEW._transform)."""
    t = X if inner is None else inner(X)
    out = np.zeros((X.shape[0], X.shape[1] + t.shape[1], X.shape[2]))
    out[:, :X.shape[1]] = X
    out[:, X.shape[1]:] = t
    return out


def nrm_rows(R: np.ndarray) -> np.ndarray:
    """fruits/preparation/transform.py:184-198 (NRM._transform, scale_dim=False)
    applied to (N, T) rows: (x-min)/(max-min), constant rows -> 0."""
    mn = R.min(axis=1, keepdims=True)
    mx = R.max(axis=1, keepdims=True)
    # zeros_like keeps R's dtype: for the integer range of Indices(relative=False)
    # the reference therefore truncates to 0/1 (transform.py:195-196) - kept.
    out = np.zeros_like(R)
    mask = (mn != mx)[:, 0]
    out[mask] = (R[mask] - mn[mask]) / (mx[mask] - mn[mask])
    return out


# --------------------------------------------------------------------------
# weighting lookups
# --------------------------------------------------------------------------

def l1_sum(X: np.ndarray) -> np.ndarray:
    """fruits/cache.py:25-31 (_L1_sum): cumsum |increments| of DIMENSION 0."""
    Y = increments(X, 1)[:, 0, :]
    return np.cumsum(np.abs(Y), axis=1)


def l2_sum(X: np.ndarray) -> np.ndarray:
    """fruits/cache.py:34-40 (_L2_sum): cumsum of squared increments of DIMENSION 0."""
    Y = increments(X, 1)[:, 0, :]
    return np.cumsum(Y * Y, axis=1)


def lookup_indices(N, T, relative=True, scale=50.0):
    """fruits/iss/weighting.py:100-110 (Indices.get_lookup, no transform)."""
    r = np.arange(1, T + 1)
    if relative:
        r = r / T
    r = nrm_rows(np.asarray(r)[None, :])[0] * scale
    return np.ones((N, T)) * r


def lookup_l1(X_for_l1, relative=False, scale=50.0):
    """fruits/iss/weighting.py:148-160 (L1.get_lookup, no transform).

    ``X_for_l1`` is the RAW fruit input unless on_prepared=True
    (fruits/cache.py:108-112 via SharedSeedCache)."""
    r = l1_sum(X_for_l1)
    if relative:
        r = r / (r[:, -1:] + 1e-5)
    return nrm_rows(r) * scale


# --------------------------------------------------------------------------
# the ISS operator
# --------------------------------------------------------------------------

def _letters(tmp, Z, el):
    # fruits/iss/semiring.py:111-117 / :143-149 : repeated multiply / divide
    for d, occ in enumerate(el):
        if occ > 0:
            for _ in range(occ):
                tmp = tmp * Z[:, d, :]
        elif occ < 0:
            for _ in range(-occ):
                tmp = tmp / Z[:, d, :]
    return tmp


def _shift(tmp):
    # np.roll(tmp, 1); tmp[0] = 0   (semiring.py:109-110, :155-156)
    out = np.empty_like(tmp)
    out[:, 1:] = tmp[:, :-1]
    out[:, 0] = 0
    return out


def iterated_sum_fast(Z, word, alpha, lookup, extended, total_weighting):
    """fruits/iss/semiring.py:167-201 (Reals._iterated_sum_fast) with its two
    per-series bodies ``_total_weighted_reals_single`` (:128-158) and
    ``_reals_single`` (:93-125), vectorised over N.

    Z (N,D,T) f8, word (L,Dw) i4, alpha (L,) f4, lookup (N,T) f8 -> (N,E,T)."""
    N, _, T = Z.shape
    L = len(word)
    word = np.asarray(word, dtype=np.int32).reshape(L, -1)
    alpha = np.asarray(alpha, dtype=np.float32)
    out = np.zeros((N, extended, T))
    tmp = np.ones((N, T))
    # the reference indexes lookup[j] for j < N (semiring.py:185-199): a lookup
    # with more rows than Z (fit on a sub-sample, see fruit_fit) is cut by position
    lookup = lookup[:N]
    if total_weighting:
        for k in range(L):
            tmp = _letters(tmp, Z, word[k])
            tmp = tmp * np.exp(lookup * alpha[k])
            tmp = np.cumsum(tmp, axis=1)
            if L - k <= extended:
                out[:, extended - (L - k), :] = tmp * np.exp(-lookup * alpha[k])
            if k < L - 1:
                tmp = _shift(tmp)
                tmp = tmp * np.exp(-lookup * alpha[k])
    else:
        for k in range(L):
            if k > 0:
                tmp = _shift(tmp)
            tmp = _letters(tmp, Z, word[k])
            if k > 0:
                tmp = tmp * np.exp(-lookup * alpha[k - 1])
            if L - k <= extended:
                out[:, extended - (L - k), :] = np.cumsum(tmp, axis=1)
            if k < L - 1:
                tmp = tmp * np.exp(lookup * alpha[k])
                tmp = np.cumsum(tmp, axis=1)
    return out


def arctic_iterated_sum_fast(Z, word, alpha, lookup, extended, total_weighting):
    """fruits/iss/semiring.py:354-400 (Arctic._iterated_sum_fast, argmax=False) with
    its bodies ``_arctic_single`` (:282-309) and ``_total_weighted_arctic_single``
    (:317-338), vectorised over N.  (max, +): letters ADD ``el * Z[dim]``, the scan
    is a running maximum, and there is no shift between letters."""
    N, _, T = Z.shape
    L = len(word)
    word = np.asarray(word, dtype=np.int32).reshape(L, -1)
    alpha = np.asarray(alpha, dtype=np.float32)
    lookup = lookup[:N]
    out = np.zeros((N, extended, T))
    tmp = np.zeros((N, T))
    for k in range(L):
        for d, el in enumerate(word[k]):
            tmp = tmp + el * Z[:, d, :]
        if total_weighting:
            tmp = tmp + lookup * alpha[k]
            tmp = np.maximum.accumulate(tmp, axis=1)
            if L - k <= extended:
                out[:, extended - (L - k), :] = tmp - lookup * alpha[k]
            if k < L - 1:
                tmp = tmp - lookup * alpha[k]
        else:
            if k > 0:
                tmp = tmp - lookup * alpha[k - 1]
            if L - k <= extended:
                out[:, extended - (L - k), :] = np.maximum.accumulate(tmp, axis=1)
            if k < L - 1:
                tmp = tmp + lookup * alpha[k]
                tmp = np.maximum.accumulate(tmp, axis=1)
    return out


def arctic_argmax_iterated_sum_fast(Z, word, alpha, lookup):
    """fruits/iss/semiring.py:239-284 (_arctic_argmax_single, called per series from
    Arctic._iterated_sum_fast :370-378): every prefix k of the word, its running maximum
    followed by the back-tracked positions of the maxima of letters 1..k+1 - L + L(L+1)/2
    rows.  The weights enter like in the non-total body whatever ``total`` says."""
    N, _, T = Z.shape
    L = len(word)
    word = np.asarray(word, dtype=np.int32).reshape(L, -1)
    alpha = np.asarray(alpha, dtype=np.float32)
    lookup = lookup[:N]
    n = L + L * (L + 1) // 2
    out = np.zeros((N, n, T))
    for j in range(N):
        result = np.zeros((2 * L, T))
        tmp = np.zeros((T,))
        for k in range(L):
            if not np.any(word[k]):
                continue
            C = np.zeros(T)
            for d, el in enumerate(word[k]):
                C = C + el * Z[j, d, :]
            tmp = tmp + C
            if k > 0:
                tmp = tmp - lookup[j] * alpha[k - 1]
            result[2 * k, 0] = tmp[0]
            for i in range(1, T):
                if result[2 * k, i - 1] >= tmp[i]:
                    result[2 * k, i] = result[2 * k, i - 1]
                    result[2 * k + 1, i] = result[2 * k + 1, i - 1]
                else:
                    result[2 * k, i] = tmp[i]
                    result[2 * k + 1, i] = i
            if k < L - 1:
                tmp = tmp + lookup[j] * alpha[k]
                tmp = np.maximum.accumulate(tmp)
        for k in range(L - 1, -1, -1):
            index = k + k * (k + 1) // 2
            out[j, index] = result[2 * k]
            out[j, index + k + 1] = result[2 * k + 1]
            for s_ in range(k, 0, -1):
                c = int(out[j, index + s_ + 1, -1]) + 1
                out[j, index + s_, :c] = result[2 * (s_ - 1) + 1, :c]
                out[j, index + s_, c:] = result[2 * (s_ - 1) + 1, c - 1]
    return out


def bayesian_iterated_sum_fast(Z, word, alpha, lookup, extended, total_weighting):
    """fruits/iss/semiring.py:530-571 (Bayesian._iterated_sum_fast) with its bodies
    ``_bayesian_single`` (:461-493) and ``_total_weighted_bayesian_single`` (:496-527),
    vectorised over N.  (max, x): the letters and exp weights of the Reals kernels, a
    running maximum instead of the cumulative sum, no shift between letters."""
    N, _, T = Z.shape
    L = len(word)
    word = np.asarray(word, dtype=np.int32).reshape(L, -1)
    alpha = np.asarray(alpha, dtype=np.float32)
    lookup = lookup[:N]
    out = np.zeros((N, extended, T))
    tmp = np.ones((N, T))
    for k in range(L):
        tmp = _letters(tmp, Z, word[k])
        if total_weighting:
            tmp = tmp * np.exp(lookup * alpha[k])
            tmp = np.maximum.accumulate(tmp, axis=1)
            if L - k <= extended:
                out[:, extended - (L - k), :] = tmp * np.exp(-lookup * alpha[k])
            if k < L - 1:
                tmp = tmp * np.exp(-lookup * alpha[k])
        else:
            if k > 0:
                tmp = tmp * np.exp(-lookup * alpha[k - 1])
            if L - k <= extended:
                out[:, extended - (L - k), :] = np.maximum.accumulate(tmp, axis=1)
            if k < L - 1:
                tmp = tmp * np.exp(lookup * alpha[k])
                tmp = np.maximum.accumulate(tmp, axis=1)
    return out


def iterated_sums(Z, word_rows, alpha=None, lookup=None, extended=1, total=False,
                  semiring="Reals"):
    """fruits/iss/semiring.py:14-41 (Semiring.iterated_sums) for a SimpleWord:
    no weighting => zero alpha, zero lookup, total=True (:27-28, :35)."""
    L = len(word_rows)
    if lookup is None:
        alpha_ = np.zeros((L,), dtype=np.float32)
        lookup_ = np.zeros((Z.shape[0], Z.shape[2]))
        total_ = True
    else:
        alpha_ = (np.ones((L,), dtype=np.float32) if alpha is None
                  else np.asarray(alpha, dtype=np.float32))
        lookup_ = lookup
        total_ = total
    fn = {"Reals": iterated_sum_fast, "Arctic": arctic_iterated_sum_fast,
          "Bayesian": bayesian_iterated_sum_fast}[semiring]
    return fn(Z, np.array(word_rows, dtype=np.int32), alpha_, lookup_, extended, total_)


def iss_transform(X, word_strings, mode="SINGLE", alphas=None, lookup=None,
                  total=False, semiring="Reals", argmax=False):
    """fruits/iss/iss.py:21-67 (_calculate_ISS, one batch of all words)
    -> (K, N, T) in the reference's row order."""
    X = np.asarray(X, dtype=np.float64)
    if argmax:
        # iss.py:37-47: L + L(L+1)/2 rows per word, EXTENDED mode only
        if mode != "EXTENDED":
            raise NotImplementedError("Arctic argmax is not implemented when using ISSMode.SINGLE")
        blocks = []
        for i, s in enumerate(word_strings):
            rows = parse_word(s)
            L = len(rows)
            if lookup is None:
                a, lk = np.zeros(L, dtype=np.float32), np.zeros((X.shape[0], X.shape[2]))
            else:
                a = (np.ones(L, dtype=np.float32) if alphas is None or alphas[i] is None
                     else np.asarray(alphas[i], dtype=np.float32))
                lk = lookup
            width = max(len(r) for r in rows)
            wmat = np.array([list(r) + [0] * (width - len(r)) for r in rows], dtype=np.int32)
            blocks.append(np.swapaxes(arctic_argmax_iterated_sum_fast(X, wmat, a, lk), 0, 1))
        return np.concatenate(blocks, axis=0)
    if mode == "EXTENDED":
        plan = cache_plan(word_strings)
    else:
        plan = [1] * len(word_strings)
    K = sum(plan)
    out = np.zeros((K, X.shape[0], X.shape[2]))
    idx = 0
    for i, s in enumerate(word_strings):
        rows = parse_word(s)
        a = None if alphas is None else alphas[i]
        r = iterated_sums(X, rows, a, lookup, plan[i], total, semiring)
        out[idx:idx + plan[i]] = np.swapaxes(r, 0, 1)
        idx += plan[i]
    return out


# --------------------------------------------------------------------------
# CosWISS (cosine weighted ISS) - "next" row
# --------------------------------------------------------------------------

def coswiss_weightings(n_letters, exponent, total):
    """fruits/iss/cos.py:265-287 (CosWISS._get_weightings): cos(a-b)^s expanded into
    products of sin/cos powers; row = [coefficient, sin/cos powers per letter ...]."""
    import itertools as it
    p = n_letters + 1 if total else n_letters
    trig_id, trig_exp, coeff = [], [exponent, 0], 1
    for k in range(exponent + 1):
        trig_id.append((coeff, trig_exp[0], trig_exp[1]))
        trig_exp[0] -= 1
        trig_exp[1] += 1
        coeff = coeff * (exponent - k) // (k + 1)
    W = np.zeros(((exponent + 1) ** (p - 1), 2 * p + 1), dtype=np.int32)
    W[:, 0] = 1
    for c, comb in enumerate(it.product(trig_id, repeat=p - 1)):
        for i in range(p - 1):
            W[c, 0] *= comb[i][0]
            W[c, 2 * i + 1] += comb[i][1]
            W[c, 2 * i + 3] += comb[i][1]
            W[c, 2 * i + 2] += comb[i][2]
            W[c, 2 * i + 4] += comb[i][2]
    return W


def coswiss_trig(T, freq):
    """sin / cos tables of fruits/iss/cos.py:23-24.  ``freq`` is float32 in the
    reference's numba signature (f4) and is promoted to float64 before the product
    with (T-1); this follows numba's typing (the un-jitted code would multiply in
    float32) - the goldens use frequencies for which both agree."""
    f = float(np.float32(freq))
    ang = np.pi * np.arange(T) / (f * (T - 1))
    return np.sin(ang), np.cos(ang)


def coswiss_ffn(X, A, b, C):
    """fruits/iss/cos.py:93-113 (_ffn) for a batch: per time step a two-layer network with
    a ReLU, Z = C relu(A x + b); the sums run over the input / hidden dimension in index
    order (numba's np.sum is a sequential loop)."""
    N, D, T = X.shape
    Y = np.zeros((N, A.shape[0], T))
    for d in range(A.shape[0]):
        acc = np.zeros((N, T))
        for i in range(D):
            acc = acc + A[d, i] * X[:, i, :]
        Y[:, d, :] = acc + b[d]
    Y = Y * (Y > 0)
    Z = np.zeros((N, C.shape[0], T))
    for d in range(C.shape[0]):
        acc = np.zeros((N, T))
        for i in range(C.shape[1]):
            acc = acc + C[d, i] * Y[:, i, :]
        Z[:, d, :] = acc
    return Z


def coswiss_transform(X, word_strings, freqs, exponent=2, total=False, ffn=None,
                      dropout_indices=None):
    """fruits/iss/cos.py:11-49,167-181,289-330 (_coswiss_single, _coswiss,
    CosWISS.batch_transform) -> (W*F, N, T), rows word-major.  ``ffn`` = (A, b, C) of
    CosWISS._fit (:250-257): the input of (word, frequency) is first sent through
    ``_ffn`` (:93-113, :128-135); ``dropout_indices`` (W, F, Lmax, rate) (:258-263): the
    summand of letter k is zeroed at these indices before its cumsum (:80)."""
    N, _, T = X.shape
    out = np.zeros((len(word_strings) * len(freqs), N, T))
    X_in = X
    for w, ws in enumerate(word_strings):
        word = parse_word(ws)
        L = len(word)
        Wt = coswiss_weightings(L, exponent, total)
        for f, freq in enumerate(freqs):
            sin_w, cos_w = coswiss_trig(T, freq)
            if ffn is not None:
                X = coswiss_ffn(X_in, ffn[0][w, f], ffn[1][w, f], ffn[2][w, f])
            res = np.zeros((N, T))
            for i in range(Wt.shape[0]):
                tmp = np.ones((N, T))
                for k in range(L):
                    if k > 0:
                        tmp = _shift(tmp)
                    tmp = _letters(tmp, X, word[k])
                    for _ in range(Wt[i, 2 * k + 1]):
                        tmp = tmp * sin_w
                    for _ in range(Wt[i, 2 * k + 2]):
                        tmp = tmp * cos_w
                    if dropout_indices is not None:
                        tmp[:, dropout_indices[w, f, k]] = 0
                    tmp = np.cumsum(tmp, axis=1)
                if Wt.shape[1] == 2 * L + 3:
                    for _ in range(Wt[i, 2 * L + 1]):
                        tmp = tmp * sin_w
                    for _ in range(Wt[i, 2 * L + 2]):
                        tmp = tmp * cos_w
                res += Wt[i, 0] * tmp
            out[w * len(freqs) + f] = res
    return out


# --------------------------------------------------------------------------
# sieves
# --------------------------------------------------------------------------

def coquantile_cuts(X_raw, q, norm="L2"):
    """fruits/cache.py:16-22,34-40 (_coquantile over _L1_sum/_L2_sum of dim 0)."""
    Y = increments(X_raw, 1)[:, 0, :]
    s = np.cumsum(np.abs(Y) if norm == "L1" else Y * Y, axis=1)
    return np.sum(s <= q * s[:, -1:], axis=1).astype(np.int64)


def transformed_cuts(N, T, cut, X_raw=None, norm="L2"):
    """fruits/sieving/segment.py:51-64 (_get_transformed_cuts)."""
    cut = tuple(cut) if isinstance(cut, (list, tuple)) else (cut,)
    new = np.zeros((N, len(cut) + 1))
    for i, c in enumerate(cut):
        if isinstance(c, float):
            new[:, i + 1] = coquantile_cuts(X_raw, c, norm)
        else:
            new[:, i + 1] = c if c >= 0 else T + c + 1
    return np.sort(new).astype(np.int64)


def fit_quantiles(q, sample=None):
    """fruits/sieving/segment.py:66-85 (_fit / _get_unfitted_quantiles)."""
    qs = np.zeros(len(q))
    fitted = False
    for i, v in enumerate(q):
        if v == 1.0:
            qs[i] = np.inf
        elif v == -1.0:
            qs[i] = -np.inf
        elif v != 0:
            if sample is None:
                raise RuntimeError("Sieve has not been fitted properly")
            qs[i] = np.quantile(sample, v)
            fitted = True
    if fitted or sample is not None:
        qs = np.sort(qs)
    return qs


def requires_fitting(q):
    """fruits/sieving/segment.py:44-49."""
    return any(v not in (-1, 0, 1) for v in q)


def pre_transform(A, inc):
    """fruits/sieving/increment.py:63-71 (IncrementSieve._pre_transform)."""
    arr = A.copy()
    if inc > 0:
        for _ in range(inc):
            arr = increments(arr[:, None, :], 1)[:, 0, :]
    elif inc < 0:
        for _ in range(-inc):
            arr = np.cumsum(arr, axis=1)
    return arr


def npi_backend(A, cuts, quantiles):
    """fruits/sieving/increment.py:107-129 (NPI._backend)."""
    N = A.shape[0]
    C, Q = cuts.shape[1] - 1, len(quantiles) - 1
    out = np.zeros((N, C * Q))
    for i in range(N):
        for j in range(C):
            seg = A[i, cuts[i, j]:cuts[i, j + 1]]
            for k in range(Q):
                out[i, j * Q + k] = np.sum(
                    np.logical_and(quantiles[k] < seg, seg <= quantiles[k + 1]))
    return out


def mpi_backend(A, cuts, quantiles):
    """fruits/sieving/increment.py:138-163 (MPI._backend)."""
    N = A.shape[0]
    C, Q = cuts.shape[1] - 1, len(quantiles) - 1
    out = np.zeros((N, C * Q))
    for i in range(N):
        for j in range(C):
            seg = A[i, cuts[i, j]:cuts[i, j + 1]]
            for k in range(Q):
                sel = seg[np.logical_and(quantiles[k] < seg,
                                         seg <= quantiles[k + 1])]
                out[i, j * Q + k] = 0 if sel.size == 0 else np.mean(sel)
    return out


def end_transform(A, cuts):
    """fruits/sieving/segment.py:210-219 (END._transform)."""
    out = np.zeros((A.shape[0], cuts.shape[1] - 1))
    for j in range(cuts.shape[1] - 1):
        out[:, j] = np.take_along_axis(A, cuts[:, j + 1:j + 2] - 1, axis=1)[:, 0]
    return out


class SieveOracle:
    """One sieve (NPI / MPI / END) with the reference's fit/transform contract
    (fruits/sieving/segment.py:14-104, increment.py:14-98)."""

    def __init__(self, kind, cut=-1, q=None, inc=1, coquantile_norm="L2"):
        self.kind = kind
        self.cut = tuple(cut) if isinstance(cut, (list, tuple)) else (cut,)
        if kind in ("NPI", "MPI"):
            self.q = tuple(q) if q is not None else (0.0, 1.0)
        else:
            self.q = tuple(q) if q is not None else (-1.0, 1.0)
        self.inc = inc if kind in ("NPI", "MPI") else 0
        self.norm = coquantile_norm
        self.quantiles = None

    def nfeatures(self):
        return len(self.cut) * (len(self.q) - 1)

    def copy(self):
        return SieveOracle(self.kind, self.cut, self.q, self.inc, self.norm)

    def fit(self, A):
        self.quantiles = fit_quantiles(self.q, pre_transform(A, self.inc))
        # (test helper: the magnitude of the rows the thresholds were taken from - the scale a
        # comparison of fitted thresholds has to use, since a quantile of differences can be tiny)
        self.fit_scale = float(np.abs(A).max()) if A.size else 0.0

    def transform(self, A, X_raw=None):
        # a sieve used on its own builds its cache from its own (N, T) input
        # (fruits/seed.py:40-52: SharedSeedCache(X[:, np.newaxis, :]))
        if X_raw is None:
            X_raw = A[:, np.newaxis, :]
        if self.kind == "END":
            cuts = transformed_cuts(A.shape[0], A.shape[1], self.cut, X_raw,
                                    self.norm)
            return end_transform(A, cuts)
        if not requires_fitting(self.q):
            self.quantiles = fit_quantiles(self.q)
        arr = pre_transform(A, self.inc)
        cuts = transformed_cuts(A.shape[0], A.shape[1], self.cut, X_raw,
                                self.norm)
        fn = npi_backend if self.kind == "NPI" else mpi_backend
        return fn(arr, cuts, self.quantiles)


    def exposure(self, A, X_raw=None, rel=1e-10, tight=False, means=None, robust_zeros=False):
        """Test helper (not part of the reference): for every feature the number of
        elements of its cut segment that lie so close to one of the band's finite thresholds
        that a band test ``q_lo < v <= q_hi`` may come out differently under a re-associated
        scan (an exact tie with a fitted quantile that IS a data point, a whole plateau of a
        running maximum equal to it, an increment the running sum absorbs next to the threshold
        0): a count may differ from the reference's by at most this number - and by nothing
        where it is 0.  (Element 0 of a series is left out: it is computed without any
        addition - or zero-padded - on every path.)

        Two widths.  ``tight=False`` (the end-to-end criterion, thresholds fitted by each side
        on its own values): ``rel * max(|A[n]|, |arr[n]|)``, A the undifferenced row.
        ``tight=True`` (the transform criterion, the SAME thresholds on both sides):
        ``rel * max|arr[n]| + 8 ulp(max|A[n]|)`` with arr the ``inc``-times differenced row
        the sieve looks at - the second term because an increment of a scan inherits the
        rounding of the running sum it is taken from, however small the increment.

        ``means`` (a dict, MPI only): filled with ``{(series, feature): candidate means}`` for
        the exposed entries of at most 4 exposed elements - the band means after moving any
        subset of the exposed elements across the threshold; a mean computed from values that
        differ in the last bits must be one of them.  ``robust_zeros``: the rows are SUMS (see
        the comment at its use)."""
        out = np.zeros((A.shape[0], self.nfeatures()), dtype=np.int64)
        if self.kind == "END":
            return out
        if not requires_fitting(self.q):
            self.quantiles = fit_quantiles(self.q)
        arr = pre_transform(A, self.inc)
        cuts = transformed_cuts(A.shape[0], A.shape[1], self.cut, X_raw, self.norm)
        if tight:
            tol = rel * np.abs(arr).max(axis=1) + 8 * np.finfo(float).eps * np.abs(A).max(axis=1)
        else:
            tol = rel * np.maximum(np.abs(A).max(axis=1), np.abs(arr).max(axis=1))
        Q = len(self.quantiles) - 1
        for i in range(A.shape[0]):
            for j in range(cuts.shape[1] - 1):
                seg = arr[i, cuts[i, j]:cuts[i, j + 1]]
                for k in range(Q):
                    near = np.zeros(seg.shape, dtype=bool)
                    for thr in (self.quantiles[k], self.quantiles[k + 1]):
                        if np.isfinite(thr):
                            close = np.abs(seg - thr) <= tol[i]
                            if tight and robust_zeros and thr == 0.0 and self.inc == 1:
                                # A first increment of a SUM that is exactly zero is a summand
                                # the running sum absorbed: exactly zero on every path that forms
                                # increments as the step of a sequential cumulative sum (the
                                # fused walk does) - it cannot land on the other side of the
                                # threshold 0.  (Not so for max-plus rows, whose inputs may differ
                                # in the last bit, nor for higher differences.)
                                close &= seg != 0.0
                            near |= close
                    # t = 0 is exact on both sides (the first value of a scan is its first
                    # summand; increments are zero-padded there): never a disagreement
                    if cuts[i, j] == 0 and near.size:
                        near[0] = False
                    n_near = int(near.sum())
                    out[i, j * Q + k] = n_near
                    if means is not None and self.kind == "MPI" and 0 < n_near <= 4:
                        inside = np.logical_and(self.quantiles[k] < seg, seg <= self.quantiles[k + 1])
                        where = np.nonzero(near)[0]
                        cands = []
                        for bits in range(1 << n_near):
                            m = inside.copy()
                            for b, pos in enumerate(where):
                                if bits >> b & 1:
                                    m[pos] = not m[pos]
                            cands.append(0.0 if not m.any() else float(np.mean(seg[m])))
                        means[(i, j * Q + k)] = cands
        return out


# --------------------------------------------------------------------------
# whole pipeline (the spec format of tests/golden/golden.json "fruit" cases)
# --------------------------------------------------------------------------

def _apply_preps(X, preps):
    for p in preps:
        kind = p["kind"]
        if kind == "INC":
            X = inc_transform(X, p.get("shift", 1), p.get("depth", 1),
                              p.get("zero_padding", True))
        elif kind == "STD":
            X = std_transform(X, p.get("var", True), p.get("std_eps", 1e-5))
        elif kind == "NEW":
            inner = p.get("inner")
            fn = None if inner is None else (
                lambda Y, inner=inner: _apply_preps(Y, [inner]))
            X = new_transform(X, fn)
        else:
            raise NotImplementedError(kind)
    return X


def lookup_l2(X_for_l2, relative=False, scale=50.0):
    """fruits/iss/weighting.py:198-210 (L2.get_lookup, no transform)."""
    r = l2_sum(X_for_l2)
    if relative:
        r = r / (r[:, -1:] + 1e-5)
    return nrm_rows(r) * scale


def _weight_lookup(spec, X_prepared, X_raw):
    if spec is None:
        return None, False
    kind = spec["kind"]
    total = spec.get("total", False)
    scale = spec.get("scale", 50)
    if kind == "Indices":
        return lookup_indices(X_prepared.shape[0], X_prepared.shape[2],
                              spec.get("relative", True), scale), total
    if kind == "L1":
        src = X_prepared if spec.get("on_prepared", False) else X_raw
        return lookup_l1(src, spec.get("relative", False), scale), total
    if kind == "L2":
        src = X_prepared if spec.get("on_prepared", False) else X_raw
        return lookup_l2(src, spec.get("relative", False), scale), total
    raise NotImplementedError(kind)


def _iterate_iss(X, iss_list, X_raw, idx=0):
    """fruits/fruit.py:440-454 (_iterate_iss): chained ISS, one (N,T) at a time."""
    if idx == len(iss_list):
        yield X[:, 0, :]
        return
    i = iss_list[idx]
    if i.get("kind") == "CosWISS":
        its = coswiss_transform(X, i["words"], i["freqs"], i.get("exponent", 2),
                                i.get("total_weighting", False))
    else:
        lookup, total = _weight_lookup(i.get("weighting"), X, X_raw)
        its = iss_transform(X, i["words"], i["mode"], i.get("alphas"), lookup, total,
                            i.get("semiring", "Reals"), argmax=i.get("argmax", False))
    for itsum in its:
        yield from _iterate_iss(itsum[:, None, :], iss_list, X_raw, idx + 1)


def _make_sieves(slice_spec):
    out = []
    for s in slice_spec["sieves"]:
        kw = {k: v for k, v in s.items() if k != "kind"}
        out.append(SieveOracle(s["kind"], **kw))
    return out


def fruit_fit(spec, X, np_seed=None):
    """fruits/fruit.py:121-136, 456-496 (Fruit.fit / FruitSlice.fit)."""
    if np_seed is not None:
        np.random.seed(np_seed)
    fitted = []
    for sl in spec["slices"]:
        fss = sl.get("fit_sample_size", 1)
        if isinstance(fss, int) and fss == 1:
            ind = np.random.randint(0, X.shape[0])
            sample_raw = X[ind:ind + 1]
        else:
            s = max(int(fss * X.shape[0]), 1)
            sample_raw = X[np.random.choice(X.shape[0], size=s, replace=False)]
        sample = _apply_preps(sample_raw, sl.get("preps", []))
        sieves = _make_sieves(sl)
        ext = []
        if any(requires_fitting(s.q) for s in sieves):
            # NB: the L1 lookup during fit uses the fruit-level cache, i.e. the
            # FULL raw input X (fruits/fruit.py:132-135, cache.py:98-112).
            for itsum in _iterate_iss(sample, sl["iss"], X):
                cp = [s.copy() for s in sieves]
                for s in cp:
                    s.fit(itsum)
                ext.append(cp)
        fitted.append((sieves, ext))
    return fitted


def fruit_transform(spec, fitted, X):
    """fruits/fruit.py:138-173, 498-553 (Fruit.transform / FruitSlice.transform)."""
    blocks = []
    for sl, (sieves, ext) in zip(spec["slices"], fitted):
        P = _apply_preps(X, sl.get("preps", []))
        cols = []
        for i, itsum in enumerate(_iterate_iss(P, sl["iss"], X)):
            for s in (ext[i] if ext else sieves):
                cols.append(s.transform(itsum, X))
        blocks.append(np.concatenate(cols, axis=1))
    res = np.concatenate(blocks, axis=1)
    return np.nan_to_num(res, copy=False, nan=0.0)


def fruit_transform_exposure(spec, fitted, X, rel=1e-10, tight=False, means=None):
    """fruit_transform and, from the same iterated sums, the (N, F) near-threshold element
    counts of every feature (SieveOracle.exposure, a test helper) in the same column order.
    ``means``: a dict filled with {(series, column): candidate band means} of the exposed MPI
    entries (see SieveOracle.exposure)."""
    feats, expos = [], []
    col0 = 0
    for sl, (sieves, ext) in zip(spec["slices"], fitted):
        P = _apply_preps(X, sl.get("preps", []))
        sums = all(i.get("semiring", "Reals") == "Reals" and i.get("kind") != "CosWISS" for i in sl["iss"])
        cols, ecols = [], []
        for i, itsum in enumerate(_iterate_iss(P, sl["iss"], X)):
            for s in (ext[i] if ext else sieves):
                cols.append(s.transform(itsum, X))
                local = {} if means is not None else None
                ecols.append(s.exposure(itsum, X, rel, tight, local, robust_zeros=sums))
                if local:
                    for (n, f), c in local.items():
                        means[(n, col0 + f)] = c
                col0 += cols[-1].shape[1]
        feats.append(np.concatenate(cols, axis=1))
        expos.append(np.concatenate(ecols, axis=1))
    res = np.nan_to_num(np.concatenate(feats, axis=1), copy=False, nan=0.0)
    return res, np.concatenate(expos, axis=1)


def fruit_exposure(spec, fitted, X, rel=1e-10, tight=False, means=None):
    return fruit_transform_exposure(spec, fitted, X, rel, tight, means)[1]


def fitted_thresholds(fitted):
    """The thresholds of a fitted fruit: per slice, per iterated sum, per sieve the sorted
    quantile values (None for a sieve without any: END)."""
    return [[[None if s.kind == "END" else (None if s.quantiles is None else np.array(s.quantiles))
              for s in row] for row in ext] for _, ext in fitted]


def fitted_scales(fitted):
    """Same shape as fitted_thresholds: max |value| of the rows every sieve copy was fitted on."""
    return [[[getattr(s, "fit_scale", 0.0) for s in row] for row in ext] for _, ext in fitted]
