"""The ISS stage (mirrors fruits/iss/iss.py).

``ISS(words, mode=..., semiring=..., weighting=...)`` computes the iterated sums
``(K, N, T)`` of a batch ``(N, D, T)``.  Where the reference loops over the words
in Python and calls one numba kernel per word (fruits/iss/iss.py:49-65), this
stage compiles the whole word list into ONE device program (a prefix trie in DFS
order, csrc/plan.cpp) and walks it in one HIP launch; the result has the
reference's row order and layout (iss.py:46).
"""
from __future__ import annotations

import os
from enum import Enum, auto
from typing import Generator, Optional, Sequence

import numpy as np

from .. import _native as nat
from ..cache import SharedSeedCache
from ..seed import Seed
from .cache import CachePlan
from .semiring import Arctic, Bayesian, Reals, Semiring

_AUTO_PREPARE_BYTES = 64 << 20   # materialised output from which a plan is prepared unasked
from .weighting import Weighting
from .words.word import SimpleWord, Word


class ISSMode(Enum):
    """SINGLE: one iterated sum per word.  EXTENDED: additionally every prefix
    of every word that no earlier word already produced."""

    SINGLE = auto()
    EXTENDED = auto()


def _device_budget_bytes() -> int:
    return int(float(os.environ.get("FRUITS_AMD_ISS_GIB", "8")) * (1 << 30))


class ISS(Seed):
    def __init__(self, words: Sequence[Word], /, *, mode: ISSMode = ISSMode.SINGLE,
                 semiring: Optional[Semiring] = None,
                 weighting: Optional[Weighting] = None) -> None:
        self.words = words
        self.mode = mode
        self.semiring = semiring if semiring is not None else Reals()
        self._cache_plan = CachePlan(self.words if mode == ISSMode.EXTENDED else [])
        self.weighting = weighting
        self._fresh_transients()

    def _fresh_transients(self) -> None:
        self._plans: dict = {}      # device programs by (words, mode, weighting)

    @property
    def requires_fitting(self) -> bool:
        return False

    def _fit(self, X: np.ndarray) -> None:
        pass

    # ------------------------------------------------------------------ plans
    @property
    def _argmax(self) -> bool:
        return isinstance(self.semiring, Arctic) and self.semiring._argmax

    def _depth(self, i: int) -> int:
        """Rows word i contributes (EXTENDED: CachePlan's depth; Arctic argmax: every prefix
        with the positions of its maxima, L + L(L+1)/2 rows - fruits/iss/iss.py:37-47)."""
        if self._argmax:
            L = len(self.words[i])
            return L + L * (L + 1) // 2
        return self._cache_plan.unique_el_depth(i) if self.mode == ISSMode.EXTENDED else 1

    def _rows_of(self, start: int, stop: int) -> int:
        return sum(self._depth(i) for i in range(start, stop))

    def _check_supported(self) -> None:
        if self._argmax and self.mode == ISSMode.SINGLE:
            raise NotImplementedError(
                "Arctic argmax is not implemented when using ISSMode.SINGLE")
        if isinstance(self.semiring, Arctic):
            pass
        elif not isinstance(self.semiring, (Reals, Bayesian)):
            raise NotImplementedError(
                f"semiring {type(self.semiring).__name__} is not on the MI355X hot path")
        for w in self.words:
            if not isinstance(w, SimpleWord):
                raise NotImplementedError(
                    "only SimpleWord is supported by the MI355X implementation")

    def _plan(self, start: int, stop: int) -> nat.Plan:
        return self._plan_indices(tuple(range(start, stop)))

    def _plan_indices(self, indices) -> nat.Plan:
        """Device program of an arbitrary subset of the words (in the given
        order).  Output depths come from the cache plan of the FULL word list, so
        shards of a word list (fruits_amd.parallel) emit exactly their share of
        the rows."""
        indices = tuple(indices)
        if self.weighting is None:
            wmode, alphas = nat.FR_W_NONE, None
        else:
            # (the argmax body weights like the non-total one whatever `total` says,
            # fruits/iss/semiring.py:262-263,271-272)
            wmode = (nat.FR_W_TOTAL if self.weighting.total and not self._argmax
                     else nat.FR_W_NONTOTAL)
            alphas = [np.asarray(self.words[i].alpha, dtype=np.float32) for i in indices]
        arctic = isinstance(self.semiring, Arctic)
        bayesian = isinstance(self.semiring, Bayesian)
        key = (indices, self.mode, wmode, arctic, bayesian,
               None if alphas is None else tuple(a.tobytes() for a in alphas))
        plan = self._plans.get(key)
        if plan is None:
            # argmax: the running maxima of ALL prefixes of every word (depth = its length)
            depths = ([len(self.words[i]) for i in indices] if self._argmax
                      else [self._depth(i) for i in indices])
            plan = nat.Plan([self.words[i].table() for i in indices], depths, alphas, wmode,
                            arctic=arctic, bayesian=bayesian, letter_sum=self._argmax)
            self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------ device path
    def _attach_cache(self, X) -> None:
        if self.weighting is None:
            return
        if hasattr(self, "_cache"):
            self.weighting._cache = self._cache
        else:
            self.weighting._cache = SharedSeedCache(X)

    def lookup_device(self, Xd):
        """(rows, T) device lookup of the weighting, rows in {1, N}; when the
        cache holds more series than Xd (fit on a sub-sample) the leading rows
        are used - the reference indexes ``lookup[j]`` by position
        (fruits/iss/semiring.py:185-199)."""
        if self.weighting is None:
            return None
        # (max-plus results meet fitted quantiles in exact ties: bit-exact lookup there)
        lk = self.weighting.lookup_device(Xd, exact=not isinstance(self.semiring, Reals))
        n = int(Xd.shape[0])
        if lk.shape[0] != 1:
            if lk.shape[0] < n:
                raise IndexError("weighting lookup has fewer rows than the input")
            lk = lk[:n]
        return lk.contiguous()

    def word_batches(self, N: int, T: int, batch_size: Optional[int] = None):
        """[start, stop) word ranges: ``batch_size`` words each, or as many as
        fit the device budget."""
        W = len(self.words)
        if batch_size is not None:
            return [(s, min(s + batch_size, W)) for s in range(0, W, batch_size)]
        budget = max(_device_budget_bytes() // max(8 * N * T, 1), 1)
        out, s = [], 0
        while s < W:
            e, rows = s, 0
            while e < W and (e == s or rows + self._depth(e) <= budget):
                rows += self._depth(e)
                e += 1
            out.append((s, e))
            s = e
        return out

    def transform_device(self, Xd, start: int = 0, stop: Optional[int] = None,
                         lookup_d="auto", out=None, groups: int = 0, indices=None):
        """Iterated sums of words [start, stop) (or of ``indices``) as a
        (K, N, T) device tensor."""
        self._check_supported()
        stop = len(self.words) if stop is None else stop
        if self._argmax:
            # values of every prefix, then positions + back-tracking (fr_arctic_argmax)
            idx = tuple(range(start, stop)) if indices is None else tuple(indices)
            plan = self._plan_indices(idx)
            if plan.max_dim > Xd.shape[1]:
                raise IndexError(
                    f"a word references dimension {plan.max_dim} but the input has "
                    f"only {Xd.shape[1]}")
            if isinstance(lookup_d, str):
                lookup_d = self.lookup_device(Xd)
            V = plan.run(Xd, lookup_d, layout="KNT", groups=groups)
            res = nat.arctic_argmax(V, [len(self.words[i]) for i in idx])
            if out is not None:
                out.copy_(res)
                return out
            return res
        plan = self._plan(start, stop) if indices is None else self._plan_indices(indices)
        if plan.max_dim > Xd.shape[1]:
            raise IndexError(
                f"a word references dimension {plan.max_dim} but the input has "
                f"only {Xd.shape[1]}")
        if isinstance(lookup_d, str):
            lookup_d = self.lookup_device(Xd)
        T = int(Xd.shape[2])
        if not plan.fits(T):
            # many distinct alphas -> more exp tables than a workgroup can stage: halves
            if indices is None:
                indices = tuple(range(start, stop))
            if len(indices) == 1:
                raise NotImplementedError(
                    f"word {self.words[indices[0]]} alone needs {plan.staged_rows} staged rows "
                    "(input dimensions + exp tables of its distinct alphas): more than the "
                    "LDS of a workgroup holds")
            if out is None:
                out = nat.torch().empty((plan.rows, int(Xd.shape[0]), T),
                                        dtype=nat.torch().float64, device=Xd.device)
            half = len(indices) // 2
            k0 = sum(self._depth(i) for i in indices[:half])
            self.transform_device(Xd, lookup_d=lookup_d, out=out[:k0], groups=groups,
                                  indices=indices[:half])
            self.transform_device(Xd, lookup_d=lookup_d, out=out[k0:], groups=groups,
                                  indices=indices[half:])
            return out
        # A big materialising launch of a small plan: prepare it once (fr_plan_prepare compiles
        # the plan's static program with hipRTC when no pre-compiled one matches - about a
        # second the first time a word list is seen, milliseconds from the disk cache after).
        # FRUITS_AMD_AUTO_PREPARE=0 leaves that to explicit Plan.prepare calls.
        N, T = int(Xd.shape[0]), int(Xd.shape[2])
        if (8 * plan.rows * N * T >= _AUTO_PREPARE_BYTES and not getattr(plan, "_prepared", False)
                and os.environ.get("FRUITS_AMD_AUTO_PREPARE", "1") != "0"
                and not nat.torch().cuda.is_current_stream_capturing()):
            plan.prepare(N, T, groups)
            plan._prepared = True
        return plan.run(Xd, lookup_d, out=out, layout="KNT", groups=groups)

    # ------------------------------------------------------------------ reference API
    @staticmethod
    def _validate(X) -> np.ndarray:
        if not isinstance(X, np.ndarray) or X.dtype != np.float64 or X.ndim != 3:
            raise TypeError("input has to be a float64 array of shape (N, D, T)")
        return X

    def _transform(self, X: np.ndarray) -> np.ndarray:
        X = self._validate(X)
        self._attach_cache(X)
        Xd = nat.to_device(X)
        return nat.to_host(self.transform_device(Xd))

    def n_iterated_sums(self) -> int:
        """Number of iterated sums ``transform`` returns."""
        if self.mode == ISSMode.EXTENDED:
            if self._argmax:
                return sum(len(w) + len(w) * (len(w) + 1) // 2 for w in self.words)
            return self._cache_plan.n_iterated_sums()
        if self._argmax:
            raise NotImplementedError(
                "Arctic argmax is not implemented when using ISSMode.SINGLE")
        return len(self.words)

    def batch_transform(self, X: np.ndarray,
                        batch_size: int = 1) -> Generator[np.ndarray, None, None]:
        """Yields ``(k, N, T)`` arrays, the iterated sums of ``batch_size`` words
        at a time (fruits/iss/iss.py:152-185)."""
        if batch_size > len(self.words):
            raise ValueError("batch_size too large, has to be < len(words)")
        X = self._validate(X)
        self._attach_cache(X)
        Xd = nat.to_device(X)
        lookup_d = self.lookup_device(Xd)
        for s, e in self.word_batches(X.shape[0], X.shape[2], batch_size):
            yield nat.to_host(self.transform_device(Xd, s, e, lookup_d))

    def _copy(self) -> "ISS":
        return ISS(self.words, mode=self.mode, semiring=self.semiring,
                   weighting=self.weighting)

    def _label(self, index: int) -> str:
        if self.mode == ISSMode.EXTENDED:
            text = self._cache_plan.get_word_string(index)
        else:
            text = str(self.words[index])
        if not isinstance(self.semiring, Reals):
            text += " : " + self.semiring.__class__.__name__
        if self.weighting is not None:
            text += " : " + self.weighting.__class__.__name__
        return text
