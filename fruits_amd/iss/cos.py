"""Cosine weighted ISS (mirrors fruits/iss/cos.py, including the randomised ffn and
dropout variants on the factorised kernels).

``cos(pi*(i_k - i_{k+1})/(f*(T-1)))**s`` expands into products of powers of
``sin`` and ``cos`` of the single time steps (fruits/iss/cos.py:265-287), so every
product - a *term* - is an ordinary Reals iterated sum of the word over the input
extended by one sin and one cos row.  The reference evaluates the
``(s+1)**(p-1)`` terms one after another in numba (cos.py:11-49).  Two device paths:

* exponents 1..8 (``fr_plan_create_coswiss``, csrc/coswiss.h): the sum over the terms
  factorises letter by letter, so one workgroup per (series, word, frequency) needs
  ``s+1`` scans per letter; the plan behaves like any other (``fr_iss_run``, fused
  sieve pipelines).
* larger exponents: the reference's own formulation - all terms of all words of one
  frequency compiled into ONE prefix-sharing walk program, reduced by
  ``fr_coswiss_combine`` in the reference's order.
"""
from __future__ import annotations

import itertools
import os
from typing import Generator, Optional, Sequence

import numpy as np

from .. import _native as nat
from .iss import ISS, _device_budget_bytes
from .words.word import SimpleWord, Word


class CosWISS(ISS):
    def __init__(self, words: Sequence[Word], freqs: Sequence[float], exponent: int = 2,
                 total_weighting: bool = False, ffn_size: Optional[int] = None,
                 dropout: Optional[float] = None) -> None:
        super().__init__(words)
        for word in words:
            if not isinstance(word, SimpleWord):
                raise ValueError("CosWISS only implemented for simple words")
        self._total_weighting = total_weighting
        self._freqs = freqs
        self._exponent = exponent
        self._ffn_size = ffn_size
        self._dropout = dropout

    def _fresh_transients(self) -> None:
        super()._fresh_transients()
        self._programs: dict = {}    # term programs (device handles) of the non-factorised path

    def n_iterated_sums(self) -> int:
        return len(self._freqs) * len(self.words)

    @property
    def requires_fitting(self) -> bool:
        return self._ffn_size is not None or self._dropout is not None

    def _check_supported(self) -> None:
        if (self._ffn_size is not None or self._dropout is not None) and not self._native():
            raise NotImplementedError(
                "the randomised ffn / dropout variants of CosWISS need the factorised kernels "
                f"(exponent <= {nat.CosPlan.MAX_EXPONENT})")
        if self._ffn_size is not None and not hasattr(self, "_A"):
            raise RuntimeError("Missing call of self.fit()")
        if self._dropout is not None and self._ffn_size is None \
                and not hasattr(self, "_dropout_indices"):
            raise RuntimeError("Missing call of self.fit()")

    def _fit(self, X: np.ndarray) -> None:
        """Draws the random state exactly like the reference (fruits/iss/cos.py:248-263), from
        numpy's global generator and in the same order, so a seeded run picks the same
        weights and indices."""
        if (d := self._ffn_size) is not None:
            W, F = len(self.words), len(self._freqs)
            self._A = np.random.random((W, F, d, X.shape[1]))
            self._b = np.random.random((W, F, d))
            self._C = np.random.random((W, F, X.shape[1], d))
        if (d := self._dropout) is not None:
            rate = int(d * X.shape[2])
            self._dropout_indices = np.array([
                [[np.random.choice(X.shape[2], size=(rate,), replace=False)
                  for _ in range(max(map(len, self.words)))]
                 for _ in range(len(self._freqs))]
                for _ in range(len(self.words))
            ], dtype=np.int32)
            self._dropout_T = int(X.shape[2])
        self._plans = {}

    # ------------------------------------------------------------------ host tables
    def _get_weightings(self, word: Word) -> np.ndarray:
        """Rows ``[coefficient, sin power, cos power, ...]`` (one pair per letter and
        one more with total weighting) of the expansion of the cosine powers
        (fruits/iss/cos.py:265-287)."""
        p = len(word) + 1 if self._total_weighting else len(word)
        s = self._exponent
        binom = [1]
        for k in range(s):
            binom.append(binom[-1] * (s - k) // (k + 1))
        pairs = [(binom[k], s - k, k) for k in range(s + 1)]       # C(s,k) sin^(s-k) cos^k
        out = np.zeros(((s + 1) ** (p - 1), 2 * p + 1), dtype=np.int32)
        out[:, 0] = 1
        for c, comb in enumerate(itertools.product(pairs, repeat=p - 1)):
            for i, (coef, ps, pc) in enumerate(comb):
                out[c, 0] *= coef
                out[c, 2 * i + 1] += ps
                out[c, 2 * i + 3] += ps
                out[c, 2 * i + 2] += pc
                out[c, 2 * i + 4] += pc
        return out

    def _trig(self, T: int) -> np.ndarray:
        """(F, 2, T) sin / cos rows (cos.py:23-24); the frequency is rounded to
        float32 and promoted to float64 before the product with T-1, as numba types
        the reference's ``f4`` argument."""
        out = np.empty((len(self._freqs), 2, T))
        with np.errstate(all="ignore"):
            for f, freq in enumerate(self._freqs):
                ang = np.pi * np.arange(T) / (float(np.float32(freq)) * (T - 1))
                out[f, 0], out[f, 1] = np.sin(ang), np.cos(ang)
        return out

    def _program(self, start: int, stop: int, D: int):
        """Term program of words [start, stop) for ONE frequency whose sin / cos rows
        sit at dimensions D+1, D+2 of the extended input: (plan, begin, coeff, desc)."""
        key = (start, stop, D)
        prog = self._programs.get(key)
        if prog is not None:
            return prog
        tables, begin, coeff, desc = [], [0], [], []
        for w in range(start, stop):
            base = np.asarray(self.words[w].table(), dtype=np.int32)
            L = base.shape[0]
            if base.shape[1] > D:
                raise IndexError(
                    f"a word references dimension {base.shape[1]} but the input has only {D}")
            wt = self._get_weightings(self.words[w])
            for row in wt:
                tab = np.zeros((L, D + 2), dtype=np.int32)
                tab[:, :base.shape[1]] = base
                tab[:, D] = row[1:2 * L + 1:2]
                tab[:, D + 1] = row[2:2 * L + 2:2]
                tables.append(tab)
                coeff.append(float(row[0]))
                tail = row[2 * L + 1:2 * L + 3] if self._total_weighting else (0, 0)
                desc.append((len(tables) - 1, int(tail[0]), int(tail[1])))
            begin.append(len(tables))
        plan = nat.Plan(tables, [1] * len(tables), None, nat.FR_W_NONE)
        t = nat.torch()
        prog = (plan,
                t.from_numpy(np.asarray(begin, dtype=np.int32)).cuda(),
                t.from_numpy(np.asarray(coeff, dtype=np.float64)).cuda(),
                t.from_numpy(np.asarray(desc, dtype=np.int32).reshape(-1, 3)).cuda())
        self._programs[key] = prog
        return prog

    def _native(self) -> bool:
        """Whether the factorised device kernels cover this configuration."""
        return (1 <= self._exponent <= nat.CosPlan.MAX_EXPONENT
                and max((len(w) for w in self.words), default=0) <= nat.CosPlan.MAX_LETTERS
                and os.environ.get("FRUITS_AMD_COSWISS_TERMS", "0") != "1")

    def _plan(self, start: int, stop: int) -> nat.Plan:
        return self._plan_indices(tuple(range(start, stop)))

    def _plan_indices(self, indices) -> nat.Plan:
        """Program of an arbitrary subset of the words (a rank's share of a word-sharded
        slice): rows word-major in the given order, F rows per word."""
        indices = tuple(indices)
        key = ("cos", indices)
        plan = self._plans.get(key)
        if plan is None:
            plan = nat.CosPlan([self.words[i].table() for i in indices], self._freqs,
                               self._exponent, self._total_weighting)
            self._plans[key] = plan
        return plan

    def _arm_plan(self, plan, idx, T: int) -> None:
        """Hands the dropout indices of the words ``idx`` to their plan (a device mask) -
        before a run of the plan itself or of a fused pipeline built on it."""
        if self._dropout is None or self._ffn_size is not None:
            return
        # The reference zeroes tmp[dropout[k]] (fruits/iss/cos.py:84) whatever the length of the
        # series: indices drawn on the fit's length work on any input they fit into, and only one
        # beyond the input's end is an IndexError (raised while the mask for T is built).
        if getattr(plan, "_dropout_set", None) != (id(self._dropout_indices), T):
            nat.coswiss_set_dropout(plan, self._dropout_indices[list(idx)], T)
            plan._dropout_set = (id(self._dropout_indices), T)

    def _n_terms(self, w: int) -> int:
        p = len(self.words[w]) + 1 if self._total_weighting else len(self.words[w])
        return (self._exponent + 1) ** (p - 1)

    # ------------------------------------------------------------------ device path
    def lookup_device(self, Xd):
        return None

    def _depth(self, i: int) -> int:
        return len(self._freqs)

    def word_batches(self, N: int, T: int, batch_size: Optional[int] = None):
        W = len(self.words)
        if batch_size is not None:
            return [(s, min(s + batch_size, W)) for s in range(0, W, batch_size)]
        budget = max(_device_budget_bytes() // max(8 * N * T, 1), 1)
        native = self._native()
        out, s = [], 0
        while s < W:
            e, rows = s, 0
            while e < W:
                need = len(self._freqs) + (0 if native else self._n_terms(e))
                if e > s and rows + need > budget:
                    break
                rows += need
                e += 1
            out.append((s, e))
            s = e
        return out

    def transform_device(self, Xd, start: int = 0, stop: Optional[int] = None,
                         lookup_d=None, out=None, groups: int = 0, indices=None):
        """Rows [start*F, stop*F) of the result, a ((stop-start)*F, N, T) device tensor."""
        self._check_supported()
        t = nat.torch()
        stop = len(self.words) if stop is None else stop
        N, D, T = (int(v) for v in Xd.shape)
        F = len(self._freqs)
        if indices is not None:
            if not self._native():
                raise NotImplementedError("word subsets need the factorised CosWISS kernels")
            n_words = len(indices)
        else:
            n_words = stop - start
        if out is None:
            out = t.empty((n_words * F, N, T), dtype=t.float64, device=Xd.device)
        if out.numel() == 0 or n_words == 0:
            return out
        if self._native():
            idx = tuple(range(start, stop)) if indices is None else tuple(indices)
            if self._ffn_size is not None:
                # _ffn_coswiss (cos.py:116-137; batch_transform tests ffn first, :306-314):
                # every (word, frequency) has its own transformed input - word by word, the
                # F copies stacked, unit f of the word's plan reading copy f
                Z = t.empty((F, N, D, T), dtype=t.float64, device=Xd.device)
                for i, w in enumerate(idx):
                    plan = self._plan_indices((w,))
                    if plan.max_dim > D:
                        raise IndexError(f"a word references dimension {plan.max_dim} but "
                                         f"the input has only {D}")
                    for f in range(F):
                        nat.coswiss_ffn(Xd, self._A[w, f], self._b[w, f], self._C[w, f], Z[f])
                    nat.check(nat.lib().fr_coswiss_set_input_stride(plan._h, N * D * T))
                    plan.run(Z[0], None, out=out[i * F:(i + 1) * F], layout="KNT")
                return out
            plan = self._plan(start, stop) if indices is None else self._plan_indices(indices)
            if plan.max_dim > D:
                raise IndexError(
                    f"a word references dimension {plan.max_dim} but the input has only {D}")
            self._arm_plan(plan, idx, T)
            return plan.run(Xd, None, out=out, layout="KNT")
        plan, begin_d, coeff_d, desc_d = self._program(start, stop, D)
        trig = t.from_numpy(self._trig(T)).to(Xd.device)              # (F, 2, T)
        Xa = t.empty((N, D + 2, T), dtype=t.float64, device=Xd.device)
        Xa[:, :D] = Xd
        terms = None
        for f in range(F):
            Xa[:, D:] = trig[f]
            terms = plan.run(Xa, None, out=terms, layout="KNT", groups=groups)
            nat.coswiss_combine(terms, begin_d, coeff_d, desc_d, trig[f], out[f], F * N * T)
        return out

    # ------------------------------------------------------------------ reference API
    def _transform(self, X: np.ndarray) -> np.ndarray:
        X = self._validate(X)
        Xd = nat.to_device(X)
        blocks = [self.transform_device(Xd, s, e)
                  for s, e in self.word_batches(X.shape[0], X.shape[2])]
        return nat.to_host(blocks[0] if len(blocks) == 1 else nat.torch().cat(blocks))

    def batch_transform(self, X: np.ndarray,
                        batch_size: int = 1) -> Generator[np.ndarray, None, None]:
        """Yields ``(batch_size*F, N, T)`` arrays, the sums of ``batch_size`` words at
        a time for all frequencies (fruits/iss/cos.py:289-330)."""
        X = self._validate(X)
        Xd = nat.to_device(X)
        for s, e in self.word_batches(X.shape[0], X.shape[2], batch_size):
            yield nat.to_host(self.transform_device(Xd, s, e))

    def _copy(self) -> "CosWISS":
        return CosWISS(freqs=self._freqs, words=self.words, exponent=self._exponent,
                       total_weighting=self._total_weighting, ffn_size=self._ffn_size,
                       dropout=self._dropout)

    def _label(self, index: int) -> str:
        d, r = divmod(index, len(self._freqs))
        string = str(self.words[d])
        string += f"!{self._freqs[r]} : ^{self._exponent}"
        if self._total_weighting:
            string += " : total"
        return string
