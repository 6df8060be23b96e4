"""The operator seam (mirrors fruits/iss/semiring.py:12-52, 161-232).

``Semiring.iterated_sum_fast(Z, word, alpha, lookup, extended, total_weighting)``
is the reference's per-word operator; :class:`Reals` implements it with the HIP
trie-walk kernel through the C ABI (``fr_iterated_sum_fast_host``).  The batched
device path used by :class:`fruits_amd.ISS` goes through the same kernel for all
words at once.  Arctic (a "next" row of the scope table) runs on the same kernel;
Bayesian (max, x) runs on the same kernel as well.
"""
from __future__ import annotations

from abc import ABC
from typing import Optional

import numpy as np

from .. import _native as nat
from .weighting import Weighting
from .words.word import SimpleWord, Word


class Semiring(ABC):
    def iterated_sums(self, Z: np.ndarray, word: Word, extended: int,
                      weighting: Optional[Weighting] = None) -> np.ndarray:
        """(N, extended, T) iterated sums of one word (semiring.py:14-41)."""
        if not isinstance(word, SimpleWord):
            raise NotImplementedError(
                "only SimpleWord is supported by the MI355X implementation")
        if weighting is not None:
            lookup = weighting.get_lookup(Z)
            alpha = word.alpha
            total = weighting.total
        else:
            # the reference passes alpha = 0, lookup = 0, total=True (semiring.py:27-35):
            # every exp factor is exactly 1, which the device skips
            lookup, alpha, total = None, None, True
        return self.iterated_sum_fast(Z, word.table(), alpha, lookup, extended, total)

    def iterated_sum_fast(self, Z: np.ndarray, word: np.ndarray, alpha: np.ndarray,
                          lookup: np.ndarray, extended: int,
                          total_weighting: bool) -> np.ndarray:
        raise NotImplementedError("No fast way of calculating iterated sums")


def _check_input(Z) -> np.ndarray:
    if not isinstance(Z, np.ndarray) or Z.dtype != np.float64 or Z.ndim != 3:
        # numba's "no matching definition" for f8[:,:,:] in the reference
        raise TypeError("input has to be a float64 array of shape (N, D, T)")
    return np.ascontiguousarray(Z)


class Reals(Semiring):
    """(R, +, x): the default semiring."""

    def iterated_sum_fast(self, Z, word, alpha, lookup, extended, total_weighting):
        Z = _check_input(Z)
        word = np.asarray(word, dtype=np.int32)
        if word.ndim != 2:
            raise TypeError("word has to be an (L, Dw) int32 table")
        if lookup is not None and not np.any(lookup) and (alpha is None or not np.any(alpha)):
            lookup = None  # the reference's explicit "unweighted" call
        return nat.iterated_sum_fast_host(Z, word, alpha, lookup, extended, total_weighting)


class Arctic(Semiring):
    """(R u {-inf}, max, +): letters add ``el * Z[dim]``, the scan is a running
    maximum and - unlike Reals - there is no shift between letters
    (fruits/iss/semiring.py:282-400).  Same HIP kernel with the scan operator and
    the letter operation exchanged; results are bit-exact (max is associative).
    ``argmax=True`` (fruits/iss/semiring.py:239-284): every prefix of the word with the
    back-tracked positions of its maxima, ``L + L(L+1)/2`` rows - the running maxima from
    the same kernel, positions and back-tracking by ``fr_arctic_argmax``."""

    def __init__(self, argmax: bool = False) -> None:
        self._argmax = argmax

    def iterated_sum_fast(self, Z, word, alpha, lookup, extended, total_weighting):
        Z = _check_input(Z)
        word = np.asarray(word, dtype=np.int32)
        if lookup is not None and not np.any(lookup) and (alpha is None or not np.any(alpha)):
            lookup = None
        if self._argmax:
            # the per-word operator: all L prefixes (non-total weights whatever the flag),
            # then positions + back-tracking on the device
            L = int(word.shape[0])
            plan = nat.Plan([word], [L], None if lookup is None else [alpha],
                            nat.FR_W_NONE if lookup is None else nat.FR_W_NONTOTAL, arctic=True,
                            letter_sum=True)
            Zd = nat.to_device(Z)
            lk = None if lookup is None else nat.to_device(np.asarray(lookup, dtype=np.float64))
            V = plan.run(Zd, lk, layout="KNT")
            return nat.to_host(nat.arctic_argmax(V, [L]).permute(1, 0, 2).contiguous())
        return nat.iterated_sum_fast_host(Z, word, alpha, lookup, extended, total_weighting,
                                          arctic=True)


class Bayesian(Semiring):
    """([0, 1], max, x): the letters and exponential weights of Reals, a running
    maximum instead of the cumulative sum and no shift between letters
    (fruits/iss/semiring.py:461-571).  Same HIP kernel (scan operator exchanged);
    the maximum is exact, the products round like the reference's."""

    def iterated_sum_fast(self, Z, word, alpha, lookup, extended, total_weighting):
        Z = _check_input(Z)
        word = np.asarray(word, dtype=np.int32)
        if lookup is not None and not np.any(lookup) and (alpha is None or not np.any(alpha)):
            lookup = None
        return nat.iterated_sum_fast_host(Z, word, alpha, lookup, extended, total_weighting,
                                          bayesian=True)
