"""Exponential time penalties g(n, t) for weighted iterated sums (mirrors
fruits/iss/weighting.py).

A summand that uses time steps i < j is scaled by ``exp(alpha * (g(i) - g(j)))``.
``get_lookup`` returns the ``(N, T)`` table g like the reference;
``lookup_device`` is what the pipeline uses: a device tensor with 1 row when g
does not depend on the series (Indices, Plateaus) or N rows (L1, L2, Custom).
L1 / L2 run on the device (HIP kernel ``fr_pathlen_lookup``); the series-
independent tables are O(T) host arithmetic.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Optional

import numpy as np

from .. import _native as nat
from ..cache import CacheType, SharedSeedCache

__all__ = ["Weighting", "L1", "L2", "Indices", "Plateaus", "Custom"]


def _minmax_rows(R: np.ndarray) -> np.ndarray:
    """NRM(scale_dim=False) on (rows, T): (x - min) / (max - min); constant
    rows give 0.  The result keeps R's dtype like the reference's
    ``np.zeros_like`` (fruits/preparation/transform.py:184-198), so an integer
    range is truncated to 0 / 1 there and here."""
    lo = R.min(axis=1, keepdims=True)
    hi = R.max(axis=1, keepdims=True)
    out = np.zeros_like(R)
    live = (lo != hi)[:, 0]
    out[live] = (R[live] - lo[live]) / (hi[live] - lo[live])
    return out


class Weighting(ABC):
    _cache: SharedSeedCache

    def __init__(self, total: bool = False) -> None:
        self.total = total

    def __getstate__(self):     # (the cache belongs to the running process: seed.py, _TRANSIENT)
        return {k: v for k, v in self.__dict__.items() if k != "_cache"}

    @abstractmethod
    def get_lookup(self, X: np.ndarray) -> np.ndarray:
        ...

    def lookup_device(self, Xd, exact: bool = True):
        """Device lookup for the prepared device input ``Xd`` (N, D, T).  ``exact``:
        bit-identical to the reference (needed where exact ties matter: max-plus
        results); False allows re-associated sums."""
        host = self.get_lookup(nat.to_host(Xd))
        return nat.to_device(np.asarray(host, dtype=np.float64))


class _SeriesIndependent(Weighting):
    def _row(self, T: int) -> np.ndarray:
        raise NotImplementedError

    def get_lookup(self, X: np.ndarray) -> np.ndarray:
        n, _, T = X.shape
        return np.ones((n, T)) * self._row(T)

    def lookup_device(self, Xd, exact: bool = True):
        row = np.asarray(self._row(int(Xd.shape[2])), dtype=np.float64)
        return nat.to_device(row[np.newaxis, :])


class Indices(_SeriesIndependent):
    """g(i) = i / T (or i), min-max normalised to [0, scale]
    (fruits/iss/weighting.py:69-110)."""

    def __init__(self, relative: bool = True,
                 transform: Optional[Callable[[float], float]] = None,
                 scale: float = 50, total: bool = False) -> None:
        super().__init__(total=total)
        self._relative = relative
        self._transform = transform
        self._scale = scale

    def _row(self, T: int) -> np.ndarray:
        steps = np.arange(1, T + 1)
        if self._relative:
            steps = steps / T
        if self._transform is not None:
            steps = np.vectorize(self._transform)(steps)
        return _minmax_rows(steps[np.newaxis, :])[0] * self._scale


class Plateaus(_SeriesIndependent):
    """g is an ascending (or descending) staircase of ``n`` plateaus
    (fruits/iss/weighting.py:213-256)."""

    def __init__(self, n: int, reverse: bool = False, scale: float = 50,
                 total: bool = False) -> None:
        super().__init__(total=total)
        if n <= 1:
            raise ValueError(f"Number of plateaus ({n}) has to be > 1")
        self._nplateaus = n
        self._reverse = reverse
        self._scale = scale

    def _row(self, T: int) -> np.ndarray:
        stairs = np.ones(T)
        width = int(T / self._nplateaus)
        for i in range(self._nplateaus):
            stairs[i * width:(i + 1) * width] = i / (self._nplateaus - 1)
        if self._reverse:
            stairs = stairs[::-1]
        return _minmax_rows(stairs[np.newaxis, :])[0] * self._scale


class _PathLength(Weighting):
    _norm = 1
    _key = "L1"

    def __init__(self, on_prepared: bool = False, relative: bool = False,
                 transform: Optional[Callable[[float], float]] = None,
                 scale: float = 50, total: bool = False) -> None:
        super().__init__(total=total)
        self._on_prepared = on_prepared
        self._relative = relative
        self._transform = transform
        self._scale = scale

    def lookup_device(self, Xd, exact: bool = True):
        if self._transform is not None:
            # an arbitrary Python callable cannot run on the device: apply it on
            # the host to the raw path length, then normalise
            raw = (nat.to_host(nat.pathlen_lookup(Xd, self._norm, 2)) if self._on_prepared
                   else self._cache.get(CacheType.ISS, self._key))
            r = raw / (raw[:, -1:] + 1e-5) if self._relative else raw
            r = np.vectorize(self._transform)(r)
            return nat.to_device(np.asarray(_minmax_rows(r) * self._scale, dtype=np.float64))
        src = Xd if self._on_prepared else self._cache.input_device()
        return nat.pathlen_lookup(src, self._norm, 1 if self._relative else 0,
                                  float(self._scale), exact=exact)

    def get_lookup(self, X: np.ndarray) -> np.ndarray:
        return nat.to_host(self.lookup_device(nat.to_device(X)))


class L1(_PathLength):
    """g(n, i) = sum of absolute increments of dimension 0 of the RAW input up
    to step i (fruits/iss/weighting.py:113-160, fruits/cache.py:25-31)."""
    _norm = 1
    _key = "L1"


class L2(_PathLength):
    """Same with squared increments (fruits/iss/weighting.py:163-210)."""
    _norm = 2
    _key = "L2"


class Custom(Weighting):
    """g = transform(X) supplied by the user, shape (N, T)
    (fruits/iss/weighting.py:41-66)."""

    def __init__(self, transform: Callable[[np.ndarray], np.ndarray],
                 total: bool = False) -> None:
        super().__init__(total=total)
        self._transform = transform

    def get_lookup(self, X: np.ndarray) -> np.ndarray:
        return self._transform(X)
