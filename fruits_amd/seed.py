"""The stage plug-in interface (mirrors fruits/seed.py:11-82 of the reference).

Every stage of a fruit - preparateur, ISS, sieve - is a ``Seed``: public
``fit / transform / fit_transform / copy / label`` around the abstract
``_fit / _transform / _copy``.  A seed used on its own gets a temporary
:class:`~fruits_amd.cache.SharedSeedCache` for the duration of the call.

Device extension (not in the reference): seeds may implement
``_transform_device(Xd)`` that consumes and returns device tensors, which the
fruit uses to keep a whole pipeline in HBM.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .cache import SharedSeedCache


class Seed(ABC):
    _cache: SharedSeedCache
    # What a seed holds for the running process only - the cache it is attached to, device
    # programs - and what therefore is no part of its pickled state (a fitted fruit is handed
    # from the rank that fitted it to the others: fruits_amd.parallel.fit_on_root).
    _TRANSIENT = ("_cache", "_plans", "_programs")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def __setstate__(self, state):
        self.__dict__.update(state)
        if hasattr(self, "_fresh_transients"):
            self._fresh_transients()

    @property
    def requires_fitting(self) -> bool:
        return True

    @abstractmethod
    def _fit(self, X: np.ndarray) -> None:
        ...

    @abstractmethod
    def _transform(self, X: np.ndarray) -> np.ndarray:
        ...

    @abstractmethod
    def _copy(self):
        ...

    def _with_cache(self, X, fn):
        own = not hasattr(self, "_cache")
        if own:
            self._cache = SharedSeedCache(X[:, np.newaxis, :] if X.ndim == 2 else X)
        try:
            return fn(X)
        finally:
            if own:
                del self._cache

    def fit(self, X: np.ndarray) -> None:
        """Fits the seed to the given data."""
        self._with_cache(X, self._fit)

    def transform(self, X: np.ndarray) -> np.ndarray:
        """Transforms the given data and returns the results."""
        return self._with_cache(X, self._transform)

    def fit_transform(self, X: np.ndarray) -> np.ndarray:
        self.fit(X)
        return self.transform(X)

    def copy(self):
        """Returns a copy of this seed (fitted state is dropped)."""
        return self._copy()

    def _label(self, index: int = 0) -> str:
        return str(self)

    def label(self, index: int = 0) -> str:
        """Label of one transform this seed produces."""
        return self._label(index)

    def __str__(self) -> str:
        return self.__class__.__name__
