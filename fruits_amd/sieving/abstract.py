"""Feature sieve base class (mirrors fruits/sieving/abstract.py:8-34)."""
from abc import ABC, abstractmethod

import numpy as np

from ..seed import Seed


class FeatureSieve(Seed, ABC):
    """A sieve maps an ``(N, T)`` iterated sum to ``(N, F)`` features."""

    @abstractmethod
    def _nfeatures(self) -> int:
        ...

    def nfeatures(self) -> int:
        return self._nfeatures()

    def _fit(self, X: np.ndarray) -> None:
        pass

    @abstractmethod
    def _summary(self) -> str:
        ...

    def summary(self) -> str:
        return self._summary()
