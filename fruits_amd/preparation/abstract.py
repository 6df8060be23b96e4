"""Preparateur base class (mirrors fruits/preparation/abstract.py:9-20)."""
from abc import ABC
from typing import Any

import numpy as np

from .. import _native as nat
from ..seed import Seed


class Preparateur(Seed, ABC):
    """A preparateur maps ``(N, D, T)`` batches to ``(N, D', T')`` batches."""

    def _fit(self, X: np.ndarray) -> None:
        pass

    def _fit_needs_data(self) -> bool:
        """False when ``fit`` looks at nothing but the call itself: a fruit then does not
        download the (prepared) fit sample for it."""
        return type(self)._fit is not Preparateur._fit

    def _transform_device(self, Xd):
        """Device tensors in, device tensor out (never mutates ``Xd``)."""
        raise NotImplementedError

    def _transform(self, X: np.ndarray) -> np.ndarray:
        if not isinstance(X, np.ndarray) or X.dtype != np.float64 or X.ndim != 3:
            raise TypeError("input has to be a float64 array of shape (N, D, T)")
        return nat.to_host(self._transform_device(nat.to_device(X)))

    def __eq__(self, other: Any) -> bool:
        return False
