"""Wrapping preparateurs (mirrors NEW of fruits/preparation/wrapper.py:53-103)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _native as nat
from .abstract import Preparateur

__all__ = ["NEW"]


class NEW(Preparateur):
    """Appends the output of another preparateur as new dimensions; with no
    preparateur given the input dimensions are duplicated."""

    def __init__(self, preparateur: Optional[Preparateur] = None) -> None:
        self._preparateur = preparateur

    @property
    def requires_fitting(self) -> bool:
        return False if self._preparateur is None else self._preparateur.requires_fitting

    def _fit_needs_data(self) -> bool:
        return self._preparateur is not None and self._preparateur._fit_needs_data()

    def _fit(self, X: np.ndarray) -> None:
        if self._preparateur is not None:
            self._preparateur.fit(X)

    def _transform_device(self, Xd):
        t = nat.torch()
        extra = Xd if self._preparateur is None else self._preparateur._transform_device(Xd)
        return t.cat((Xd, extra), dim=1).contiguous()

    def _copy(self) -> "NEW":
        return NEW() if self._preparateur is None else NEW(self._preparateur.copy())

    def __str__(self) -> str:
        return f"NEW({str(self._preparateur)})"
