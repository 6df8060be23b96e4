"""Increment sieves (mirrors IncrementSieve / NPI / MPI of
fruits/sieving/increment.py): statistics of the ``inc``-times differenced
iterated sum inside quantile bands ``q_k < x <= q_{k+1}``, per cut segment.
The differencing is fused into the HIP kernel's load (``fr_sieve``)."""
from __future__ import annotations

from collections.abc import Sequence
from typing import Literal, Optional, Union

import numpy as np

from .. import _native as nat
from .segment import SegmentSieve

__all__ = ["NPI", "MPI"]


class IncrementSieve(SegmentSieve):
    def __init__(self, cut: Union[Sequence[float], float] = -1,
                 q: Optional[Sequence[float]] = None, inc: int = 1,
                 coquantile_norm: Literal["L1", "L2"] = "L2") -> None:
        super().__init__(cut, q if q is not None else (0.0, 1.0), coquantile_norm)
        self._inc = inc

    def _pre_transform_device(self, Ad):
        """``inc`` > 0: increments applied ``inc`` times; < 0: cumulative sums
        (fruits/sieving/increment.py:63-71)."""
        if self._inc >= 0:
            return nat.pre_transform(Ad, self._inc)
        out = Ad
        plan = _cumsum_plan()
        for _ in range(-self._inc):
            out = plan.run(out.unsqueeze(1).contiguous(), None, layout="KNT")[0]
        return out

    def _pre_transform(self, X: np.ndarray) -> np.ndarray:
        return nat.to_host(self._pre_transform_device(nat.to_device(X)))

    def _fit(self, X: np.ndarray) -> None:
        # np.quantile over the whole pre-transformed fit sample (segment.py:66-75)
        super()._fit(self._pre_transform(X))

    def transform_device(self, Ad, out, col: int):
        if self._inc < 0:
            src, inc = self._pre_transform_device(Ad), 0
        else:
            src, inc = Ad, self._inc
        N, T = src.shape
        nat.sieve(self._kind, src, inc, self.cuts_device(N, T), self.quantiles_device(),
                  out, col)

    def _copy(self):
        # (Fruit.fit makes one copy per iterated sum and sieve - thousands: the state set
        # directly; like the reference's __class__(cut, q, inc) it forgets coquantile_norm)
        dup = object.__new__(self.__class__)
        dup._cut, dup._q, dup._inc, dup._coquantile_norm = self._cut, self._q, self._inc, "L2"
        return dup

    def __str__(self) -> str:
        return f"{self.__class__.__name__}({self._cut}, {self._q}, {self._inc})"

    def _label(self, index: int) -> str:
        label = super()._label(index)
        return label[:3] + f"[inc={self._inc}]" + label[3:]


_CUMSUM = None


def _cumsum_plan():
    """cumsum along time == the iterated sum of the word ``[1]``."""
    global _CUMSUM
    if _CUMSUM is None:
        _CUMSUM = nat.Plan([np.array([[1]], dtype=np.int32)], [1])
    return _CUMSUM


class NPI(IncrementSieve):
    """Number of (positive) increments inside each band
    (fruits/sieving/increment.py:101-129)."""
    _kind = nat.FR_SIEVE_NPI


class MPI(IncrementSieve):
    """Mean of the increments inside each band, 0 for an empty band
    (fruits/sieving/increment.py:132-163)."""
    _kind = nat.FR_SIEVE_MPI
