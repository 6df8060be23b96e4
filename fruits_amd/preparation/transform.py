"""Preparateurs on the MI355X path (mirrors the INC / STD part of
fruits/preparation/transform.py).  Each runs as a HIP kernel on device
tensors; the numpy-facing ``transform`` uploads, runs and downloads."""
from __future__ import annotations

from typing import Any, Callable, Union

import numpy as np

from .. import _native as nat
from .abstract import Preparateur

__all__ = ["INC", "STD"]


class INC(Preparateur):
    """Increments: ``[x_1, ..., x_n] -> [0, x_2 - x_1, ..., x_n - x_{n-1}]``
    (fruits/preparation/transform.py:15-89; kernel: fruits/cache.py:8-13).

    Args:
        shift: lag of the difference; a float is a fraction of the series
            length (rounded up), a callable maps the length to the lag.
        depth: how many times the transform is applied.
        zero_padding: if False the first ``shift`` values are restored from
            the input instead of being zero.
    """

    def __init__(self, shift: Union[int, float, Callable[[int], int]] = 1,
                 depth: int = 1, zero_padding: bool = True) -> None:
        self._shift = shift
        if depth < 1:
            raise ValueError("depth has to be a positive integer > 0")
        self._depth = depth
        self._zero_padding = zero_padding

    @property
    def requires_fitting(self) -> bool:
        return False

    def _lag(self, T: int) -> int:
        if isinstance(self._shift, int):
            return self._shift
        if isinstance(self._shift, float):
            return int(np.ceil(self._shift * T))
        if callable(self._shift):
            return int(self._shift(T))
        raise TypeError(f"Type {type(self._shift)} not supported for argument shift")

    def _transform_device(self, Xd):
        lag = self._lag(int(Xd.shape[2]))
        out = Xd
        for _ in range(self._depth):
            if self._zero_padding:
                out = nat.increments(out, lag)
            else:
                # the reference restores X[:, :, :shift] after every pass
                # (transform.py:73-74); it needs an integer shift there too
                out = nat.increments(out, lag, head_src=Xd, head=int(self._shift))
        return out

    def _copy(self) -> "INC":
        return INC(self._shift, self._depth, self._zero_padding)

    def __eq__(self, other: Any) -> bool:
        return (isinstance(other, INC) and self._shift == other._shift
                and self._depth == other._depth
                and self._zero_padding == other._zero_padding)

    def __str__(self) -> str:
        return f"INC({self._shift}, {self._depth}, {self._zero_padding})"


class STD(Preparateur):
    """Standardisation ``(x - mean) / (std + eps)`` per series and dimension
    (fruits/preparation/transform.py:92-158).  ``separately=False`` uses one
    mean / std of the whole fit sample."""

    def __init__(self, separately: bool = True, var: bool = True,
                 std_eps: float = 1e-5) -> None:
        self._separately = separately
        self._div_std = var
        self._mean = None
        self._std = None
        self._eps = std_eps

    def _fit_needs_data(self) -> bool:
        return not self._separately

    def _fit(self, X: np.ndarray) -> None:
        if not self._separately:
            self._mean = np.mean(X)
            self._std = np.std(X) if self._div_std else 1

    def _transform_device(self, Xd):
        if self._separately:
            return nat.standardize(Xd, self._div_std, float(self._eps))
        if self._mean is None or self._std is None:
            raise RuntimeError("Missing call of self.fit()")
        return (Xd - float(self._mean)) / (float(self._std) + float(self._eps))

    def _copy(self) -> "STD":
        return STD(self._separately, self._div_std)

    def __eq__(self, other: Any) -> bool:
        return (isinstance(other, STD) and self._separately == other._separately
                and self._div_std == other._div_std)

    def __str__(self) -> str:
        return f"STD({self._separately}, {self._div_std})"
