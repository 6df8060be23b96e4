"""Segment sieves (mirrors SegmentSieve / END of fruits/sieving/segment.py).

A segment sieve looks at the pieces ``X[cut_j : cut_{j+1}]`` of an iterated
sum; ``cut`` entries are indices (negative = from the end) or floats in [0, 1]
that select a "coquantile" position of the raw input's path length.  Quantile
bands ``q`` are turned into value thresholds by ``fit`` (np.quantile over the
fit sample).  The counting / gathering itself is the HIP kernel ``fr_sieve``.
"""
from __future__ import annotations

from abc import ABC
from collections.abc import Sequence
from typing import Literal, Optional, Union

import numpy as np

from .. import _native as nat
from ..cache import CacheType
from .abstract import FeatureSieve

__all__ = ["END"]


class SegmentSieve(FeatureSieve, ABC):
    _kind = -1   # FR_SIEVE_* code
    _inc = 0

    def __init__(self, cut: Union[Sequence[float], float] = -1,
                 q: Optional[Sequence[float]] = None,
                 coquantile_norm: Literal["L1", "L2"] = "L2") -> None:
        self._cut = cut if isinstance(cut, Sequence) else (cut,)
        self._q = q if isinstance(q, Sequence) else (-1.0, 1.0)
        self._coquantile_norm = coquantile_norm

    @property
    def requires_fitting(self) -> bool:
        return any(q not in [-1, 0, 1] for q in self._q)

    # ---- cuts ---------------------------------------------------------------
    def _has_float_cuts(self) -> bool:
        return any(isinstance(c, float) for c in self._cut)

    def _int_cut_row(self, T: int) -> np.ndarray:
        row = [0] + [c if c >= 0 else T + c + 1 for c in self._cut]
        return np.sort(np.asarray(row, dtype=np.float64)).astype(np.int64)

    def _get_transformed_cuts(self, X: np.ndarray) -> np.ndarray:
        """(N, C+1) int64 boundaries, sorted, with a leading 0
        (fruits/sieving/segment.py:51-64)."""
        N, T = X.shape
        cuts = np.zeros((N, len(self._cut) + 1))
        for i, c in enumerate(self._cut):
            if isinstance(c, float):
                cuts[:, i + 1] = self._cache.get(
                    CacheType.COQUANTILE, str(c) + ":" + self._coquantile_norm)
            else:
                cuts[:, i + 1] = c if c >= 0 else T + c + 1
        return np.sort(cuts).astype(np.int64)

    def cuts_device(self, N: int, T: int):
        """Device cut table: one broadcast row for integer cuts, else (N, C+1)."""
        if not self._has_float_cuts():
            return nat.to_device(self._int_cut_row(T)[np.newaxis, :], dtype=np.int64)
        return nat.to_device(self._get_transformed_cuts(np.empty((N, T))), dtype=np.int64)

    # ---- quantiles ------------------------------------------------------------
    def _fit(self, X: np.ndarray) -> None:
        qs = np.zeros(len(self._q))
        for i, q in enumerate(self._q):
            if q == 1.0:
                qs[i] = np.inf
            elif q == -1.0:
                qs[i] = -np.inf
            elif q != 0:
                qs[i] = np.quantile(X, q)
        self._quantiles = np.sort(qs)

    # device-side fit: np.quantile's two order statistics come from fr_select_ranks
    def _quantile_requests(self, n: int):
        """[(index in q, lower rank, upper rank, gamma)] for the thresholds that need
        data, exactly as np.quantile(method="linear") places them: virtual index
        (n-1)*q, its floor and floor+1 (clipped), gamma = the fractional part."""
        reqs = []
        for i, q in enumerate(self._q):
            if q in (1.0, -1.0) or q == 0:
                continue
            pos = (n - 1) * q
            lo = int(np.floor(pos))
            gamma = pos - lo
            if pos >= n - 1:
                lo = hi = n - 1
            elif pos < 0:
                lo = hi = 0
            else:
                hi = lo + 1
            reqs.append((i, lo, hi, gamma))
        return reqs

    def _set_quantiles_from_stats(self, reqs, lo_vals, hi_vals) -> None:
        qs = [np.inf if q == 1.0 else (-np.inf if q == -1.0 else 0.0) for q in self._q]
        for (i, _, _, gamma), a, b in zip(reqs, lo_vals, hi_vals):
            # numpy's _lerp (lib/_function_base_impl.py): a + (b-a)*t, from the other end
            # when t >= 0.5
            d = b - a
            r = a + d * gamma
            if gamma >= 0.5:
                r = b - d * (1 - gamma)
            qs[i] = float(r)
        # (plain floats: thousands of sieve copies are fitted per Fruit.fit; np.sort's NaN-last
        # order only matters when a statistic is NaN)
        if any(v != v for v in qs):
            self._quantiles = np.sort(np.array(qs))
        else:
            qs.sort()
            self._quantiles = np.array(qs)

    def _get_unfitted_quantiles(self) -> None:
        qs = np.zeros(len(self._q))
        for i, q in enumerate(self._q):
            if q == 1.0:
                qs[i] = np.inf
            elif q == -1.0:
                qs[i] = -np.inf
            elif q != 0:
                raise RuntimeError("Sieve has not been fitted properly")
        self._quantiles = qs

    def quantiles_device(self):
        if not self.requires_fitting:
            self._get_unfitted_quantiles()
        return nat.to_device(np.asarray(self._quantiles, dtype=np.float64))

    # ---- transform ------------------------------------------------------------
    def transform_device(self, Ad, out, col: int):
        """Writes this sieve's features of the (N, T) device array ``Ad`` into
        columns [col, col + nfeatures) of the (N, F) device tensor ``out``."""
        N, T = Ad.shape
        nat.sieve(self._kind, Ad, self._inc, self.cuts_device(N, T),
                  self.quantiles_device(), out, col)

    def _transform(self, X: np.ndarray) -> np.ndarray:
        if not isinstance(X, np.ndarray) or X.dtype != np.float64 or X.ndim != 2:
            raise TypeError("input has to be a float64 array of shape (N, T)")
        t = nat.torch()
        Ad = nat.to_device(X)
        out = t.zeros((X.shape[0], self.nfeatures()), dtype=t.float64, device=Ad.device)
        self.transform_device(Ad, out, 0)
        return nat.to_host(out)

    def _nfeatures(self) -> int:
        return len(self._cut) * (len(self._q) - 1)

    def _copy(self):
        # (see IncrementSieve._copy; the reference's __class__(cut, q) forgets the norm too)
        dup = object.__new__(self.__class__)
        dup._cut, dup._q, dup._coquantile_norm = self._cut, self._q, "L2"
        return dup

    def __str__(self) -> str:
        return f"{self.__class__.__name__}({self._cut}, {self._q})"

    def _label(self, index: int) -> str:
        r, m = divmod(index, len(self._q) - 1)
        return (f"{self.__class__.__name__}"
                f"!{self._cut[r]}![{self._q[m]}, {self._q[m + 1]}]")

    def _summary(self) -> str:
        text = f"{self.__class__.__name__} -> {self.nfeatures()}:"
        for x in self._cut:
            text += f"\n   > {x}"
        return text


class END(SegmentSieve):
    """Last value of every segment, ``X[:, cut - 1]``
    (fruits/sieving/segment.py:203-225)."""
    _kind = nat.FR_SIEVE_END

    def quantiles_device(self):
        return None

    @staticmethod
    def _check_cuts(rows: np.ndarray, T: int) -> None:
        """The reference gathers ``X[:, cuts[:, 1:] - 1]`` with np.take_along_axis
        (fruits/sieving/segment.py:213-218) from the SORTED cut rows: an index
        outside [-T, T-1] raises IndexError there, and here."""
        idx = np.asarray(rows)[..., 1:] - 1
        bad = (idx < -T) | (idx > T - 1)
        if bad.any():
            raise IndexError(
                f"index {int(idx[bad].ravel()[0])} is out of bounds for axis 1 with size {T} "
                f"(END cut {self_cut_repr(rows)})")

    def _int_cut_row(self, T: int) -> np.ndarray:
        row = super()._int_cut_row(T)
        self._check_cuts(row, T)
        return row

    def _get_transformed_cuts(self, X: np.ndarray) -> np.ndarray:
        cuts = super()._get_transformed_cuts(X)
        self._check_cuts(cuts, X.shape[1])
        return cuts


def self_cut_repr(rows) -> str:
    return np.array2string(np.asarray(rows).reshape(-1, np.asarray(rows).shape[-1])[0])
