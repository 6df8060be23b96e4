"""Scratch shared by all seeds of a fruit (mirrors fruits/cache.py:43-135).

Holds the RAW fruit input and memoises what several stages derive from it: the
cumulative L1 / L2 path length of dimension 0 (``CacheType.ISS`` - used by the
L1 / L2 weightings) and coquantile cut positions (``CacheType.COQUANTILE`` -
used by sieves with float cuts).  Everything is computed by the HIP kernel
``fr_pathlen_lookup`` and kept on the device; ``get`` hands out host copies for
API compatibility, ``get_device`` the device tensors.
"""
from __future__ import annotations

from enum import Enum, auto
from typing import Optional

import numpy as np

from . import _native as nat


class CacheType(Enum):
    COQUANTILE = auto()
    ISS = auto()


def _as3d(X):
    if X.ndim == 1:
        return X[np.newaxis, np.newaxis, :]
    if X.ndim == 2:
        return X[:, np.newaxis, :]
    return X


class SharedSeedCache:
    def __init__(self, X: Optional[np.ndarray] = None) -> None:
        self._input = None if X is None else _as3d(X)
        self._input_dev = None
        self._dev: dict = {CacheType.COQUANTILE: {}, CacheType.ISS: {}}
        self._host: dict = {CacheType.COQUANTILE: {}, CacheType.ISS: {}}

    # -- device side ------------------------------------------------------
    def input_device(self, X: Optional[np.ndarray] = None):
        """The raw input as a device tensor (uploaded once)."""
        if self._input_dev is not None:      # uploaded before, or adopted from the caller
            return self._input_dev
        if self._input is not None:
            self._input_dev = nat.to_device(self._input)
            return self._input_dev
        if X is None:
            raise RuntimeError("No input for cache given")
        return nat.to_device(_as3d(X))

    def adopt_device_input(self, Xd) -> None:
        """Lets the fruit share its already uploaded input with the cache."""
        self._input_dev = Xd

    def get_device(self, cache_id: CacheType, key: str, X: Optional[np.ndarray] = None):
        store = self._dev[cache_id]
        if store.get(key) is None:
            Xd = self.input_device(X)
            if cache_id == CacheType.ISS:
                norm = {"L1": 1, "L2": 2}[key]
                store[key] = nat.pathlen_lookup(Xd, norm=norm, relative=2)
            else:
                c, norm = key.split(":")
                path = self.get_device(CacheType.ISS, norm, X)
                # fruits/cache.py:16-22: number of t with path[t] <= q * path[-1]
                store[key] = (path <= float(c) * path[:, -1:]).sum(dim=1)
        return store[key]

    # -- reference API ----------------------------------------------------
    def get(self, cache_id: CacheType, key: str, X: Optional[np.ndarray] = None) -> np.ndarray:
        store = self._host[cache_id]
        if store.get(key) is None:
            arr = nat.to_host(self.get_device(cache_id, key, X))
            if cache_id == CacheType.COQUANTILE:
                arr = arr.astype(np.int64)
            store[key] = arr
        return store[key]
