"""Callback hooks of ``Fruit.transform`` (mirrors fruits/callback.py:6-41).

Supplying callbacks switches the fruit to its materialising path: hooks such as
``on_iterated_sum`` receive full host arrays, which the fused device path never
builds.
"""
from abc import ABC

import numpy as np


class AbstractCallback(ABC):
    def on_next_slice(self) -> None:
        """Called whenever the fruit moves on to its next slice."""

    def on_preparateur(self, X: np.ndarray) -> None:
        """Called with the prepared data after every preparateur."""

    def on_preparation_end(self, X: np.ndarray) -> None:
        """Called once with the fully prepared data."""

    def on_iterated_sum(self, X: np.ndarray) -> None:
        """Called with every (N, T) iterated sum."""

    def on_sieve(self, X: np.ndarray) -> None:
        """Called after each sieve."""

    def on_sieving_end(self, X: np.ndarray) -> None:
        """Called once with all features of the slice."""
