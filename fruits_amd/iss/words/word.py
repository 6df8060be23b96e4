"""Words of the iterated-sums signature (mirrors fruits/iss/words/word.py).

Only :class:`SimpleWord` is on the MI355X hot path: a word whose letters pick
single input dimensions, stored as an integer exponent table that crosses the
C ABI as ``(L, Dw) int32``.  The generic :class:`Word` of named Python letter
functions (fruits/iss/words/letters.py) cannot cross a C ABI and is out of
scope; it exists here as the common base type only.
"""
from __future__ import annotations

import re
from typing import Optional, Sequence

import numpy as np

_SIMPLE_WORD = re.compile(r"(\[(-?\d|\(-?\d+\))+\])+")
_LETTER = re.compile(r"\((-?\d+)\)|(-?\d)")


class Word:
    """Base class: an ordered collection of extended letters with per-letter
    ``alpha`` values for weighted iterated sums."""

    def __init__(self, word_string: Optional[str] = None) -> None:
        self._extended_letters: list = []
        self._alpha: Optional[np.ndarray] = None
        self._cursor = -1
        if word_string is not None:
            self.multiply(word_string)

    @property
    def alpha(self) -> np.ndarray:
        """float32 ``(L,)``; all ones unless set (fruits/iss/words/word.py:71-82)."""
        if self._alpha is None:
            return np.ones((len(self),), dtype=np.float32)
        return self._alpha

    @alpha.setter
    def alpha(self, alpha: Sequence[float]) -> None:
        if len(alpha) != len(self):
            raise ValueError("Size of alpha array does not match word length")
        self._alpha = np.array(alpha, dtype=np.float32)

    def multiply(self, other) -> None:
        raise NotImplementedError(
            "generic words of named letter functions are not supported by the "
            "MI355X implementation; use SimpleWord")

    def copy(self) -> "Word":
        raise NotImplementedError

    def __len__(self) -> int:
        return len(self._extended_letters)

    def __iter__(self):
        self._cursor = -1
        return self

    def __next__(self):
        if self._cursor + 1 < len(self._extended_letters):
            self._cursor += 1
            return self._extended_letters[self._cursor]
        raise StopIteration()

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Word):
            raise NotImplementedError
        return False

    def __str__(self) -> str:
        return "".join(str(el) for el in self._extended_letters)


class SimpleWord(Word):
    """``SimpleWord("[11][122]")``: each bracket is an extended letter, each
    digit a (1-based) input dimension; ``(10)`` for multi-digit dimensions and a
    leading ``-`` for a reciprocal factor.  Stored as one row per extended
    letter holding the signed multiplicity of every dimension up to the highest
    one mentioned (fruits/iss/words/word.py:128-268)."""

    def __init__(self, string: str) -> None:
        super().__init__()
        self._max_dim = 0
        self._name = ""
        self.multiply(string)

    def multiply(self, other) -> None:
        if not isinstance(other, str):
            raise NotImplementedError
        if not _SIMPLE_WORD.fullmatch(other):
            raise ValueError("SimpleWord can only be multiplied with a "
                             "string matching the regular expression "
                             r"'(\[(-?\d|\(-?\d+\))+\])+'")
        self._name += other
        parsed = []
        for body in other[1:-1].split("]["):
            parsed.append([int(a or b) for a, b in _LETTER.findall(body)])
        widest = max(abs(x) for letters in parsed for x in letters)
        if widest > self._max_dim:
            pad = widest - self._max_dim
            for row in self._extended_letters:
                row.extend([0] * pad)
            self._max_dim = widest
        for letters in parsed:
            row = [0] * self._max_dim
            for x in letters:
                row[abs(x) - 1] += 1 if x > 0 else -1
            self._extended_letters.append(row)

    def table(self) -> np.ndarray:
        """The ``(L, Dw) int32`` exponent table handed to the device
        (``np.array(list(word), dtype=np.int32)``, fruits/iss/semiring.py:31)."""
        return np.array(self._extended_letters, dtype=np.int32).reshape(len(self), -1)

    def copy(self) -> "SimpleWord":
        dup = SimpleWord(self._name)
        dup._extended_letters = [list(row) for row in self._extended_letters]
        return dup

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, SimpleWord):
            raise NotImplementedError
        return list(self._extended_letters) == list(other._extended_letters)

    def __str__(self) -> str:
        return self._name
