"""Word generators (mirrors fruits/iss/words/creation.py)."""
from __future__ import annotations

import itertools
from collections.abc import Sequence

from .word import SimpleWord


def _compositions_seed(n: int, smallest: int = 1):
    """Partitions of n into non-decreasing parts, in the order the reference
    enumerates them (fruits/iss/words/creation.py:8-12)."""
    yield (n,)
    for first in range(smallest, n // 2 + 1):
        for rest in _compositions_seed(n - first, first):
            yield (first,) + rest


def _letters_of_weight(w: int, dim: int) -> list[str]:
    out = []
    for combo in itertools.combinations_with_replacement(range(1, dim + 1), w):
        out.append("[" + "".join(str(d) if d < 10 else f"({d})" for d in combo) + "]")
    return out


def of_weight(w: int, dim: int = 1) -> tuple[SimpleWord, ...]:
    """All words with exactly ``w`` letters over ``dim`` dimensions, in the
    reference's order (fruits/iss/words/creation.py:26-50).  The order of the
    permutations of one partition is CPython's set order, exactly as in the
    reference, so both enumerate identically on the same interpreter; the golden
    fixtures pin it."""
    by_weight = [_letters_of_weight(i, dim) for i in range(1, w + 1)]
    words = []
    for partition in _compositions_seed(w):
        for arrangement in set(itertools.permutations(partition)):
            for pieces in itertools.product(*(by_weight[k - 1] for k in arrangement)):
                words.append(SimpleWord("".join(pieces)))
    return tuple(words)


def alternate_sign(words: Sequence[SimpleWord]) -> list[SimpleWord]:
    """For every word two words whose extended letters alternate between plain
    and reciprocal factors (fruits/iss/words/creation.py:86-103)."""
    out = []
    for word in words:
        first, second = "", ""
        for i, row in enumerate(word):
            neg = "".join(f"-{d + 1}" * c for d, c in enumerate(row))
            pos = neg.replace("-", "")
            first += f"[{neg if i % 2 == 0 else pos}]"
            second += f"[{pos if i % 2 == 0 else neg}]"
        out.append(SimpleWord(first))
        out.append(SimpleWord(second))
    return out


def replace_letters(word, letter_gen):
    raise NotImplementedError(
        "named letter functions are outside the MI355X hot path (SimpleWord only)")
