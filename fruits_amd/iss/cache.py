"""CachePlan: which trailing prefixes each word has to output in EXTENDED mode
(mirrors fruits/iss/cache.py:6-81).

A prefix counts as already produced only if an EARLIER word's raw string starts
with it - strings are compared un-canonicalised, so ``[12]`` and ``[21]`` are
different prefixes, exactly like the reference.  The device goes further and
also shares the WORK of common prefixes (csrc/plan.cpp); this class only decides
the outputs and their labels.
"""
from __future__ import annotations

from typing import Optional, Sequence


class CachePlan:
    def __init__(self, words: Sequence) -> None:
        self._words = words
        self._plan: list[int] = []
        self._create_plan()

    def _create_plan(self) -> None:
        strings = [str(w) for w in self._words]
        self._plan = []
        for i, s in enumerate(strings):
            letters = s.split("[")[1:]
            depth = len(letters)
            first_candidate = 0
            prefix = ""
            for letter in letters:
                prefix += "[" + letter
                hit = next((k for k in range(first_candidate, i)
                            if strings[k].startswith(prefix)), None)
                if hit is None:
                    break
                first_candidate = hit
                depth -= 1
            self._plan.append(depth)

    def unique_el_depth(self, index: int) -> int:
        """Number of iterated sums word ``index`` contributes."""
        return self._plan[index]

    def _locate(self, is_index: int) -> tuple[int, int]:
        for i, depth in enumerate(self._plan):
            if is_index < depth:
                return i, is_index - depth  # negative offset from the word's end
            is_index -= depth
        raise IndexError("Not enough iterated sums in cache plan")

    def get_word_index(self, is_index: int) -> int:
        return self._locate(is_index)[0]

    def get_word_string(self, is_index: int) -> str:
        """The (prefix of a) word that iterated sum ``is_index`` belongs to."""
        i, back = self._locate(is_index)
        return "]".join(str(self._words[i]).split("]")[:back]) + "]"

    def n_iterated_sums(self, word_indices: Optional[Sequence[int]] = None) -> int:
        if word_indices is None:
            return sum(self._plan)
        return sum(self._plan[i] for i in word_indices)
