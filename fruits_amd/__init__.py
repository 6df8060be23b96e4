"""fruits_amd - the FRUITS iterated-sums hot path on AMD MI355X (gfx950).

Same public surface as the reference package for the path
``INC -> ISS (Reals, SimpleWords, optional Indices / L1 weighting) -> NPI / END``
(plus NEW, STD, MPI, the Arctic semiring and CosWISS): ``fruits_amd.ISS(...).fit_transform(X)`` and
``fruits_amd.Fruit.fit / transform``.  All arithmetic runs in hand-written HIP
kernels behind the C ABI of ``include/fruits_hip.h``; there is no CPU fallback.
"""
from . import cache, callback, iss, preparation, seed, sieving
from .fruit import Fruit, FruitSlice
from .iss import semiring, words
from .iss.cos import CosWISS
from .iss.iss import ISS, ISSMode

__version__ = "0.1.0"


def release_scratch() -> None:
    """Frees the device and page-locked scratch a device-side ``Fruit.fit`` keeps for the next
    one (fr_release_scratch); not in the reference."""
    from . import _native
    _native.release_scratch()
