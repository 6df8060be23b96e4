from . import semiring, weighting, words
from .cache import CachePlan
from .cos import CosWISS
from .iss import ISS, ISSMode
