from . import semiring, weighting, words
from .cache import CachePlan
from .iss import ISS, ISSMode
