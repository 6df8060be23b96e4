from .abstract import FeatureSieve
from .segment import *
from .increment import *
