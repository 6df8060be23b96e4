from .abstract import Preparateur
from .transform import *
from .wrapper import *
