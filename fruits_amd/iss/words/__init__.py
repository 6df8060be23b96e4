from .creation import alternate_sign, of_weight, replace_letters
from .word import SimpleWord, Word
