from .core import *
from . import loudness
from . import spectrogram
