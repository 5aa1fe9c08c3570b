from .core import get_padding
from .fargan import FARGAN
from .generator import Generator
from .hifigan import HiFiGAN
