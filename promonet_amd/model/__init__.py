from .core import get_padding
from .generator import Generator
from .hifigan import HiFiGAN
