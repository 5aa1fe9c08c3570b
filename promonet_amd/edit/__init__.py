from .core import *
from . import grid
