from .core import *
from .core import (
    generate, load_checkpoint, save_audio, set_model, timer)
