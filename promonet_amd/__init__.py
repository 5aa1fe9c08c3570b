"""promonet_amd: promonet's synthesis hot path, native to AMD MI355X.

`promonet.synthesize.from_features` -> `promonet.model.Generator.forward`
and the STFT / mel / loudness preprocessing around it, behind the
reference's Python API. Host code is Python on PyTorch-ROCm (device memory,
streams, torch.distributed); all arithmetic runs in hand-written HIP kernels
(`libpromonet_hip.so`, C ABI in include/promonet_hip.h).
"""
from .config import *                                          # noqa: F401
from .config import configure, derived                         # noqa: F401
from .config import (                                          # noqa: F401
    GLOBAL_CHANNELS, LOG_DYNAMIC_RANGE_COMPRESSION_THRESHOLD, LOG_FMAX,
    LOG_FMIN, NUM_FEATURES, NUM_PREVIOUS_SAMPLES, NUM_SPEAKERS)
from . import _lib                                             # noqa: F401
from . import convert                                          # noqa: F401
from . import edit                                             # noqa: F401
from . import load                                             # noqa: F401
from . import model                                            # noqa: F401
from . import preprocess                                       # noqa: F401
from . import synthesize                                       # noqa: F401
from . import distributed                                      # noqa: F401
from .patch import patch                                       # noqa: F401
