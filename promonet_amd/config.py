"""Import-time constants of the default configuration (`config/promonet.py`).

Mirrors the values the reference freezes at import through yapecs
(`promonet/config/defaults.py`, `promonet/config/static.py`); line numbers
refer to those files. Only the constants the synthesis hot path and its
preprocessing read are reproduced.
"""
import math
from pathlib import Path

CONFIG = 'promonet'

# Audio parameters (defaults.py:24-52)
DYNAMIC_RANGE_COMPRESSION_THRESHOLD = None
FMIN = 50.
FMAX = 550.
HOPSIZE = 256
MIN_DB = -100.
NUM_MELS = 80
NUM_FFT = 1024
REF_DB = 20.
SAMPLE_RATE = 22050
WINDOW_SIZE = 1024

# Data / feature parameters (defaults.py:60-135)
AUGMENT_LOUDNESS = True
AUGMENT_PITCH = True
LOUDNESS_BANDS = 8
PITCH_EMBEDDING = True
PITCH_BINS = 256
PITCH_EMBEDDING_SIZE = 64
PPG_CHANNELS = 40
PPG_INTERP_METHOD = 'linear'
SPARSE_PPG_METHOD = 'percentile'
SPARSE_PPG_THRESHOLD = 0.85
SPECTROGRAM_ONLY = False
TRAINING_DATASET = 'vctk'
VARIABLE_PITCH_BINS = True
VITERBI_DECODE_PITCH = True
INPUT_FEATURES = ['loudness', 'pitch', 'periodicity', 'ppg']

# Phoneme inventory of the PPGs (third-party `ppgs` package: PHONEMES, VOICED;
# `pypar.SILENCE`), read by the selective time-stretch of promonet.edit
# (edit/core.py:57-80). Restated from the published packages - parity unpinned.
PHONEMES = [
    'aa', 'ae', 'ah', 'ao', 'aw', 'ay', 'b', 'ch', 'd', 'dh', 'eh', 'er', 'ey',
    'f', 'g', 'hh', 'ih', 'iy', 'jh', 'k', 'l', 'm', 'n', 'ng', 'ow', 'oy', 'p',
    'r', 's', 'sh', 't', 'th', 'uh', 'uw', 'v', 'w', 'y', 'z', 'zh', '<silent>']
VOICED = [
    'aa', 'ae', 'ah', 'ao', 'aw', 'ay', 'b', 'd', 'dh', 'eh', 'er', 'ey', 'g',
    'ih', 'iy', 'jh', 'l', 'm', 'n', 'ng', 'ow', 'oy', 'r', 'uh', 'uw', 'v',
    'w', 'y', 'z', 'zh']
SILENCE = '<silent>'

# Model parameters (defaults.py:213-289)
LRELU_SLOPE = .1
MODEL = 'hifigan'
HIFIGAN_RESBLOCK_KERNEL_SIZES = [3, 7, 11]
HIFIGAN_RESBLOCK_DILATION_SIZES = [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
HIFIGAN_UPSAMPLE_INITIAL_SIZE = 512
HIFIGAN_UPSAMPLE_KERNEL_SIZES = [16, 16, 4, 4]
HIFIGAN_UPSAMPLE_RATES = [8, 8, 2, 2]
SPEAKER_CHANNELS = 256
WAVLM_EMBEDDING_CHANNELS = 512
ZERO_SHOT = False
STEPS = 800000
NUM_WORKERS = 10

# Storage type of the streamed FARGAN weights (math is fp32): 'fp32', 'f16'
# (6.6e-5 max-abs against the reference on random-init weights, over a 10 s
# utterance) or 'mixed' - the GRU cells and the GLU gates (80 % of the stream,
# all in front of a sigmoid / tanh) as f16, the conditioning network,
# framewise conv, skip dense and output layer as fp32: 6e-6
# (scripts/fargan_weight_sensitivity.py, tests/test_gpu_fargan.py)
FARGAN_WEIGHT_DTYPE = 'fp32'
FARGAN_PREVIOUS_FRAMES = 2
FARGAN_SUBFRAMES = 4

# MFMA operand type of the HIP engine (accumulation and the activations
# between kernels are always fp32). Max-abs error against the fp32 reference
# (tests/test_gpu_model.py, scripts/precision_sweep.py) at
# the random-init output scale (audio peak 0.017: what BASELINE.json's 1e-4
# gate is stated on) | with the output conv rescaled so that the audio peaks at
# 0.5 | at 0.99, a trained checkpoint's scale, and the batch-32 x 10 s step:
#   'checkpoint'      5e-7   | 1.5e-5 | 5.7e-5   20.1 ms  (DEFAULT) f16; in the last
#                     upsampling stage the ACTIVATIONS split into hi + lo ('f16a2':
#                     two MFMAs per step in its Blocks, its upsampler fully split),
#                     the upsampler of the stage before it fully split too ('f16ux'):
#                     the mode that holds 1e-4 at a real checkpoint's output scale.
#                     Spelled out for the default 4-stage model: 'f16+f16+f16ux+f16a2'
#   'f16+f16+f16+f16x3'  4e-7 | 1.3e-5 | 5.2e-5   21.3 ms  the last stage fully split
#                     (hi + lo of both operands, three MFMAs per step): the default
#                     until round 6
#   'f16'             3.1e-6 | 7.2e-5 | 3.9e-4   18.4 ms  operands saturate at 65504
#   'bf16'            3.0e-5 | 7.7e-4 |   -      17.6 ms  what BASELINE.json config 3
#                     names and bench.py asks for; fp32's exponent range
#   'fp32'            5.4e-8 | 2.0e-6 | 8.7e-6   ~145 ms  exact-fp32 MFMA, 1/16 rate
#   'f16x3'           split f16 everywhere (fp32-like, ~3x the f16 step)
#   one type per upsampling stage joined by '+', e.g. 'bf16+bf16+bf16+f16'
#                     (1.2e-4 at peak 0.5, at bf16's speed)
# One f16 rounding of the last stage's activations alone is 3e-4 of an output
# that peaks near 1, so no single-MFMA 16-bit mode holds 1e-4 there.
DEFAULT_COMPUTE_DTYPE = 'checkpoint'
COMPUTE_DTYPE = DEFAULT_COMPUTE_DTYPE

# Validate speaker ids that arrive as DEVICE tensors (one sync per call); off:
# an id outside the embedding table yields NaN audio instead of IndexError
CHECK_DEVICE_SPEAKERS = False

ASSETS_DIR = Path(__file__).parent / 'assets'


def derived():
    """Constants of config/static.py, recomputed from the values above."""
    g = globals()
    g['LOG_DYNAMIC_RANGE_COMPRESSION_THRESHOLD'] = (
        None if DYNAMIC_RANGE_COMPRESSION_THRESHOLD is None else
        math.log(DYNAMIC_RANGE_COMPRESSION_THRESHOLD))          # static.py:12-14
    g['LOG_FMIN'] = math.log2(FMIN)                             # static.py:17
    g['LOG_FMAX'] = math.log2(FMAX)
    g['GLOBAL_CHANNELS'] = (
        SPEAKER_CHANNELS + AUGMENT_PITCH + AUGMENT_LOUDNESS)    # static.py:42-45
    g['NUM_FEATURES'] = NUM_MELS if SPECTROGRAM_ONLY else (
        PPG_CHANNELS +
        ('loudness' in INPUT_FEATURES) * LOUDNESS_BANDS +
        ('periodicity' in INPUT_FEATURES) +
        ('pitch' in INPUT_FEATURES) * (
            PITCH_EMBEDDING_SIZE if PITCH_EMBEDDING else 1))    # static.py:48-53
    g['NUM_SPEAKERS'] = {
        'daps': 20, 'libritts': 1230, 'vctk': 109}[TRAINING_DATASET]
    g['NUM_PREVIOUS_SAMPLES'] = (
        HOPSIZE * FARGAN_PREVIOUS_FRAMES if MODEL == 'fargan' else 1)  # static.py:69-74


derived()


# every override applied so far (a `spawn`ed worker process imports the package
# with the defaults above: synthesize.core re-applies these in its workers)
OVERRIDES = {}


def configure(**overrides):
    """Override constants BEFORE constructing models (the reference does this
    once, at import, from `--config` files: promonet/__init__.py:7-15)."""
    import promonet_amd
    for key, value in overrides.items():
        if key not in globals() or key == 'OVERRIDES':
            raise ValueError(f'Unknown configuration parameter {key}')
        globals()[key] = value
        OVERRIDES[key] = value
    derived()
    for key, value in globals().items():
        if key.isupper():
            setattr(promonet_amd, key, value)
