"""Second source for the restated librosa arithmetic:
`transformers.audio_utils`. Run by tests/test_cpu_oracle.py in a fresh
interpreter; exit code 0 = agree, 77 = transformers absent."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'oracle'))
sys.path.insert(0, str(ROOT))
import restatement as oracle  # noqa: E402

try:
    from transformers import audio_utils as au
except ImportError:
    sys.exit(77)


def check(name, error, bound):
    print(f'{name}: {error:.3e} (bound {bound:g})')
    if not error < bound:
        sys.exit(1)


# librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80) - Slaney scale, slaney
# norm (preprocess/spectrogram.py:118-121) - restated, and the product's copy
theirs = au.mel_filter_bank(
    513, 80, 0., 11025., 22050, norm='slaney', mel_scale='slaney').T
mine = oracle.mel_basis().double().numpy()
assert mine.shape == theirs.shape == (80, 513)
check('mel basis, oracle (peak 0.024)', np.abs(mine - theirs).max(), 1e-8)
import promonet_amd  # noqa: E402
ours = promonet_amd.preprocess.spectrogram.mel_basis().double().numpy()
check('mel basis, product', np.abs(ours - theirs).max(), 1e-8)

# the framed hann-1024 / hop-256 STFT of spectrogram.from_audio
# (spectrogram.py:15-60) over the reflect-padded signal
gen = torch.Generator().manual_seed(5)
audio = .1 * torch.randn(1, 256 * 48, generator=gen)
audio[:, 256 * 20:256 * 30] *= 1e-6              # a stretch below the floor
window = au.window_function(1024, 'hann', periodic=True)
padded = np.pad(audio[0].double().numpy(), (384, 384), mode='reflect')
stft = au.spectrogram(
    padded, window, 1024, 256, fft_length=1024, power=None, center=False,
    dtype=np.complex64)
assert stft.shape == (513, 48)
magnitude = np.sqrt(
    stft.real.astype(np.float64) ** 2 + stft.imag.astype(np.float64) ** 2 +
    1e-6)
mine = oracle.spectrogram(audio[None]).double().numpy()
check('stft magnitude, relative',
      np.abs(mine - magnitude).max() / magnitude.max(), 1e-5)

# loudness.from_audio's dB stage: librosa.stft + amplitude_to_db(ref=1,
# amin=1e-5, top_db=80) (loudness.py:38-46), then the A-weights (second
# sourced against IEC 61672 in test_cpu_oracle.py), the -100 floor, 8 bands
db = au.amplitude_to_db(
    np.abs(stft).astype(np.float64), reference=1., min_value=1e-5,
    db_range=80.)
assert abs(db.min() - (db.max() - 80.)) < 1e-9, 'the floor is not active'
weighted = np.maximum(db + oracle.perceptual_weights(), -100.)
want = oracle.band_average(torch.from_numpy(weighted).float(), 8)
got = oracle.loudness(audio, bands=8)
assert got.shape == want.shape == (8, 48)
check('loudness, dB', (got - want).abs().max().item(), 2e-4)
