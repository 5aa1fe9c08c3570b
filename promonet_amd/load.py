"""Loading utilities on the hot path (reference: promonet/load.py)."""
import numpy as np
import torch

import promonet_amd


def pitch_distribution(dataset=None, partition='train'):
    """The 256 ascending pitch-bin edges (Hz) of the default configuration.

    Reference: promonet/load.py:54-111 reads
    `assets/stats/{dataset}-{PITCH_BINS}{-loudness}{-pitch}{-viterbi}.pt`;
    the same values ship here as a .npy (the reference recomputes them from
    the training set when the file is missing - out of scope).
    """
    if not hasattr(pitch_distribution, 'distribution'):
        dataset = dataset or promonet_amd.TRAINING_DATASET
        key = ''
        if promonet_amd.AUGMENT_LOUDNESS:
            key += '-loudness'
        if promonet_amd.AUGMENT_PITCH:
            key += '-pitch'
        if promonet_amd.VITERBI_DECODE_PITCH:
            key += '-viterbi'
        file = (
            promonet_amd.ASSETS_DIR / 'stats' /
            f'{dataset}-{promonet_amd.PITCH_BINS}{key}.npy')
        if not file.exists():
            raise FileNotFoundError(
                f'{file}: pitch statistics for this configuration are not '
                'bundled')
        pitch_distribution.distribution = torch.from_numpy(np.load(file))
    return pitch_distribution.distribution.clone()


def audio(file):
    """Load mono audio at SAMPLE_RATE as (1, samples) float32
    (promonet/load.py:16-28 uses torchaudio; wav via scipy here)."""
    import scipy.io.wavfile
    import scipy.signal
    rate, data = scipy.io.wavfile.read(file)
    if data.dtype.kind == 'i':
        data = data.astype(np.float32) / np.iinfo(data.dtype).max
    elif data.dtype.kind == 'u':
        data = (data.astype(np.float32) - 128.) / 128.
    data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if rate != promonet_amd.SAMPLE_RATE:
        gcd = np.gcd(rate, promonet_amd.SAMPLE_RATE)
        data = scipy.signal.resample_poly(
            data, promonet_amd.SAMPLE_RATE // gcd, rate // gcd
        ).astype(np.float32)
    return torch.from_numpy(data)[None]


def ppg(file, resample_length=None):
    """Load a PPG (40, T') and linearly resample to `resample_length` frames
    (promonet/load.py:172-188 delegates to ppgs.edit.grid; restated as
    align-corners-free linear interpolation on the frame grid)."""
    result = torch.load(file)
    if resample_length is not None and result.shape[-1] != resample_length:
        source = result.shape[-1]
        grid = torch.linspace(0., source - 1., resample_length)
        below = grid.floor().long().clamp(0, source - 1)
        above = (below + 1).clamp(0, source - 1)
        weight = (grid - below).to(result.dtype)
        result = (
            result[..., below] * (1 - weight) + result[..., above] * weight)
    return result


def features(prefix):
    """promonet/load.py:31-41"""
    viterbi = '-viterbi' if promonet_amd.VITERBI_DECODE_PITCH else ''
    return (
        torch.load(f'{prefix}-loudness.pt'),
        torch.load(f'{prefix}{viterbi}-pitch.pt'),
        torch.load(f'{prefix}{viterbi}-periodicity.pt'),
        torch.load(f'{prefix}-ppg.pt'))
