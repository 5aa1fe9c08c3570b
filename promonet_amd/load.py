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
    (promonet/load.py:16-28: torchaudio.load + torchaudio.functional.resample
    + channel mean). torchaudio is not a dependency here: wav files are read
    with scipy, integer PCM is scaled by 2^(bits - 1) as torchaudio.load
    (normalize=True) does, and `resample` below restates torchaudio's
    windowed-sinc resampler (parity unpinned: restated from its published
    algorithm)."""
    import scipy.io.wavfile
    rate, data = scipy.io.wavfile.read(file)
    if data.dtype.kind == 'i':
        data = data.astype(np.float32) / float(
            2 ** (8 * data.dtype.itemsize - 1))
    elif data.dtype.kind == 'u':
        data = (data.astype(np.float32) - 128.) / 128.
    data = torch.from_numpy(np.ascontiguousarray(data.astype(np.float32)))
    data = data[None] if data.ndim == 1 else data.T      # (channels, samples)
    data = resample(data, rate, promonet_amd.SAMPLE_RATE)
    return data.mean(dim=0, keepdim=True)


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6,
             rolloff=.99):
    """torchaudio.functional.resample(waveform, orig_freq, new_freq) with its
    defaults ('sinc_interp_hann', lowpass_filter_width 6, rolloff 0.99):
    a polyphase bank of Hann-windowed sinc filters applied as one strided
    conv1d. Host-side file loading only (CPU torch), not on the device path."""
    import math
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return waveform
    gcd = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // gcd, new_freq // gcd
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    index = torch.arange(
        -width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(
        0, -new, -1, dtype=torch.float64)[:, None, None] / new + index
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t)
    kernels = (kernels * window * (base / orig)).to(torch.float32)
    shape = waveform.shape
    flat = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = flat.shape[-1]
    padded = torch.nn.functional.pad(flat, (width, width + orig))
    out = torch.nn.functional.conv1d(padded[:, None], kernels, stride=orig)
    out = out.transpose(1, 2).reshape(flat.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return out[..., :target].reshape(shape[:-1] + (target,))


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
