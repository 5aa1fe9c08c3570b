"""A-weighted loudness on the HIP kernels.

API of `promonet.preprocess.loudness` (promonet/preprocess/loudness.py); the
editing utilities `limit` / `scale` / `shift` (:114-193) are out of scope.
"""
import numpy as np
import torch

import promonet_amd
from promonet_amd import _lib


def from_audio(audio, bands=1):
    """A-weighted loudness (loudness.py:17-55), computed on the GPU.

    audio (1, N) -> (bands, N // 256); `bands=None` keeps the 513 bins.
    Also accepts (B, N) and returns (B, bands, T): every utterance gets its
    own max - 80 dB floor, as separate reference calls would.
    """
    lib = _lib.lib()
    _lib.require_gpu(audio)
    flat = audio.to(torch.float32).contiguous()
    batch, samples = flat.shape
    frames = samples // promonet_amd.HOPSIZE
    bins = promonet_amd.WINDOW_SIZE // 2 + 1
    nbands = bins if bands is None else int(bands)
    if nbands > 16 and nbands != bins:
        raise ValueError('at most 16 loudness bands (or None for all bins)')
    weights = perceptual_weights_tensor(flat.device)
    device = flat.device
    with torch.cuda.device(device):
        size = lib.pm_loudness_scratch_bytes(batch, samples)
        scratch = torch.empty(max(size, 1), dtype=torch.uint8, device=device)
        out = torch.empty(batch, nbands, frames, device=device)
        _lib.check(lib.pm_loudness(
            _lib.ptr(flat), _lib.ptr(weights), _lib.ptr(out), batch,
            samples, nbands, promonet_amd.MIN_DB, scratch.data_ptr(),
            scratch.numel(), _lib.stream()))
    return out[0] if batch == 1 else out


def from_file(audio_file, bands=None, gpu=0):
    """loudness.py:58-60"""
    bands = promonet_amd.LOUDNESS_BANDS if bands is None else bands
    return from_audio(
        promonet_amd.load.audio(audio_file).to(f'cuda:{gpu}'), bands)


def from_file_to_file(audio_file, output_file, bands=None, gpu=0):
    """loudness.py:63-65"""
    torch.save(from_file(audio_file, bands, gpu).cpu(), output_file)


def from_files_to_files(audio_files, output_files, bands=None, gpu=0):
    """loudness.py:68-76"""
    for audio_file, output_file in zip(audio_files, output_files):
        from_file_to_file(audio_file, output_file, bands, gpu)


###############################################################################
# Loudness utilities
###############################################################################


def band_average(loudness, bands=None):
    """Average over frequency bands (loudness.py:84-111). Index arithmetic
    on an existing tensor; the fused path is `from_audio(audio, bands)`."""
    bands = promonet_amd.LOUDNESS_BANDS if bands is None else bands
    if bands == 1:
        return loudness.mean(dim=-2, keepdim=True)
    step = loudness.shape[-2] / bands
    return torch.stack(
        [
            loudness[..., int(b * step):int((b + 1) * step), :].mean(dim=-2)
            for b in range(int(bands))
        ],
        dim=-2)


def normalize(loudness):
    """Normalize loudness to [-1., 1.] (loudness.py:144-146)"""
    return (loudness - promonet_amd.MIN_DB) / (
        promonet_amd.REF_DB - promonet_amd.MIN_DB)


def perceptual_weights():
    """A_weighting(fft_frequencies) - REF_DB, (513, 1) (loudness.py:149-160).
    `librosa.A_weighting` restated (IEC 61672, floor -80 dB): PARITY
    UNPINNED, librosa is absent from the build container."""
    freqs = np.linspace(
        0, promonet_amd.SAMPLE_RATE / 2, 1 + promonet_amd.WINDOW_SIZE // 2)
    f2 = freqs ** 2
    c = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2
    with np.errstate(divide='ignore'):
        weights = 2. + 20. * (
            np.log10(c[0]) + 2 * np.log10(f2) - np.log10(f2 + c[0]) -
            np.log10(f2 + c[1]) - .5 * np.log10(f2 + c[2]) -
            .5 * np.log10(f2 + c[3]))
    weights = np.maximum(-80., weights)
    return weights[:, None] - float(promonet_amd.REF_DB)


def perceptual_weights_tensor(device):
    cache = perceptual_weights_tensor.__dict__.setdefault('cache', {})
    if device not in cache:
        cache[device] = torch.from_numpy(
            perceptual_weights()[:, 0].astype(np.float32)).to(device)
    return cache[device]
