"""Spectrogram / mel preprocessing on the HIP kernels.

API of `promonet.preprocess.spectrogram` (promonet/preprocess/spectrogram.py).
"""
import functools
import multiprocessing as mp

import numpy as np
import torch

import promonet_amd
from promonet_amd import _lib


def from_audio(
    audio,
    mels=False,
    log_dynamic_range_compression_threshold=None
):
    """Magnitude (or log-mel) spectrogram (spectrogram.py:15-60).

    audio (B, 1, N) or (1, N) on the GPU -> (B, 513|80, N // 256), with the
    batch axis squeezed when B == 1, exactly as the reference returns it.
    """
    if log_dynamic_range_compression_threshold is None:
        log_dynamic_range_compression_threshold = \
            promonet_amd.LOG_DYNAMIC_RANGE_COMPRESSION_THRESHOLD
    _lib.require_gpu(audio)
    flat = audio.squeeze(1) if audio.ndim == 3 else audio
    flat = flat.to(torch.float32).contiguous()
    if torch.is_grad_enabled() and flat.requires_grad:
        # training-side use (the mel loss, train/core.py:277-305): keep the
        # graph through the HIP kernels
        out = _Magnitude.apply(flat)
    elif mels:
        # inference: log-mel straight out of the FFT kernel, the (B, 513, T)
        # magnitudes never reach HBM
        return _log_mel(
            flat, log_dynamic_range_compression_threshold).squeeze(0)
    else:
        out = _magnitude(flat)
    if mels:
        out = linear_to_mel(out, log_dynamic_range_compression_threshold)
    return out.squeeze(0)


def _log_mel(flat, threshold):
    lib = _lib.lib()
    batch, samples = flat.shape
    frames = samples // promonet_amd.HOPSIZE
    mels = promonet_amd.NUM_MELS
    out = torch.empty(batch, mels, frames, device=flat.device)
    with torch.cuda.device(flat.device):
        _lib.check(lib.pm_stft_mel(
            _lib.ptr(flat), _prepared_mel_basis(flat.device).data_ptr(),
            _lib.ptr(out), batch, samples, mels, int(threshold is not None),
            float(threshold or 0.), _lib.stream()))
    return out


def _prepared_mel_basis(device):
    """The mel filterbank in the compact form the fused kernel reads, built
    once per device (and per configuration)."""
    key = (device, promonet_amd.SAMPLE_RATE, promonet_amd.NUM_FFT,
           promonet_amd.NUM_MELS)
    cache = _prepared_mel_basis.__dict__.setdefault('cache', {})
    if key not in cache:
        lib = _lib.lib()
        basis = mel_basis().to(device).contiguous()
        with torch.cuda.device(device):
            size = lib.pm_stft_mel_scratch_bytes(basis.shape[0])
            prepared = torch.empty(size, dtype=torch.uint8, device=device)
            _lib.check(lib.pm_stft_mel_prepare(
                _lib.ptr(basis), basis.shape[0], prepared.data_ptr(),
                prepared.numel(), _lib.stream()))
            # the buffer is cached for every later caller, whatever stream
            # (or graph capture) it runs on: finish building it first
            torch.cuda.current_stream(device).synchronize()
        cache[key] = prepared
    return cache[key]


def _magnitude(flat):
    lib = _lib.lib()
    batch, samples = flat.shape
    frames = samples // promonet_amd.HOPSIZE
    bins = promonet_amd.NUM_FFT // 2 + 1
    out = torch.empty(batch, bins, frames, device=flat.device)
    with torch.cuda.device(flat.device):
        # (a real FFT per frame in LDS: no scratch)
        _lib.check(lib.pm_stft_magnitude(
            _lib.ptr(flat), _lib.ptr(out), batch, samples, None, 0,
            _lib.stream()))
    return out


class _Magnitude(torch.autograd.Function):
    """spectrogram magnitude with a HIP backward: the framed DFT is a linear
    map, so its adjoint is the overlap-add of grad / |X| * (re, im) against
    the transposed windowed basis (pm_stft_magnitude_backward)."""

    @staticmethod
    def forward(ctx, flat):
        ctx.save_for_backward(flat)
        return _magnitude(flat)

    @staticmethod
    def backward(ctx, grad):
        flat, = ctx.saved_tensors
        lib = _lib.lib()
        grad = grad.to(torch.float32).contiguous()
        batch, samples = flat.shape
        result = torch.empty_like(flat)
        with torch.cuda.device(flat.device):
            size = lib.pm_stft_backward_scratch_bytes(batch, samples)
            scratch = torch.empty(
                max(size, 1), dtype=torch.uint8, device=flat.device)
            _lib.check(lib.pm_stft_magnitude_backward(
                _lib.ptr(flat), _lib.ptr(grad), _lib.ptr(result), batch,
                samples, scratch.data_ptr(), scratch.numel(), _lib.stream()))
        return result


def from_file(
    audio_file,
    mels=False,
    log_dynamic_range_compression_threshold=None,
    gpu=0
):
    """spectrogram.py:63-71 (the reference computes on CPU; here on `gpu`)"""
    audio = promonet_amd.load.audio(audio_file).to(f'cuda:{gpu}')
    return from_audio(audio, mels, log_dynamic_range_compression_threshold)


def from_file_to_file(
    audio_file,
    output_file,
    mels=False,
    log_dynamic_range_compression_threshold=None,
    gpu=0
):
    """spectrogram.py:74-86"""
    output = from_file(
        audio_file, mels, log_dynamic_range_compression_threshold, gpu)
    torch.save(output.cpu(), output_file)


def from_files_to_files(
    audio_files,
    output_files,
    mels=False,
    log_dynamic_range_compression_threshold=None,
    gpu=0
):
    """spectrogram.py:89-103 forks NUM_WORKERS CPU processes; one GPU stream
    is faster than that pool, so this is a loop."""
    for audio_file, output_file in zip(audio_files, output_files):
        from_file_to_file(
            audio_file, output_file, mels,
            log_dynamic_range_compression_threshold, gpu)


###############################################################################
# Utilities
###############################################################################


def mel_basis():
    """`librosa.filters.mel(sr=22050, n_fft=1024, n_mels=80)` restated
    (Slaney scale, slaney norm; called at spectrogram.py:118-121).
    PARITY UNPINNED: librosa is absent from the build container."""
    # (keyed on the configuration: configure(NUM_MELS=...) after a first use
    # must not pair a stale basis with the new filter count)
    key = (promonet_amd.SAMPLE_RATE, promonet_amd.NUM_FFT,
           promonet_amd.NUM_MELS)
    cache = mel_basis.__dict__.setdefault('cache', {})
    if key not in cache:
        sr, n_fft, n_mels = key

        def to_mel(f):
            f = np.asarray(f, dtype=np.float64)
            linear = f / (200. / 3)
            log = 15. + np.log(np.maximum(f, 1e-30) / 1000.) / (np.log(6.4) / 27.)
            return np.where(f >= 1000., log, linear)

        def to_hz(m):
            m = np.asarray(m, dtype=np.float64)
            return np.where(
                m >= 15., 1000. * np.exp((np.log(6.4) / 27.) * (m - 15.)),
                m * (200. / 3))

        freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
        edges = to_hz(np.linspace(to_mel(0.), to_mel(sr / 2), n_mels + 2))
        widths = np.diff(edges)
        ramps = edges[:, None] - freqs[None]
        lower = -ramps[:-2] / widths[:-1, None]
        upper = ramps[2:] / widths[1:, None]
        weights = np.maximum(0, np.minimum(lower, upper))
        weights *= (2. / (edges[2:] - edges[:-2]))[:, None]
        cache[key] = torch.from_numpy(weights.astype(np.float32))
    return cache[key]


def linear_to_mel(spectrogram, log_dynamic_range_compression_threshold=None):
    """log(mel_basis @ spectrogram), optional clamp (spectrogram.py:111-133).
    The basis is cached per device (the reference rebuilds it on every call
    through a misnamed cache attribute, :117 vs :124)."""
    _lib.require_gpu(spectrogram)
    squeeze = spectrogram.ndim == 2
    spec = (spectrogram[None] if squeeze else spectrogram).to(
        torch.float32).contiguous()
    basis = mel_basis().to(spec.device).contiguous()
    threshold = log_dynamic_range_compression_threshold
    if torch.is_grad_enabled() and spec.requires_grad:
        out = _LinearToMel.apply(spec, basis, threshold)
    else:
        out = _linear_to_mel(spec, basis, threshold)
    return out[0] if squeeze else out


def _linear_to_mel(spec, basis, threshold):
    lib = _lib.lib()
    batch, bins, frames = spec.shape
    out = torch.empty(batch, basis.shape[0], frames, device=spec.device)
    with torch.cuda.device(spec.device):
        _lib.check(lib.pm_linear_to_mel(
            _lib.ptr(spec), _lib.ptr(basis), _lib.ptr(out), batch, bins,
            basis.shape[0], frames, int(threshold is not None),
            float(threshold or 0.), _lib.stream()))
    return out


class _LinearToMel(torch.autograd.Function):
    """log(basis @ spec) (clamped) with a HIP backward
    (pm_linear_to_mel_backward); the clamp passes gradient where the forward
    value is >= the threshold, as torch.clamp does."""

    @staticmethod
    def forward(ctx, spec, basis, threshold):
        ctx.save_for_backward(spec, basis)
        ctx.threshold = threshold
        return _linear_to_mel(spec, basis, threshold)

    @staticmethod
    def backward(ctx, grad):
        spec, basis = ctx.saved_tensors
        lib = _lib.lib()
        grad = grad.to(torch.float32).contiguous()
        batch, bins, frames = spec.shape
        result = torch.empty_like(spec)
        scratch = torch.empty(
            batch, basis.shape[0], frames, device=spec.device)
        threshold = ctx.threshold
        with torch.cuda.device(spec.device):
            _lib.check(lib.pm_linear_to_mel_backward(
                _lib.ptr(spec), _lib.ptr(basis), _lib.ptr(grad),
                _lib.ptr(result), _lib.ptr(scratch), batch, bins,
                basis.shape[0], frames, int(threshold is not None),
                float(threshold or 0.), _lib.stream()))
        return result, None, None
