"""Unit conversions (reference: promonet/convert.py)."""
import math

import torch

import promonet_amd


def db_to_ratio(db):
    """Decibels -> perceptual loudness ratio (convert.py:14-16)."""
    return 2 ** (db / 10)


def ratio_to_db(ratio):
    """Perceptual loudness ratio -> decibels (convert.py:19-24)."""
    if isinstance(ratio, torch.Tensor):
        return 10 * torch.log2(ratio)
    return 10 * math.log2(ratio)


def cents_to_ratio(cents):
    return 2 ** (cents / 1200)


def ratio_to_cents(ratio):
    return 1200 * math.log2(ratio)


def hz_to_bins(hz):
    """Pitch in Hz -> embedding bin (convert.py:69-91, variable-width bins)."""
    hz = torch.clip(hz, promonet_amd.FMIN, promonet_amd.FMAX)
    edges = promonet_amd.load.pitch_distribution().to(hz.device)
    return torch.clip(
        torch.searchsorted(edges, hz), 0, promonet_amd.PITCH_BINS - 1)


def seconds_to_frames(seconds):
    """convert.py:104-106"""
    return int(seconds * promonet_amd.SAMPLE_RATE / promonet_amd.HOPSIZE)


def frames_to_samples(frames):
    return frames * promonet_amd.HOPSIZE


def samples_to_seconds(samples, sample_rate=None):
    return samples / (sample_rate or promonet_amd.SAMPLE_RATE)


def frames_to_seconds(frames):
    return frames * samples_to_seconds(promonet_amd.HOPSIZE)


def samples_to_frames(samples):
    """convert.py:124-128"""
    return samples // promonet_amd.HOPSIZE
