"""Edit speech features on the GPU (reference: promonet/edit/core.py).

The step right before `synthesize.from_features` in every usage example of
the reference (README.md:121-152): keeps the edit -> synthesize chain on the
device. SURVEY.md 8(f) item 1.
"""
from typing import Optional

import torch

import promonet_amd
from . import grid as _grid


def from_features(
    loudness: torch.Tensor,
    pitch: torch.Tensor,
    periodicity: torch.Tensor,
    ppg: torch.Tensor,
    pitch_shift_cents: Optional[float] = None,
    time_stretch_ratio: Optional[float] = None,
    loudness_scale_db: Optional[float] = None,
    stretch_unvoiced: bool = True,
    stretch_silence: bool = True,
    return_grid: bool = False
):
    """Apply pitch-shift / time-stretch / loudness edits to one utterance's
    features on the GPU. Same signature and return convention as the
    reference (edit/core.py:17-132).

    loudness (bands, T) dB, pitch (1, T) Hz, periodicity (1, T) and ppg
    (40, T) come back edited, in that order (plus the stretch grid when
    `return_grid`). `pitch_shift_cents` multiplies the pitch by 2^(cents /
    1200) and clips it to [FMIN, FMAX]; `time_stretch_ratio` > 1 shortens the
    utterance (all four features are resampled on one grid, the pitch in the
    log2 domain); `loudness_scale_db` is added to the loudness.
    `stretch_unvoiced=False` / `stretch_silence=False` leave unvoiced /
    silent frames at their original speed: the grid then steps by the
    probability mass of the remaining phonemes (edit/core.py:57-110).
    Inputs are never modified (the reference's `loudness +=` is in place).
    """
    grid = None
    if time_stretch_ratio is not None:
        if stretch_unvoiced and stretch_silence:
            grid = _grid.constant(ppg, time_stretch_ratio)
        else:
            grid = _grid.selective(
                ppg, time_stretch_ratio,
                stretched_phonemes(stretch_unvoiced, stretch_silence))
    ratio = 1. if pitch_shift_cents is None else \
        promonet_amd.convert.cents_to_ratio(pitch_shift_cents)
    offset = 0. if loudness_scale_db is None else float(loudness_scale_db)

    # One fused kernel per feature: resample (+ pitch ratio & clip, + dB offset)
    if grid is not None or pitch_shift_cents is not None:
        clip = pitch_shift_cents is not None
        pitch = _grid._sample(
            pitch, grid, _grid.LOG2 if grid is not None else _grid.LINEAR,
            scale=ratio,
            lo=promonet_amd.FMIN if clip else -_grid.INFINITY,
            hi=promonet_amd.FMAX if clip else _grid.INFINITY)
    if grid is not None:
        periodicity = _grid._sample(periodicity, grid, _grid.LINEAR)
        ppg = _grid.sample(ppg, grid, promonet_amd.PPG_INTERP_METHOD)
    if grid is not None or loudness_scale_db is not None:
        loudness = _grid._sample(loudness, grid, _grid.LINEAR, offset=offset)

    if return_grid:
        return loudness, pitch, periodicity, ppg, grid
    return loudness, pitch, periodicity, ppg


def stretched_phonemes(stretch_unvoiced, stretch_silence):
    """Rows of the PPG whose probability mass is time-stretched
    (edit/core.py:57-76): the voiced phonemes, plus silence and / or the
    unvoiced rest on request. The inventory is the third-party ppgs / pypar
    one, restated in promonet_amd.config (PHONEMES, VOICED, SILENCE)."""
    index = {p: i for i, p in enumerate(promonet_amd.PHONEMES)}
    indices = [index[p] for p in promonet_amd.VOICED]
    if stretch_silence:
        indices.append(index[promonet_amd.SILENCE])
    if stretch_unvoiced:
        indices.extend(
            index[p] for p in promonet_amd.PHONEMES
            if p not in promonet_amd.VOICED and p != promonet_amd.SILENCE)
    return indices


def from_file(
    loudness_file, pitch_file, periodicity_file, ppg_file,
    pitch_shift_cents=None, time_stretch_ratio=None, loudness_scale_db=None,
    stretch_unvoiced=True, stretch_silence=True, return_grid=False, gpu=0
):
    """Edit speech features on disk (edit/core.py:135-176)"""
    device = torch.device(f'cuda:{gpu}')
    pitch = torch.load(pitch_file).to(device)
    return from_features(
        torch.load(loudness_file).to(device), pitch,
        torch.load(periodicity_file).to(device),
        promonet_amd.load.ppg(ppg_file, pitch.shape[-1]).to(device),
        pitch_shift_cents, time_stretch_ratio, loudness_scale_db,
        stretch_unvoiced, stretch_silence, return_grid)


def from_file_to_file(
    loudness_file, pitch_file, periodicity_file, ppg_file, output_prefix,
    pitch_shift_cents=None, time_stretch_ratio=None, loudness_scale_db=None,
    stretch_unvoiced=True, stretch_silence=True, save_grid=False, gpu=0
):
    """Edit speech features on disk and save (edit/core.py:179-233)"""
    results = from_file(
        loudness_file, pitch_file, periodicity_file, ppg_file,
        pitch_shift_cents, time_stretch_ratio, loudness_scale_db,
        stretch_unvoiced, stretch_silence, save_grid, gpu)
    viterbi = '-viterbi' if promonet_amd.VITERBI_DECODE_PITCH else ''
    torch.save(results[0].cpu(), f'{output_prefix}-loudness.pt')
    torch.save(results[1].cpu(), f'{output_prefix}{viterbi}-pitch.pt')
    torch.save(results[2].cpu(), f'{output_prefix}{viterbi}-periodicity.pt')
    torch.save(results[3].cpu(), f'{output_prefix}-ppg.pt')
    if save_grid and results[4] is not None:
        torch.save(results[4].cpu(), f'{output_prefix}-grid.pt')
