"""Interpolation grids and grid sampling (reference: promonet/edit/grid.py)."""
import torch

from promonet_amd import _lib

LINEAR, LOG2, NEAREST = 0, 1, 2
INFINITY = float('inf')


def sample(sequence, grid, method='linear'):
    """Perform 1D grid-based sampling on the GPU (edit/grid.py:12-45).

    sequence (..., frames) -> (..., len(grid))"""
    if method not in ('linear', 'nearest'):
        raise ValueError(f'Grid sampling method {method} is not defined')
    return _sample(
        sequence, grid, LINEAR if method == 'linear' else NEAREST)


def _sample(sequence, grid, mode, scale=1., offset=0., lo=-INFINITY,
            hi=INFINITY):
    _lib.require_gpu(sequence)
    lib = _lib.lib()
    flat = sequence.to(torch.float32).contiguous()
    frames = flat.shape[-1]
    rows = flat.numel() // frames
    pointer, length = None, frames
    if grid is not None:
        grid = grid.to(device=flat.device, dtype=torch.float32).contiguous()
        pointer, length = _lib.ptr(grid), grid.numel()
    out = torch.empty(
        tuple(flat.shape[:-1]) + (length,), dtype=torch.float32,
        device=flat.device)
    with torch.cuda.device(flat.device):
        for start in range(0, rows, 65535):
            count = min(65535, rows - start)
            _lib.check(lib.pm_grid_sample(
                flat.data_ptr() + 4 * start * frames, pointer,
                out.data_ptr() + 4 * start * length, count, frames, length,
                mode, scale, offset, lo, hi, _lib.stream()))
    return out


def of_length(tensor, length):
    """Grid that resamples `tensor` to `length` frames. `ppgs.edit.grid.
    of_length` restated (third-party, absent: PARITY UNPINNED): `length`
    points evenly spaced over [0, frames - 1]."""
    return torch.linspace(
        0., tensor.shape[-1] - 1., int(length), dtype=torch.float32,
        device=tensor.device)


def constant(tensor, ratio):
    """Grid for constant-ratio time-stretching (> 1 is faster).
    `ppgs.edit.grid.constant` restated (PARITY UNPINNED)."""
    return of_length(tensor, round(tensor.shape[-1] / ratio + 1e-4))


def selective(ppg, ratio, indices):
    """Grid that time-stretches only where the phonemes `indices` (rows of
    `ppg` (P, frames)) carry the probability mass (edit/core.py:77-110): the
    reference's sequential recurrence, run on the device
    (`pm_stretch_grid`)."""
    _lib.require_gpu(ppg)
    lib = _lib.lib()
    ppg = ppg.to(torch.float32).contiguous()
    if ppg.ndim != 2:
        raise ValueError('selective grid: ppg must be (phonemes, frames)')
    frames = ppg.shape[-1]
    target = round(frames / ratio)
    indices = [int(index) for index in indices]
    if not indices or min(indices) < 0 or max(indices) >= ppg.shape[0]:
        raise ValueError(
            f'selective grid: phoneme rows {indices} outside the '
            f'{ppg.shape[0]}-row PPG')
    if target < 1:
        raise ValueError('selective grid: ratio leaves no output frame')
    rows = torch.tensor(indices, dtype=torch.int32, device=ppg.device)
    selected = torch.empty(frames, device=ppg.device)
    grid = torch.empty(target, device=ppg.device)
    with torch.cuda.device(ppg.device):
        _lib.check(lib.pm_stretch_grid(
            _lib.ptr(ppg), ppg.shape[0], _lib.ptr(rows, torch.int32),
            rows.numel(), _lib.ptr(selected), _lib.ptr(grid), frames, target,
            _lib.stream()))
    # (editing is not the throughput path: one sync buys a loud failure where
    # the reference's arithmetic silently yields inf / NaN / a grid that runs
    # backwards - no probability mass on the selected phonemes, or more
    # unselected mass than output frames. Non-DEcreasing is enough: a step
    # below one ulp of the position - a very large effective ratio on a long
    # grid - repeats a value, which the reference returns as well. Deviation
    # from the reference, deliberate: where its position runs past the last
    # frame it raises IndexError; the kernel clamps the read and the grid keeps
    # the formula's values.)
    if target > 1 and not bool(
            (torch.isfinite(grid).all() & (grid[1:] >= grid[:-1]).all() &
             (grid[-1] > grid[0]))):
        raise ValueError(
            'selective grid: the selected phonemes carry too little '
            'probability mass for this ratio (non-finite or decreasing grid)')
    return grid
