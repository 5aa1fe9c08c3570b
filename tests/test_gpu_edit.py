"""GPU parity of promonet_amd.edit against goldens computed by the real
reference's `promonet.edit.from_features` / `promonet.edit.grid.sample`
(the grid CONSTRUCTOR is third-party ppgs: stored in the golden)."""
import pytest
import torch

import restatement as oracle
from util import max_abs

pytestmark = pytest.mark.gpu


def test_grid_sample_golden(device, golden_default):
    import promonet_amd
    entry = golden_default['edit']['sample']
    sequence, grid = entry['sequence'].to(device), entry['grid'].to(device)
    assert max_abs(
        promonet_amd.edit.grid.sample(sequence, grid), entry['linear']) < 1e-5
    assert max_abs(
        promonet_amd.edit.grid.sample(sequence, grid, 'nearest'),
        entry['nearest']) == 0.
    with pytest.raises(ValueError):
        promonet_amd.edit.grid.sample(sequence, grid, 'cubic')


def test_from_features_golden(device, golden_default):
    import promonet_amd
    entry = golden_default['edit']
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    loud, pit, per, pg = (
        inputs[0][0].to(device), inputs[1].to(device), inputs[2].to(device),
        inputs[3][0].to(device))
    for case in entry['cases']:
        got = promonet_amd.edit.from_features(
            loud, pit, per, pg, case['pitch_shift_cents'],
            case['time_stretch_ratio'], case['loudness_scale_db'],
            return_grid=True)
        if case['grid'] is not None:
            assert max_abs(got[4], case['grid']) < 1e-5
        for mine, want, tolerance in zip(
                got[:4], case['outputs'], (2e-4, 2e-3, 1e-5, 1e-5)):
            assert mine.shape == want.shape
            assert max_abs(mine, want) < tolerance, case
    # inputs are never mutated (the reference's `loudness +=` is in place)
    assert torch.equal(loud.cpu(), inputs[0][0])


def test_selective_time_stretch_golden(device):
    """stretch_unvoiced / stretch_silence off (edit/core.py:57-110): the grid
    is a sequential fp32 recurrence over the selected phonemes' probability;
    golden from the REAL reference loop (phoneme inventory restated)."""
    import promonet_amd
    from conftest import GOLDEN
    entry = torch.load(GOLDEN / 'edit_voiced.pt', weights_only=False)
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    loud, pit, per, pg = (
        inputs[0][0].to(device), inputs[1].to(device), inputs[2].to(device),
        inputs[3][0].to(device))
    for case in entry['cases']:
        got = promonet_amd.edit.from_features(
            loud, pit, per, pg, case['pitch_shift_cents'],
            case['time_stretch_ratio'], None, case['stretch_unvoiced'],
            case['stretch_silence'], return_grid=True)
        assert got[4].shape == case['grid'].shape
        error = max_abs(got[4], case['grid'])
        print(f"selective stretch {case['time_stretch_ratio']}: grid {error:.3e}")
        assert error < 1e-3          # frames; 60 dependent fp32 steps
        for mine, want, tolerance in zip(
                got[:4], case['outputs'], (2e-2, 2e-1, 1e-3, 1e-3)):
            assert mine.shape == want.shape
            assert max_abs(mine, want) < tolerance, case
    # the combination the reference cannot run (it mixes phoneme strings into
    # its index list): every phoneme but silence is stretched
    got = promonet_amd.edit.from_features(
        loud, pit, per, pg, time_stretch_ratio=1.2, stretch_unvoiced=True,
        stretch_silence=False, return_grid=True)
    want = oracle.grid_selective(
        inputs[3][0], 1.2, oracle.stretched_phonemes(True, False))
    assert max_abs(got[4], want) < 1e-3


def test_edit_then_synthesize_stays_on_device(device, golden_default):
    """README usage: edit -> synthesize without leaving the GPU."""
    import promonet_amd
    state = oracle.random_state(seed=0)
    state['pitch_distribution'] = golden_default['pitch_distribution'].clone()
    promonet_amd.configure(COMPUTE_DTYPE='fp32')
    try:
        model = promonet_amd.model.Generator()
        model.load_state_dict(state)
    finally:
        promonet_amd.configure(
            COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
    promonet_amd.synthesize.set_model(model, device)
    inputs = oracle.synthetic_inputs(1, 30, seed=2)
    args = [inputs[0][0], inputs[1], inputs[2], inputs[3][0]]
    edited = promonet_amd.edit.from_features(
        *[t.to(device) for t in args], pitch_shift_cents=300.,
        time_stretch_ratio=1.5)
    audio = promonet_amd.synthesize.from_features(
        edited[0], edited[1], edited[2], edited[3][None], gpu=0)
    want_features = oracle.edit_from_features(*args, 300., 1.5)
    want = oracle.from_features(
        want_features[0], want_features[1], want_features[2],
        want_features[3][None], state)
    assert audio.shape == want.shape == (1, 20 * 256)
    assert max_abs(audio, want) < 1e-5


def test_selective_grid_rejects_bad_selections(device):
    """Rows outside the PPG and selections without probability mass fail
    loudly (the reference's arithmetic yields NaN / inf / a grid that runs
    backwards, silently)."""
    import promonet_amd
    gen = torch.Generator().manual_seed(2)
    ppg = torch.softmax(torch.randn(40, 50, generator=gen), dim=0).to(device)
    grid = promonet_amd.edit.grid.selective(ppg, 1.3, [1, 5, 9])
    assert grid.shape == (round(50 / 1.3),) and bool((grid[1:] > grid[:-1]).all())
    with pytest.raises(ValueError, match='outside'):
        promonet_amd.edit.grid.selective(ppg, 1.3, [1, 40])
    with pytest.raises(ValueError, match='outside'):
        promonet_amd.edit.grid.selective(ppg[:8], 1.3, [1, 9])
    empty = ppg.clone()
    empty[3] = 0.
    with pytest.raises(ValueError, match='probability mass'):
        promonet_amd.edit.grid.selective(empty, 1.3, [3])


def test_stretch_grid_out_of_range_row_is_nan_bits(device):
    """C ABI below the Python validation: a phoneme row outside the PPG reads
    nothing and poisons grid and selection with the quiet-NaN BIT PATTERN
    (written through integer stores - the library is built -fno-honor-nans,
    so nothing may rest on NaN arithmetic); valid rows are unaffected."""
    from promonet_amd import _lib
    gen = torch.Generator().manual_seed(5)
    ppg = torch.softmax(torch.randn(40, 50, generator=gen), dim=0).to(device)
    for rows, poisoned in (([1, 40], True), ([-1], True), ([1, 5, 9], False)):
        index = torch.tensor(rows, dtype=torch.int32, device=device)
        selected = torch.zeros(50, device=device)
        grid = torch.zeros(38, device=device)
        _lib.check(_lib.lib().pm_stretch_grid(
            _lib.ptr(ppg), 40, _lib.ptr(index, torch.int32), len(rows),
            _lib.ptr(selected), _lib.ptr(grid), 50, 38, _lib.stream()))
        torch.cuda.synchronize()
        bits = grid.view(torch.int32).cpu()
        if poisoned:
            assert bool((bits == 0x7fc00000).all())
            assert bool((selected.view(torch.int32).cpu() == 0x7fc00000).all())
        else:
            assert bool(torch.isfinite(grid).all())
            assert max_abs(selected, ppg[rows].sum(0)) < 1e-6
