"""CPU: the oracle (oracle/restatement.py) against the golden vectors the
REAL reference produced (oracle/make_golden.py, build container only)."""
import pytest
import torch

import restatement as oracle
from util import max_abs


def test_small_config_reference_state(golden_small):
    """Reference-constructed weights (weight_g / weight_v pairs), every stage."""
    g = golden_small
    with torch.inference_mode():
        features = oracle.prepare_features(
            *g['inputs'][:4], g['state']['pitch_distribution'],
            g['state']['pitch_embedding.weight'], g['state']['ppg_threshold'])
        glob = oracle.prepare_global_features(
            *g['inputs'][4:7], g['state']['speaker_embedding.weight'])
        audio = oracle.generator_forward(*g['inputs'], g['state'])
    assert max_abs(features, g['features']) == 0.
    assert max_abs(glob, g['global_features']) == 0.
    assert audio.shape == g['audio'].shape == (2, 1, 24 * 256)
    assert max_abs(audio, g['audio']) < 1e-6


def test_default_config_seeded_weights(golden_default):
    g = golden_default
    state = oracle.random_state(seed=g['seed'])
    assert torch.equal(state['pitch_distribution'][:0], g['pitch_distribution'][:0])
    state['pitch_distribution'] = g['pitch_distribution'].clone()
    for name in ('b2_t40', 'b1_t7'):
        entry = g[name]
        inputs = oracle.synthetic_inputs(
            entry['batch'], entry['frames'], seed=g['input_seed'])
        with torch.inference_mode():
            audio = oracle.generator_forward(*inputs, state)
        assert max_abs(audio, entry['audio']) < 1e-6, name


def test_from_features_contract(golden_default):
    """(1, 256 T) float32, utterance 0 only (synthesize/core.py:59, :281)."""
    g = golden_default
    entry = g['from_features']
    state = oracle.random_state(seed=g['seed'])
    state['pitch_distribution'] = g['pitch_distribution'].clone()
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    audio = oracle.from_features(
        inputs[0][0], inputs[1], inputs[2], inputs[3], state,
        entry['speaker'], entry['spectral_balance_ratio'],
        entry['loudness_ratio'])
    assert audio.shape == (1, entry['frames'] * 256)
    assert audio.dtype == torch.float32
    assert max_abs(audio, entry['audio']) < 1e-6


def test_prepare_features_golden(golden_default):
    g = golden_default
    entry = g['features']
    state = oracle.random_state(seed=g['seed'])
    inputs = oracle.synthetic_inputs(2, entry['frames'], seed=entry['input_seed'])
    wide = oracle.synthetic_inputs(
        2, entry['frames'], seed=entry['input_seed'], loudness_rows=513)[0]
    args = (g['pitch_distribution'], state['pitch_embedding.weight'],
            state['ppg_threshold'])
    assert max_abs(
        oracle.prepare_features(*inputs[:4], *args), entry['rows8']) < 1e-6
    assert max_abs(
        oracle.prepare_features(wide, *inputs[1:4], *args),
        entry['rows513']) < 1e-6
    assert entry['rows8'].shape == (2, 113, entry['frames'])
    # channel layout [ppg 40 | pitch 64 | loudness 8 | periodicity 1]
    assert torch.equal(entry['rows8'][:, 112], inputs[2])
    assert max_abs(
        entry['rows8'][:, 104:112], (inputs[0] + 100.) / 120.) < 1e-6
    # sparsify keeps the top 6 of 40 and renormalises
    ppg = entry['rows8'][:, :40]
    assert ((ppg > 1e-6).sum(1) == 6).all()
    assert max_abs(ppg.sum(1), torch.ones(2, entry['frames'])) < 1e-5


def test_pitch_bin_edges(golden_default):
    """searchsorted(right=False) semantics quoted in SURVEY.md 3.2."""
    edges = golden_default['pitch_distribution']
    hz = torch.tensor(
        [[1., 50., 56., edges[-1].item(), 600., edges[0].item()]])
    bins = oracle.pitch_bins(hz, edges)
    assert bins.tolist() == [[0, 0, 1, 255, 255, 0]]
    assert (edges[1:] >= edges[:-1]).all() and edges.shape == (256,)


def test_spectrogram_golden(golden_default):
    entry = golden_default['spectrogram']
    for key in ('one', 'many'):
        got = oracle.spectrogram(entry[key + '_input'])
        assert got.shape == entry[key].shape
        assert max_abs(got, entry[key]) < 1e-6
        dft = oracle.spectrogram_dft(entry[key + '_input'])
        assert max_abs(dft.reshape(entry[key].shape), entry[key]) < 2e-5


def test_weight_norm_fold():
    gen = torch.Generator().manual_seed(0)
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 4, 3))
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2, 1))
    for module in (conv, convt):
        module.weight_g.data = torch.rand(
            module.weight_g.shape, generator=gen) + .5
        x = torch.randn(1, 6, 9, generator=gen)
        want = module(x)
        folded = oracle.fold_weight_norm(
            module.weight_g.data, module.weight_v.data)
        if module is conv:
            got = torch.nn.functional.conv1d(x, folded, module.bias, padding=0)
        else:
            got = torch.nn.functional.conv_transpose1d(
                x, folded, module.bias, stride=2, padding=1)
        assert max_abs(got, want) < 1e-6


def test_unpinned_third_party_sanity():
    """librosa / ppgs restatements are parity-unpinned (packages absent);
    these are the self-consistency checks of SURVEY.md appendix A."""
    import numpy as np
    weights = oracle.a_weighting(np.array([0., 21.5, 100., 1000., 10000.]))
    assert weights[0] == -80.
    assert abs(weights[3]) < 1e-2 and abs(weights[2] + 19.14) < 0.05
    assert abs(weights[4] + 2.49) < 0.05
    basis = oracle.mel_basis()
    assert basis.shape == (80, 513) and (basis.sum(1) > 0).all()
    assert abs(basis[0].max().item() - 0.02317) < 1e-4
    gen = torch.Generator().manual_seed(1)
    audio = torch.randn(1, 256 * 12, generator=gen) * .1
    loud = oracle.loudness(audio, 8)
    assert loud.shape == (8, 12) and (loud >= -100.).all()
    full = oracle.loudness(audio, None)
    assert full.shape == (513, 12)
    assert max_abs(oracle.band_average(full, 8), loud) < 1e-5


def test_fargan_golden(golden_fargan, golden_default):
    """FARGAN restatement vs audio computed by the real reference
    (config/fargan.py), incl. non-zero previous samples."""
    state = oracle.random_state_fargan(seed=golden_fargan['seed'])
    state['pitch_distribution'] = golden_default['pitch_distribution'].clone()
    for name in ('b2_t8', 'b1_t60', 'b3_t25'):
        entry = golden_fargan[name]
        inputs = oracle.synthetic_inputs(
            entry['batch'], entry['frames'], seed=entry['input_seed'])
        with torch.inference_mode():
            audio = oracle.fargan_generator_forward(
                *inputs, state, entry['previous'])
        assert audio.shape == (entry['batch'], 1, entry['frames'] * 256)
        assert max_abs(audio, entry['audio']) < 1e-6, name


def test_edit_golden(golden_default):
    """promonet.edit restatement vs the real reference (grid from the golden:
    its constructor is third-party ppgs)."""
    entry = golden_default['edit']
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    args = (inputs[0][0], inputs[1], inputs[2], inputs[3][0])
    for case in entry['cases']:
        got = oracle.edit_from_features(
            *[t.clone() for t in args], case['pitch_shift_cents'],
            case['time_stretch_ratio'], case['loudness_scale_db'],
            grid=case['grid'])
        for mine, want in zip(got, case['outputs']):
            assert mine.shape == want.shape and max_abs(mine, want) < 1e-5
        if case['grid'] is not None:
            assert max_abs(oracle.grid_constant(
                args[3], case['time_stretch_ratio']), case['grid']) == 0.
    sample = entry['sample']
    assert max_abs(oracle.grid_sample(
        sample['sequence'], sample['grid']), sample['linear']) < 1e-6
    assert max_abs(oracle.grid_sample(
        sample['sequence'], sample['grid'], 'nearest'), sample['nearest']) == 0.


@pytest.mark.parametrize('which', ['zero_shot', 'sparse_none'])
def test_conditioning_variants(which):
    """ZERO_SHOT / SPARSE_PPG_METHOD = None goldens (real reference,
    oracle/make_golden.py) against the restatement."""
    from conftest import GOLDEN
    g = torch.load(GOLDEN / f'variant_{which}.pt', weights_only=False)
    state = g['state']
    with torch.inference_mode():
        features = oracle.prepare_features(
            *g['inputs'][:4], state['pitch_distribution'],
            state['pitch_embedding.weight'], state.get('ppg_threshold', 0.),
            g['config']['SPARSE_PPG_METHOD'])
        glob = oracle.prepare_global_features(
            *g['inputs'][4:7], state['speaker_embedding.weight'],
            state.get('speaker_embedding.bias'))
        audio = oracle.hifigan_forward(features, glob, state)
    assert max_abs(features, g['features']) < 1e-6
    assert max_abs(glob, g['global_features']) < 1e-6
    assert max_abs(audio, g['audio']) < 1e-6


def test_sparsify_second_source():
    """ppgs.sparsify is unpinned (package absent): second-source the quantile
    against numpy's default 'linear' method, ties included, and check the
    topk / constant variants against direct definitions."""
    import numpy as np
    gen = torch.Generator().manual_seed(4)
    ppg = torch.softmax(3. * torch.randn(3, 40, 50, generator=gen), dim=1)
    ppg[0, :, 7] = 1. / 40
    ppg[1, 10:20, 8] = ppg[1, 20:30, 8]
    for q in (.85, .5, 0., 1., .33):
        mine = torch.quantile(ppg, q, dim=-2, keepdim=True)
        theirs = np.quantile(ppg.double().numpy(), q, axis=-2, keepdims=True)
        assert np.abs(mine.numpy() - theirs).max() < 1e-7
        out = oracle.sparsify(ppg, 'percentile', q)
        assert torch.allclose(out.sum(-2), torch.ones(3, 50), atol=1e-5)
        # softmax(log(p + 1e-8)) == (p + 1e-8) / sum(p + 1e-8)
        masked = torch.where(ppg > mine, ppg, torch.zeros_like(ppg)) + 1e-8
        direct = masked / masked.sum(-2, keepdim=True)
        assert torch.allclose(out, direct, rtol=1e-4, atol=1e-12)
    top = oracle.sparsify(ppg, 'topk', 5)
    fifth = torch.sort(ppg, dim=-2, descending=True).values[:, 4:5]
    distinct = ((ppg == fifth).sum(-2) == 1)               # no tie at the cut
    assert ((top > 1e-7).sum(-2)[distinct] == 5).all()
    constant = oracle.sparsify(ppg, 'constant', torch.tensor(.05))
    masked = torch.where(ppg > .05, ppg, torch.zeros_like(ppg)) + 1e-8
    assert torch.allclose(
        constant, masked / masked.sum(-2, keepdim=True), rtol=1e-4, atol=1e-12)


def test_selective_time_stretch():
    """edit/core.py:57-110 restated vs the golden of the real reference."""
    from conftest import GOLDEN
    entry = torch.load(GOLDEN / 'edit_voiced.pt', weights_only=False)
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    for case in entry['cases']:
        out = oracle.edit_from_features(
            inputs[0][0], inputs[1], inputs[2], inputs[3][0],
            case['pitch_shift_cents'], case['time_stretch_ratio'], None, None,
            case['stretch_unvoiced'], case['stretch_silence'])
        grid = oracle.grid_selective(
            inputs[3][0], case['time_stretch_ratio'],
            oracle.stretched_phonemes(
                case['stretch_unvoiced'], case['stretch_silence']))
        assert max_abs(grid, case['grid']) < 1e-5
        for mine, want in zip(out, case['outputs']):
            assert mine.shape == want.shape and max_abs(mine, want) < 1e-3
    # a stretched-everything selection degenerates to a constant step
    grid = oracle.grid_selective(inputs[3][0], 2., list(range(40)))
    assert grid.shape == (30,)
    assert torch.allclose(
        grid[1:] - grid[:-1], torch.full((29,), 61. / 30.), atol=1e-3)


def test_a_weighting_second_source():
    """librosa.A_weighting is unpinned (package absent): second-source the
    restated closed form against the ANALOG A-weighting filter of IEC 61672
    evaluated with scipy.signal.freqs (poles at 20.6 Hz x2, 107.7 Hz, 737.9 Hz,
    12194 Hz x2, four zeros at DC, 0 dB at 1 kHz)."""
    import numpy as np
    import scipy.signal
    f1, f2, f3, f4 = 20.598997, 107.65265, 737.86223, 12194.217
    poles = -2 * np.pi * np.array([f1, f1, f2, f3, f4, f4])
    b, a = scipy.signal.zpk2tf([0., 0., 0., 0.], poles, 1.)
    freqs = np.linspace(0, 11025, 513)[1:]                 # skip DC (-inf)
    _, h = scipy.signal.freqs(b, a, 2 * np.pi * freqs)
    _, h1k = scipy.signal.freqs(b, a, [2 * np.pi * 1000.])
    analog = 20 * np.log10(np.abs(h) / np.abs(h1k))
    restated = oracle.a_weighting(freqs, min_db=-1e9)
    # the closed form's "+ 2.0" is the 1 kHz normalisation rounded to 0.01 dB
    assert np.abs(restated - analog).max() < 5e-3
    full = oracle.a_weighting(np.linspace(0, 11025, 513))
    assert full[0] == -80. and abs(full[46] - 0.) < .05     # DC floor; ~991 Hz
    # the IEC 61672 table
    for hz, db in ((100., -19.1), (1000., 0.), (10000., -2.5), (31.5, -39.4)):
        assert abs(oracle.a_weighting(np.array([hz]))[0] - db) < .2   # nominal band centres
    weights = oracle.perceptual_weights()
    assert weights.shape == (513, 1) and weights[0, 0] == -100.


def test_mel_basis_second_source():
    """librosa.filters.mel is unpinned: check the restated basis against the
    Slaney Auditory-Toolbox definition it implements - linear below 1 kHz at
    200/3 Hz per mel, log-spaced above with 27 steps per factor 6.4, unit-area
    triangles (slaney norm) - and against the product's copy."""
    import numpy as np
    assert abs(float(oracle.hz_to_mel_slaney(torch.tensor(1000.))) - 15.) < 1e-6
    assert abs(float(oracle.hz_to_mel_slaney(torch.tensor(6400.))) - 42.) < 1e-5
    assert abs(float(oracle.mel_to_hz_slaney(torch.tensor(3.))) - 200.) < 1e-4
    round_trip = oracle.mel_to_hz_slaney(
        oracle.hz_to_mel_slaney(torch.tensor([50., 999., 1001., 8000.])))
    assert torch.allclose(
        torch.as_tensor(round_trip).float(),
        torch.tensor([50., 999., 1001., 8000.]), rtol=1e-6)
    basis = oracle.mel_basis().double().numpy()
    assert basis.shape == (80, 513) and (basis >= 0).all()
    assert (basis.sum(1) > 0).all()                      # no empty filter
    # slaney norm: every triangle has unit area in Hz (sampled every 21.5 Hz;
    # the narrow low filters are sampled coarsely)
    area = basis.sum(1) * (11025. / 512)
    assert np.abs(area[30:] - 1.).max() < .05
    assert np.abs(area - 1.).max() < .35
    # peaks ascend, one filter's peak is the next one's lower edge
    peaks = basis.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] >= 1 and peaks[-1] < 512
    import promonet_amd
    ours = promonet_amd.preprocess.spectrogram.mel_basis()
    assert max_abs(ours, oracle.mel_basis()) < 1e-7



def test_librosa_arithmetic_against_transformers():
    """Rows a13 / a14 second-sourced against transformers.audio_utils - an
    independent third-party implementation of librosa's mel filter bank / STFT
    / amplitude_to_db (its documentation states the equivalence; the image
    pins transformers 5.15; librosa itself is absent, so this is a SECOND
    SOURCE, not a pin against the reference's own dependency). Runs in a
    fresh interpreter: oracle/reference_import.py replaces `transformers`
    (and what it imports) by mocks once a test has imported the reference."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / 'second_source_hf.py'
    out = subprocess.run(
        [sys.executable, str(script)], capture_output=True, text=True,
        timeout=300)
    print(out.stdout)
    if out.returncode == 77:
        pytest.skip('transformers is not installed')
    assert out.returncode == 0, out.stdout + out.stderr
