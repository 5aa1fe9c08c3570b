"""GPU parity of the whole hot path (through the Python drop-in API, which
calls the C ABI) against the committed reference goldens and the CPU oracle.

Gate (BASELINE.json north_star): max-abs <= 1e-4 vs the fp32 reference
generator; the fp32-operand mode is held to 2e-6."""
import pytest
import torch

import restatement as oracle
from util import check, max_abs, rel_err

pytestmark = pytest.mark.gpu

GATE = {'fp32': 2e-6, 'f16': 1e-4, 'bf16': 1e-4}
# What the default configuration (random init, output abs-max 0.017) actually
# measures on MI355X is 5e-8 / 3.1e-6 / 3.0e-5 (profiles/r04/measured_errors.json):
# besides BASELINE.json's gate every default-config check is held to <= 3x that,
# so a regression that triples the error fails here and not only at full size.
TIGHT = {'fp32': 2e-7, 'f16': 9e-6, 'bf16': 8e-5, 'f16+f16+f16+f16x3': 9e-7,
         'f16+f16+f16ux+f16a2': 1.5e-6}
GATE['f16+f16+f16+f16x3'] = 1e-4
GATE['f16+f16+f16ux+f16a2'] = 1e-4     # (what 'checkpoint' stands for)
# relative to the output's abs-max, for the tiny test vocoders (below); the
# 64-initial-channel fixture (4 channels in its last stage: few products per
# output) measures 1.4e-6 / 7.6e-4 / 6.8e-3, the 32-channel conditioning
# variants 3.3e-7 / 2.5e-4 / 1.4e-3
RELATIVE_SMALL = {'fp32': 4e-6, 'f16': 2e-3, 'bf16': 1.5e-2}
RELATIVE_VARIANT = {'fp32': 1e-6, 'f16': 7.5e-4, 'bf16': 4e-3}


def gate(dtype, scale, relative=RELATIVE_SMALL):
    """The 1e-4 max-abs gate is BASELINE.json's, stated for the default
    configuration (512 initial channels, random-init output abs-max 0.017) and
    applied to it unchanged (plus TIGHT). The tiny vocoders of the
    reference-weight fixtures (64 / 32 initial channels, outputs up to 0.45,
    far fewer products averaged per output) are held to a bound relative to
    their output scale instead, <= 3x what they measure."""
    return relative[dtype] * scale


def make_model(state, dtype, device):
    import promonet_amd
    promonet_amd.configure(COMPUTE_DTYPE=dtype)
    model = promonet_amd.model.Generator()
    model.load_state_dict(state)
    promonet_amd.configure(
        COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
    return model.to(device).eval()


@pytest.fixture(scope='module')
def default_state(golden_default):
    state = oracle.random_state(seed=golden_default['seed'])
    state['pitch_distribution'] = golden_default['pitch_distribution'].clone()
    return state


def on(device, inputs):
    return [t.to(device) for t in inputs]


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16', 'f16+f16+f16+f16x3',
                                   'f16+f16+f16ux+f16a2'])
def test_generator_matches_reference_golden(
    device, golden_default, default_state, dtype
):
    """Default config: weights regenerated from the seed, audio computed by
    the REAL reference at fixture time (oracle/make_golden.py)."""
    model = make_model(default_state, dtype, device)
    for name in ('b2_t40', 'b1_t7'):
        entry = golden_default[name]
        inputs = oracle.synthetic_inputs(
            entry['batch'], entry['frames'], seed=golden_default['input_seed'])
        with torch.inference_mode():
            got = model(*on(device, inputs), None)
        assert got.shape == entry['audio'].shape
        error = max_abs(got, entry['audio'])
        print(f'{dtype} {name}: max-abs {error:.3e}')
        check(error, min(GATE[dtype], TIGHT[dtype]), f'golden_default:{dtype}',
              name)


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16'])
def test_small_config_reference_weights(device, golden_small, dtype):
    """Tiny config (64 initial channels -> 32/16/8/4, exercises channel
    padding) with the reference-constructed state dict."""
    import promonet_amd
    promonet_amd.configure(
        HIFIGAN_UPSAMPLE_INITIAL_SIZE=64, COMPUTE_DTYPE=dtype)
    try:
        model = promonet_amd.model.Generator()
        model.load_state_dict(golden_small['state'])
        model = model.to(device).eval()
        with torch.inference_mode():
            got = model(*on(device, golden_small['inputs']), None)
            features = model.prepare_features(
                *on(device, golden_small['inputs'][:4]))
            global_features = model.prepare_global_features(
                *on(device, golden_small['inputs'][4:7]))
    finally:
        promonet_amd.configure(
            HIFIGAN_UPSAMPLE_INITIAL_SIZE=512,
            COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
    assert max_abs(features, golden_small['features']) < 1e-6
    assert max_abs(global_features, golden_small['global_features']) == 0.
    error = max_abs(got, golden_small['audio'])
    scale = golden_small['audio'].abs().max().item()
    print(f'small {dtype}: max-abs {error:.3e} (abs-max {scale:.3e})')
    check(error / scale, gate(dtype, 1.), f'small_config_relative:{dtype}')


def test_prepare_features_golden(device, golden_default, default_state):
    model = make_model(default_state, 'f16', device)
    entry = golden_default['features']
    inputs = oracle.synthetic_inputs(2, entry['frames'], seed=entry['input_seed'])
    wide = oracle.synthetic_inputs(
        2, entry['frames'], seed=entry['input_seed'], loudness_rows=513)[0]
    with torch.inference_mode():
        rows8 = model.prepare_features(*on(device, inputs[:4]))
        rows513 = model.prepare_features(
            wide.to(device), *on(device, inputs[1:4]))
        glob = model.prepare_global_features(*on(device, inputs[4:7]))
    # pitch-embedding channels are gathers: exact
    assert max_abs(rows8[:, 40:104], entry['rows8'][:, 40:104]) == 0.
    assert max_abs(rows8, entry['rows8']) < 1e-6
    assert max_abs(rows513, entry['rows513']) < 1e-6
    assert max_abs(glob, entry['global']) == 0.


def test_prepare_features_edge_cases(device, default_state):
    """Pitch outside [FMIN, FMAX], exactly on bin edges, ties in the PPG."""
    model = make_model(default_state, 'f16', device)
    frames = 300
    inputs = list(oracle.synthetic_inputs(1, frames, seed=8))
    edges = default_state['pitch_distribution']
    pitch = torch.cat([
        torch.tensor([0., 10., 49.9, 50., 550., 551., 1e4]),
        edges[:100], edges[-5:], edges[100:105] + 1e-4])
    inputs[1] = torch.cat([pitch, inputs[1][0, pitch.numel():]])[None]
    inputs[3][0, :, 5] = 1. / 40           # all-equal PPG frame
    inputs[3][0, :, 6] = 0.
    inputs[3][0, 3, 6] = 1.                # one-hot PPG frame
    want = oracle.prepare_features(
        *inputs[:4], edges, default_state['pitch_embedding.weight'],
        default_state['ppg_threshold'])
    with torch.inference_mode():
        got = model.prepare_features(*on(device, inputs[:4]))
    assert max_abs(got[:, 40:104], want[:, 40:104]) == 0.
    assert max_abs(got, want) < 1e-6


def test_from_features_api(device, golden_default, default_state):
    """promonet.synthesize.from_features signature and return convention:
    (1, 256 T) float32, item 0 only (synthesize/core.py:18-59, :281)."""
    import promonet_amd
    entry = golden_default['from_features']
    model = make_model(default_state, 'fp32', device)
    promonet_amd.synthesize.set_model(model, device)
    inputs = oracle.synthetic_inputs(1, entry['frames'], seed=entry['input_seed'])
    got = promonet_amd.synthesize.from_features(
        inputs[0][0], inputs[1], inputs[2], inputs[3],
        speaker=entry['speaker'],
        spectral_balance_ratio=entry['spectral_balance_ratio'],
        loudness_ratio=entry['loudness_ratio'], gpu=0)
    assert got.dtype == torch.float32
    assert got.shape == entry['audio'].shape == (1, entry['frames'] * 256)
    assert max_abs(got, entry['audio']) < GATE['fp32']
    with pytest.raises(RuntimeError):
        promonet_amd.synthesize.from_features(
            inputs[0][0], inputs[1], inputs[2], inputs[3])     # no gpu


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16'])
def test_config1_two_second_from_features(device, default_state, dtype):
    """BASELINE.json configs[0] on the HIP path: ONE synthetic 2 s utterance
    (172 frames -> 44 032 samples) through promonet.synthesize.from_features
    (synthesize/core.py:18-59), every sample against the CPU oracle. The
    single-utterance call takes the latency (narrow-tile) variants of the
    kernels, which the batch-32 tests never reach."""
    import promonet_amd
    frames = 172
    model = make_model(default_state, dtype, device)
    promonet_amd.synthesize.set_model(model, device)
    inputs = oracle.synthetic_inputs(1, frames, seed=2024)
    got = promonet_amd.synthesize.from_features(
        inputs[0][0], inputs[1], inputs[2], inputs[3], speaker=3, gpu=0)
    with torch.inference_mode():
        want = oracle.generator_forward(
            inputs[0], inputs[1], inputs[2], inputs[3],
            torch.tensor([3]), torch.ones(1), torch.ones(1), default_state)[0]
    assert got.dtype == torch.float32     # (returned on `gpu`, as the reference does)
    assert got.shape == want.shape == (1, frames * 256)
    error = max_abs(got, want)
    print(f'config 1 (1 x {frames} frames, from_features) {dtype}: max-abs '
          f'{error:.3e} (abs-max {want.abs().max().item():.3e})')
    check(error, min(GATE[dtype], TIGHT[dtype]), f'config1_from_features:{dtype}')


def test_batched_matches_single_and_is_deterministic(device, default_state):
    """Size-independent properties at a larger size than the oracle is
    comfortable with: every utterance of a batch equals the same utterance
    synthesised alone (no cross-utterance coupling, tiles never straddle
    utterances), repeated runs are bit-identical."""
    model = make_model(default_state, 'f16', device)
    batch, frames = 6, 333
    inputs = on(device, oracle.synthetic_inputs(batch, frames, seed=21))
    with torch.inference_mode():
        full = model(*inputs, None)
        again = model(*inputs, None)
        assert torch.equal(full, again)
        for item in (0, 3, 5):
            single = model(*[t[item:item + 1] for t in inputs], None)
            assert torch.equal(single[0], full[item])
    assert full.shape == (batch, 1, frames * 256)
    assert torch.isfinite(full).all()


def test_time_tiling_with_halo(device, default_state):
    """Fully-convolutional property: synthesising frames
    [a - 14, b + 14) and cropping reproduces the whole-utterance output."""
    model = make_model(default_state, 'fp32', device)
    frames, a, b, halo = 120, 40, 80, 14
    inputs = on(device, oracle.synthetic_inputs(1, frames, seed=4))
    with torch.inference_mode():
        full = model(*inputs, None)
        sliced = [
            t[..., a - halo:b + halo] if t.ndim >= 2 and t.shape[-1] == frames
            else t for t in inputs]
        part = model(*sliced, None)
    want = full[..., a * 256:b * 256]
    got = part[..., halo * 256:(halo + b - a) * 256]
    assert max_abs(got, want) < 2e-6


def oracle_window(inputs, item, first, last, frames, state, halo=14):
    """Reference audio of frames [first, last) of utterance `item`, computed
    by the CPU oracle on frames [first - halo, last + halo) (the generator's
    receptive field is 14 frames: test_time_tiling_with_halo) clipped to the
    utterance, where the true edges (zero padding) apply."""
    lo, hi = max(0, first - halo), min(frames, last + halo)
    piece = [t[item:item + 1, ..., lo:hi] if t.ndim >= 2 else t[item:item + 1]
             for t in inputs]
    audio = oracle.generator_forward(*piece, state)
    return audio[..., (first - lo) * 256:(last - lo) * 256]


def check_windows(
    model, device, state, batch, frames, seed, gate, windows, kind
):
    inputs = oracle.synthetic_inputs(batch, frames, seed=seed)
    with torch.inference_mode():
        full = model(*on(device, inputs), None)
    assert full.shape == (batch, 1, frames * 256)
    assert torch.isfinite(full).all() and full.abs().max() <= 1.
    gen = torch.Generator().manual_seed(seed)
    span = 172                                            # 2 s of audio
    picks = [(0, 0), (batch - 1, frames - span)]          # both true edges
    while len(picks) < windows:
        picks.append((
            int(torch.randint(0, batch, (1,), generator=gen)),
            int(torch.randint(0, frames - span, (1,), generator=gen))))
    worst = 0.
    for item, first in picks:
        want = oracle_window(inputs, item, first, first + span, frames, state)
        got = full[item:item + 1, :, first * 256:(first + span) * 256]
        error = max_abs(got, want)
        print(f'utterance {item} frames [{first}, {first + span}): '
              f'max-abs {error:.3e} (abs-max {want.abs().max().item():.3e})')
        worst = max(worst, error)
    check(worst, gate, kind)
    return full, inputs


def test_full_size_config3_windows(device, default_state):
    """BASELINE.json configs[2] at full size (batch 32 x 861 frames, bf16 MFMA
    operands): 6 two-second windows - the first and last of the batch, which
    contain the true utterance edges, and 4 random (utterance, offset) pairs
    - against the fp32 CPU oracle at the 1e-4 gate (every sample is checked
    by test_full_size_every_sample); plus the size-independent property: an
    utterance equals its stand-alone synthesis bit for bit."""
    model = make_model(default_state, 'bf16', device)
    full, inputs = check_windows(
        model, device, default_state, 32, 861, 1234,
        min(GATE['bf16'], TIGHT['bf16']), windows=6, kind='config3_windows:bf16')
    with torch.inference_mode():
        single = model(*[t[7:8] for t in on(device, inputs)], None)
    assert torch.equal(single[0], full[7])


def test_full_size_fp32_config2(device, default_state):
    """BASELINE.json configs[1] at full size (batch 8 x 430 frames, exact
    fp32 MFMA): 4 windows against the CPU oracle at the 2e-6 gate."""
    model = make_model(default_state, 'fp32', device)
    check_windows(
        model, device, default_state, 8, 430, 77,
        min(GATE['fp32'], TIGHT['fp32']), windows=4, kind='config2_windows:fp32')


def test_full_size_every_sample(device, default_state):
    """BASELINE.json configs[2] without sampling: the CPU oracle synthesises
    the WHOLE batch (32 x 861 frames = 7 053 312 samples, about a minute on 8
    host threads) and every sample of the bf16 run (the dtype the config
    names) and of the f16 run (the library default) is held to the 1e-4
    max-abs gate; the error relative to the output peak is printed beside it
    (test_precision_at_trained_scale pins what it means at full scale)."""
    inputs = oracle.synthetic_inputs(32, 861, seed=1234)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))     # all-core runs are slower
    try:
        with torch.inference_mode():
            want = torch.cat([
                oracle.generator_forward(
                    *[t[i:i + 4] for t in inputs], default_state)
                for i in range(0, 32, 4)])
    finally:
        torch.set_num_threads(threads)
    assert want.shape == (32, 1, 861 * 256)
    for dtype in ('bf16', 'f16'):
        model = make_model(default_state, dtype, device)
        with torch.inference_mode():
            got = model(*on(device, inputs), None).cpu()
        difference = (got - want).abs()
        error = difference.max().item()
        rms = difference.pow(2).mean().sqrt().item()
        print(f'full size, all {want.numel()} samples, {dtype}: max-abs '
              f'{error:.3e} rms {rms:.3e} (abs-max {want.abs().max():.3e}, '
              f'rel_to_absmax {error / want.abs().max().item():.3e})')
        check(error, min(1e-4, TIGHT[dtype]), f'full_size_max_abs:{dtype}')
        # (rms: 8.2e-6 bf16 / 6.0e-7 f16 measured)
        check(rms, {'bf16': 1.5e-5, 'f16': 1.8e-6}[dtype],
              f'full_size_rms:{dtype}')
        del model
    # The same 7 053 312 samples at a TRAINED checkpoint's output scale, in the
    # library's default operand mode: the output conv rescaled so that the
    # audio peaks at 0.99 (section 3 of DESIGN.md). Its reference needs no
    # second CPU pass: the conv is linear in its weights, so the scaled model's
    # audio is tanh(f atanh(audio)) of the oracle's (evaluated in float64).
    import math
    peak = want.abs().max().item()
    factor = math.atanh(.99) / math.atanh(peak)
    scaled = dict(default_state)
    scaled['model.model.5.weight'] = scaled['model.model.5.weight'] * factor
    want_scaled = torch.tanh(factor * torch.atanh(want.double()))
    model = make_model(scaled, 'checkpoint', device)
    with torch.inference_mode():
        got = model(*on(device, inputs), None).cpu()
    difference = (got.double() - want_scaled).abs()
    error = difference.max().item()
    print(f'full size at audio peak {want_scaled.abs().max().item():.3f}, '
          f"'checkpoint' operands: max-abs {error:.3e} rms "
          f'{difference.pow(2).mean().sqrt().item():.3e}')
    check(error, 1e-4, 'full_size_peak0.99:checkpoint')


def test_precision_at_trained_scale(device, default_state):
    """The 1e-4 gate is met on the random-init default model at an output peak
    of 0.017; a trained checkpoint's audio peaks near 1. Rescaling the output
    conv so that the audio peaks at 0.5 leaves every operand rounding inside
    the trunk as it is and shows the absolute error a real checkpoint would
    see: pinned here per operand type (scripts/precision_sweep.py measures
    the same on every sample of batch 8 x 10 s)."""
    import math
    inputs = oracle.synthetic_inputs(2, 120, seed=99)
    with torch.inference_mode():
        peak = oracle.generator_forward(*inputs, default_state).abs().max()
    state = dict(default_state)
    state['model.model.5.weight'] = state['model.model.5.weight'] * (
        math.atanh(.5) / math.atanh(float(peak)))
    with torch.inference_mode():
        want = oracle.generator_forward(*inputs, state)
    scale = want.abs().max().item()
    assert .4 < scale < .6
    # absolute bounds at this scale (measured: 2.0e-6 / 7.2e-5 / 1.2e-4 / 7.7e-4)
    bounds = {'fp32': 6e-6, 'f16+f16+f16+f16x3': 3.5e-5,
              'f16+f16+f16ux+f16a2': 4.5e-5, 'f16': 2e-4,
              'bf16+bf16+bf16+f16': 3e-4, 'bf16': 2e-3}
    errors = {}
    for dtype, bound in bounds.items():
        model = make_model(state, dtype, device)
        with torch.inference_mode():
            got = model(*on(device, inputs), None)
        errors[dtype] = max_abs(got, want)
        print(f'trained scale (peak {scale:.2f}) {dtype}: max-abs '
              f'{errors[dtype]:.3e} = {errors[dtype] / scale:.3e} of the peak')
        check(errors[dtype], bound, f'trained_scale_peak0.5:{dtype}')
        del model
    # the last stage's operand type decides: f16 there recovers f16 accuracy
    assert errors['bf16+bf16+bf16+f16'] < .4 * errors['bf16']
    assert errors['fp32'] < errors['f16+f16+f16+f16x3'] < errors['f16'] \
        < errors['bf16']
    assert errors['f16+f16+f16ux+f16a2'] < .5 * errors['f16']


def test_trained_checkpoint_mode_holds_the_gate_near_full_scale(
    device, default_state
):
    """The operand mode INTEGRATION.md names for real checkpoints -
    'checkpoint' = 'f16+f16+f16ux+f16a2': f16 MFMA operands; in the last
    upsampling stage the ACTIVATIONS split into hi + lo (two MFMAs per step in
    its Blocks, its upsampler fully split), the upsampler in front of the
    stage before it fully split too - holds BASELINE.json's 1e-4 max-abs gate
    with the output conv rescaled so that the audio peaks at 0.99 (a trained
    generator's scale; random init peaks at 0.017), on every sample of batch
    4 x 4 s. Plain f16 does not (3e-4: one rounding of the last stage's
    activations), which is what the split exists for; the fully split last
    stage ('f16+f16+f16+f16x3', three MFMAs per step: the default until round
    6) is 0.85x the error for 1.04x the step; exact fp32 operands at 8x."""
    import math
    inputs = oracle.synthetic_inputs(4, 344, seed=99)
    with torch.inference_mode():
        peak = oracle.generator_forward(*inputs, default_state).abs().max()
    state = dict(default_state)
    state['model.model.5.weight'] = state['model.model.5.weight'] * (
        math.atanh(.99) / math.atanh(float(peak)))
    with torch.inference_mode():
        want = oracle.generator_forward(*inputs, state)
    scale = want.abs().max().item()
    assert .98 < scale < 1.
    import promonet_amd
    assert promonet_amd.config.DEFAULT_COMPUTE_DTYPE == 'checkpoint'
    errors, outputs = {}, {}
    for dtype in ('f16+f16+f16ux+f16a2', 'checkpoint', 'f16+f16+f16+f16x3',
                  'f16', 'fp32'):
        model = make_model(state, dtype, device)
        with torch.inference_mode():
            got = model(*on(device, inputs), None)
        outputs[dtype] = got
        errors[dtype] = max_abs(got, want)
        print(f'trained scale (peak {scale:.3f}) {dtype}: max-abs '
              f'{errors[dtype]:.3e}')
        del model
    # ('checkpoint', the library default, is that mode for any stage count)
    assert torch.equal(outputs['checkpoint'], outputs['f16+f16+f16ux+f16a2'])
    check(errors['checkpoint'], 1e-4, 'trained_scale_peak0.99:checkpoint')
    check(errors['f16+f16+f16+f16x3'], 1e-4, 'trained_scale_peak0.99:f16x3')
    check(errors['fp32'], 3e-5, 'trained_scale_peak0.99:fp32')
    assert errors['f16'] > 1e-4       # (why the mode exists)


###############################################################################
# Conditioning variants reachable from the same forward
###############################################################################


@pytest.mark.parametrize('which', ['zero_shot', 'sparse_none'])
@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16'])
def test_conditioning_variants_golden(device, which, dtype):
    """ZERO_SHOT (Linear over x-vectors, generator.py:35-38) and
    SPARSE_PPG_METHOD = None (:140-147), goldens from the REAL reference."""
    import promonet_amd
    from conftest import GOLDEN
    golden = torch.load(GOLDEN / f'variant_{which}.pt', weights_only=False)
    config = dict(golden['config'], COMPUTE_DTYPE=dtype)
    promonet_amd.configure(**config)
    try:
        model = promonet_amd.model.Generator()
        assert set(model.state_dict()) == set(golden['state'])
        model.load_state_dict(golden['state'])
        model = model.to(device).eval()
        inputs = on(device, golden['inputs'])
        with torch.inference_mode():
            features = model.prepare_features(*inputs[:4])
            global_features = model.prepare_global_features(*inputs[4:7])
            audio = model(*inputs, None)
    finally:
        promonet_amd.configure(
            HIFIGAN_UPSAMPLE_INITIAL_SIZE=512, ZERO_SHOT=False,
            SPARSE_PPG_METHOD='percentile',
            COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
    assert max_abs(features, golden['features']) < 1e-6
    assert max_abs(global_features, golden['global_features']) < 2e-6
    error = max_abs(audio, golden['audio'])
    scale = golden['audio'].abs().max().item()
    print(f'{which} {dtype}: max-abs {error:.3e} (abs-max {scale:.3f})')
    check(error / scale, gate(dtype, 1., RELATIVE_VARIANT),
          f'variant_relative:{dtype}', which)
    # this 32-channel vocoder's output is 27x larger than the default
    # config's (0.45 vs 0.017 abs-max): same relative precision


@pytest.mark.parametrize('method,threshold', [
    ('constant', .05), ('topk', 5.), ('topk', 40.), ('percentile', .5),
    ('percentile', 1.), (None, 0.)])
def test_sparse_ppg_methods(device, default_state, method, threshold):
    """promonet.SPARSE_PPG_METHOD variants against the oracle's restatement
    of ppgs.sparsify (third party, unpinned), ties included."""
    import promonet_amd
    promonet_amd.configure(
        SPARSE_PPG_METHOD=method, SPARSE_PPG_THRESHOLD=threshold)
    try:
        model = promonet_amd.model.Generator()
        state = {k: v for k, v in default_state.items()
                 if k != 'ppg_threshold' or method is not None}
        if method is not None:
            state['ppg_threshold'] = torch.tensor(threshold)
        model.load_state_dict(state)
        model = model.to(device).eval()
    finally:
        promonet_amd.configure(
            SPARSE_PPG_METHOD='percentile', SPARSE_PPG_THRESHOLD=.85)
    inputs = list(oracle.synthetic_inputs(2, 70, seed=12))
    inputs[3][0, :, 3] = 1. / 40                   # all-equal frame (ties)
    inputs[3][1, :20, 9] = inputs[3][1, 20:, 9]    # pairwise ties
    inputs[3][1, :, 9] /= inputs[3][1, :, 9].sum()
    want = oracle.prepare_features(
        *inputs[:4], default_state['pitch_distribution'],
        default_state['pitch_embedding.weight'], torch.tensor(threshold),
        method)
    with torch.inference_mode():
        got = model.prepare_features(*on(device, inputs[:4]))
    assert max_abs(got, want) < 1e-6


def test_speaker_ids_are_bounds_checked(device, default_state):
    """The reference's Embedding raises on a bad id; host-side ids raise here
    too, device-side ids (no sync) never read out of bounds: NaN row."""
    import promonet_amd
    model = make_model(default_state, 'f16', device)
    promonet_amd.synthesize.set_model(model, device)
    inputs = oracle.synthetic_inputs(1, 8, seed=2)
    for bad in (-1, promonet_amd.NUM_SPEAKERS):
        with pytest.raises(IndexError):
            promonet_amd.synthesize.from_features(
                inputs[0][0], inputs[1], inputs[2], inputs[3], speaker=bad,
                gpu=0)
    ones = torch.ones(3, device=device)
    ids = torch.tensor([0, promonet_amd.NUM_SPEAKERS, -5], device=device)
    with torch.inference_mode():
        glob = model.prepare_global_features(ids, ones, ones)
    assert torch.isfinite(glob[0]).all()
    assert torch.isnan(glob[1, :256]).all() and torch.isnan(glob[2, :256]).all()
    # opt-in: device-side ids validated like the reference (one sync)
    batch = oracle.synthetic_inputs(2, 8, seed=2)
    bad_ids = torch.tensor([1, promonet_amd.NUM_SPEAKERS], device=device)
    promonet_amd.configure(CHECK_DEVICE_SPEAKERS=True)
    try:
        with pytest.raises(IndexError):
            promonet_amd.synthesize.from_features_batched(
                *batch[:4], speakers=bad_ids, gpu=0)
    finally:
        promonet_amd.configure(CHECK_DEVICE_SPEAKERS=False)


###############################################################################
# Preprocessing
###############################################################################


def test_spectrogram_golden(device, golden_default):
    import promonet_amd
    entry = golden_default['spectrogram']
    one = promonet_amd.preprocess.spectrogram.from_audio(
        entry['one_input'].to(device))
    many = promonet_amd.preprocess.spectrogram.from_audio(
        entry['many_input'].to(device))
    assert one.shape == (513, 20) and many.shape == (3, 513, 10)
    for got, want in ((one, entry['one']), (many, entry['many'])):
        diff = (got.cpu() - want).abs()
        assert (diff <= 2e-5 + 1e-5 * want.abs()).all(), diff.max()


def test_spectrogram_long_and_ragged(device):
    """Lengths that are not hop multiples, minimum length, many frames."""
    import promonet_amd
    gen = torch.Generator().manual_seed(3)
    for samples in (385, 1000, 256 * 300 + 17):
        audio = torch.randn(2, 1, samples, generator=gen) * .1
        want = oracle.spectrogram(audio)
        got = promonet_amd.preprocess.spectrogram.from_audio(audio.to(device))
        assert got.shape == want.shape
        diff = (got.cpu() - want).abs()
        assert (diff <= 2e-5 + 1e-5 * want.abs()).all(), (samples, diff.max())


@pytest.mark.parametrize('group', [16, 32])
def test_spectrogram_fft_against_brute_force_dft(device, group):
    """The forward STFT is a 1024-point real FFT in LDS (pm_fft.h); the framed
    DFT GEMM on the exact-fp32 MFMA kernel (what rounds 1-2 shipped, still the
    backward pass) is an independent evaluation of the same transform. Both
    workgroup shapes, ragged length, utterance edges inside a workgroup."""
    import promonet_amd
    from promonet_amd import _lib
    lib = _lib.lib()
    gen = torch.Generator().manual_seed(21)
    try:
        _lib.check(lib.pm_stft_set_frames_per_group(group))
        for batch, samples in ((3, 256 * 70 + 31), (1, 500), (2, 256 * 33)):
            audio = (torch.randn(batch, samples, generator=gen) * .1).to(device)
            frames = samples // 256
            fft = promonet_amd.preprocess.spectrogram.from_audio(audio[:, None])
            fft = fft.reshape(batch, 513, frames)
            dft = torch.empty(batch, 513, frames, device=device)
            size = lib.pm_stft_scratch_bytes(batch, samples)
            scratch = torch.empty(size, dtype=torch.uint8, device=device)
            _lib.check(lib.pm_stft_magnitude_dft(
                _lib.ptr(audio), _lib.ptr(dft), batch, samples,
                scratch.data_ptr(), scratch.numel(), _lib.stream()))
            want = oracle.spectrogram(audio.cpu()[:, None]).reshape(
                batch, 513, frames)
            for name, got in (('fft', fft), ('dft', dft)):
                diff = (got.cpu() - want).abs()
                print(f'stft {name} group {group} {batch}x{samples}: '
                      f'max-abs {diff.max():.3e}')
                assert (diff <= 2e-5 + 1e-5 * want.abs()).all(), (name, samples)
            assert ((fft - dft).abs() <= 2e-5 + 1e-5 * dft.abs()).all()
    finally:
        _lib.check(lib.pm_stft_set_frames_per_group(16))


def test_mel(device):
    import promonet_amd
    gen = torch.Generator().manual_seed(5)
    audio = torch.randn(2, 1, 256 * 40, generator=gen) * .1
    want = oracle.spectrogram(audio, mels=True)
    got = promonet_amd.preprocess.spectrogram.from_audio(
        audio.to(device), mels=True)
    assert got.shape == want.shape == (2, 80, 40)
    # STFT / mel <= 2e-5 abs + 1e-5 rel (measured: 2.1e-6)
    assert ((got.cpu() - want).abs() <= 2e-5 + 1e-5 * want.abs()).all()
    basis = promonet_amd.preprocess.spectrogram.mel_basis()
    assert max_abs(basis, oracle.mel_basis()) < 1e-7
    # the fused kernel (FFT tile -> log-mel) against the two-step path
    two_step = promonet_amd.preprocess.spectrogram.linear_to_mel(
        promonet_amd.preprocess.spectrogram.from_audio(audio.to(device)))
    assert max_abs(got, two_step) < 1e-6
    clamped = promonet_amd.preprocess.spectrogram.from_audio(
        audio.to(device), mels=True,
        log_dynamic_range_compression_threshold=-1.)
    floor = torch.clamp(want, min=-1.)
    assert ((clamped.cpu() - floor).abs() <= 2e-5 + 1e-5 * floor.abs()).all()


def test_spectrogram_backward(device):
    """The training mel loss differentiates through spectrogram.from_audio
    (promonet/train/core.py:277-305): gradients of the HIP path (framed-DFT
    adjoint on the exact-fp32 MFMA kernel) against torch autograd through the
    oracle's torch.stft restatement, linear and mel, ragged length."""
    import promonet_amd
    gen = torch.Generator().manual_seed(11)
    for batch, samples, mels in ((2, 256 * 24, False), (1, 256 * 9 + 100, False),
                                 (3, 256 * 31, True)):
        audio = torch.randn(batch, 1, samples, generator=gen) * .1
        frames = samples // 256
        weight = torch.randn(
            batch, 80 if mels else 513, frames, generator=gen).squeeze(0)
        reference = audio.clone().requires_grad_(True)
        (oracle.spectrogram(reference, mels=mels) * weight).sum().backward()
        ours = audio.clone().to(device).requires_grad_(True)
        spec = promonet_amd.preprocess.spectrogram.from_audio(ours, mels)
        assert spec.requires_grad
        (spec * weight.to(device)).sum().backward()
        assert ours.grad.shape == reference.grad.shape
        error = max_abs(ours.grad, reference.grad)
        scale = reference.grad.abs().max().item()
        print(f'spectrogram backward {batch}x{samples} mels={mels}: '
              f'max-abs {error:.3e} (grad abs-max {scale:.3e})')
        assert error < 2e-5 * max(1., scale)
    # mel alone, clamp active on part of the tensor
    spec = (torch.rand(2, 513, 12, generator=gen) + .01)
    weight = torch.randn(2, 80, 12, generator=gen)
    reference = spec.clone().requires_grad_(True)
    (oracle.linear_to_mel(reference, threshold=-2.) * weight).sum().backward()
    ours = spec.clone().to(device).requires_grad_(True)
    out = promonet_amd.preprocess.spectrogram.linear_to_mel(ours, -2.)
    (out * weight.to(device)).sum().backward()
    assert max_abs(ours.grad, reference.grad) < 1e-4 * max(
        1., reference.grad.abs().max().item())
    # no graph is kept (and nothing extra allocated) under inference mode
    with torch.inference_mode():
        plain = promonet_amd.preprocess.spectrogram.from_audio(
            audio.to(device), True)
    assert not plain.requires_grad


def test_loudness(device):
    import promonet_amd
    gen = torch.Generator().manual_seed(6)
    audio = torch.randn(1, 256 * 50, generator=gen) * .1
    audio[:, 256 * 20:256 * 30] *= 1e-5        # exercise the max - 80 dB floor
    for bands in (8, 1, None):
        want = oracle.loudness(audio, bands)
        got = promonet_amd.preprocess.loudness.from_audio(
            audio.to(device), bands)
        assert got.shape == want.shape
        # dB over a 100 dB range; measured 3e-5 (bands) / 2.7e-4 (per bin)
        assert ((got.cpu() - want).abs() <= 1e-4 + 1e-5 * want.abs()).all(), \
            bands
    # batched: each utterance keeps its own floor
    pair = torch.cat([audio, audio * .01])
    got = promonet_amd.preprocess.loudness.from_audio(pair.to(device), 8)
    for item in range(2):
        want = oracle.loudness(pair[item:item + 1], 8)
        assert ((got[item].cpu() - want).abs() <=
                1e-4 + 1e-5 * want.abs()).all()


def test_ragged_batch_is_exact(device, default_state):
    """Zero-padded utterances of different lengths in one batch: each equals
    its stand-alone synthesis bit for bit, tails are zero (lengths include 1
    frame and the full width)."""
    model = make_model(default_state, 'f16', device)
    lengths = [57, 1, 100, 33, 100, 8]
    frames = max(lengths)
    inputs = on(device, oracle.synthetic_inputs(len(lengths), frames, seed=31))
    for item, length in enumerate(lengths):       # garbage past the end
        for tensor in inputs[:4]:
            tensor[item, ..., length:] = 7.
    with torch.inference_mode():
        ragged = model(*inputs, None, lengths=lengths)
        for item, length in enumerate(lengths):
            single = model(
                *[t[item:item + 1, ..., :length] if t.ndim >= 2
                  else t[item:item + 1] for t in inputs], None)
            assert torch.equal(
                ragged[item, :, :length * 256], single[0]), item
            assert ragged[item, :, length * 256:].abs().max().item() == 0. \
                if length < frames else True
    # and against the CPU oracle for one of them
    want = oracle.generator_forward(
        *[t[3:4, ..., :33].cpu() if t.ndim >= 2 else t[3:4].cpu()
          for t in inputs], default_state)
    check(max_abs(ragged[3:4, :, :33 * 256], want), TIGHT['f16'],
          'ragged_vs_oracle:f16')


@pytest.mark.parametrize('dtype', ['bf16', 'f16', 'checkpoint', 'fp32'])
def test_walked_kernels_match_standalone_tiling(device, default_state, dtype):
    """A few long utterances in one batch run the walked whole-Block / MRF
    kernels (one workgroup per utterance segment, left halo carried through
    LDS; the 4-byte operand layouts of 'checkpoint' / 'fp32': the SKEWED
    whole-MRF walk of the last stage, conv_mrf_skew_kernel, carries through
    scratch); each utterance alone runs the stand-alone tiling. Same arithmetic
    per column, same order of the MRF sum: bit-identical, tails zero."""
    model = make_model(default_state, dtype, device)
    lengths = [2300, 700, 1500]
    frames = max(lengths)
    inputs = on(device, oracle.synthetic_inputs(len(lengths), frames, seed=53))
    with torch.inference_mode():
        batched = model(*inputs, None, lengths=lengths)
        for item, length in enumerate(lengths):
            single = model(
                *[t[item:item + 1, ..., :length] if t.ndim >= 2
                  else t[item:item + 1] for t in inputs], None)
            assert torch.equal(batched[item, :, :length * 256], single[0]), item
            if length < frames:
                assert batched[item, :, length * 256:].abs().max().item() == 0.


def test_files_to_files_batched(device, default_state, tmp_path):
    """Batched file path == the reference-style sequential path, file by file."""
    import promonet_amd
    import scipy.io.wavfile
    model = make_model(default_state, 'f16', device)
    promonet_amd.synthesize.set_model(model, device)
    lengths = [20, 45, 31]
    files = {key: [] for key in (
        'loudness', 'pitch', 'periodicity', 'ppg', 'batched', 'sequential')}
    for index, length in enumerate(lengths):
        inputs = oracle.synthetic_inputs(1, length, seed=40 + index)
        torch.save(inputs[0][0], tmp_path / f'{index}-loudness.pt')
        torch.save(inputs[1], tmp_path / f'{index}-pitch.pt')
        torch.save(inputs[2], tmp_path / f'{index}-periodicity.pt')
        torch.save(inputs[3][0], tmp_path / f'{index}-ppg.pt')
        for key in ('loudness', 'pitch', 'periodicity', 'ppg'):
            files[key].append(tmp_path / f'{index}-{key}.pt')
        files['batched'].append(tmp_path / 'batched' / f'{index}.wav')
        files['sequential'].append(tmp_path / 'sequential' / f'{index}.wav')
    args = [files[k] for k in ('loudness', 'pitch', 'periodicity', 'ppg')]
    promonet_amd.synthesize.from_files_to_files_batched(
        *args, files['batched'], speakers=[1, 2, 3], gpu=0, batch_size=2)
    promonet_amd.synthesize.from_files_to_files(
        *args, files['sequential'], speakers=[1, 2, 3], gpu=0)
    for a, b, length in zip(files['batched'], files['sequential'], lengths):
        rate_a, audio_a = scipy.io.wavfile.read(a)
        rate_b, audio_b = scipy.io.wavfile.read(b)
        assert rate_a == rate_b == 22050
        assert audio_a.shape == audio_b.shape == (length * 256,)
        assert (audio_a == audio_b).all()


def test_files_to_files_batched_worker_pipeline(device, default_state, tmp_path):
    """The worker-process IO pipeline of the batched file path (feature files
    unpickled and padded, wav files written by CPU processes while the GPU
    synthesises; the previous batch's audio leaves the device on a side
    stream): same bytes in every output file as the in-process loop, for a job
    large enough to take the pool (9 batches of 4, uneven last batch)."""
    import promonet_amd
    import scipy.io.wavfile
    model = make_model(default_state, 'bf16', device)
    promonet_amd.synthesize.set_model(model, device)
    gen = torch.Generator().manual_seed(11)
    lengths = [int(torch.randint(3, 60, (1,), generator=gen)) for _ in range(34)]
    files = {key: [] for key in (
        'loudness', 'pitch', 'periodicity', 'ppg', 'pool', 'serial')}
    for index, length in enumerate(lengths):
        inputs = oracle.synthetic_inputs(1, length, seed=400 + index)
        torch.save(inputs[0][0], tmp_path / f'{index}-loudness.pt')
        torch.save(inputs[1], tmp_path / f'{index}-pitch.pt')
        torch.save(inputs[2], tmp_path / f'{index}-periodicity.pt')
        torch.save(inputs[3][0], tmp_path / f'{index}-ppg.pt')
        for key in ('loudness', 'pitch', 'periodicity', 'ppg'):
            files[key].append(tmp_path / f'{index}-{key}.pt')
        files['pool'].append(tmp_path / 'pool' / f'{index}.wav')
        files['serial'].append(tmp_path / 'serial' / f'{index}.wav')
    args = [files[k] for k in ('loudness', 'pitch', 'periodicity', 'ppg')]
    speakers = [index % 7 for index in range(len(lengths))]
    promonet_amd.synthesize.from_files_to_files_batched(
        *args, files['pool'], speakers=speakers, gpu=0, batch_size=4,
        num_workers=3)
    promonet_amd.synthesize.from_files_to_files_batched(
        *args, files['serial'], speakers=speakers, gpu=0, batch_size=4,
        num_workers=0)
    for a, b, length in zip(files['pool'], files['serial'], lengths):
        rate_a, audio_a = scipy.io.wavfile.read(a)
        rate_b, audio_b = scipy.io.wavfile.read(b)
        assert rate_a == rate_b == 22050
        assert audio_a.shape == audio_b.shape == (length * 256,)
        assert (audio_a == audio_b).all()
    # configure() overrides reach the spawned workers (a fresh interpreter
    # imports the package with its defaults): the wav header carries the
    # parent's SAMPLE_RATE, and the pool is re-keyed on the configuration
    promonet_amd.configure(SAMPLE_RATE=16000)
    try:
        promonet_amd.synthesize.from_files_to_files_batched(
            *args, files['pool'], speakers=speakers, gpu=0, batch_size=4,
            num_workers=3)
        for a, b, length in zip(files['pool'], files['serial'], lengths):
            rate_a, audio_a = scipy.io.wavfile.read(a)
            _, audio_b = scipy.io.wavfile.read(b)
            assert rate_a == 16000
            assert (audio_a == audio_b).all()
    finally:
        promonet_amd.configure(SAMPLE_RATE=22050)
    # an empty job with a live pool is a no-op
    promonet_amd.synthesize.from_files_to_files_batched(
        [], [], [], [], [], gpu=0, num_workers=3)
    promonet_amd.synthesize.shutdown_workers()


def test_packed_interface(device, golden_default, default_state):
    """pack_features / unpack_features / packed_inference (the nn~ buffer
    contract, generator.py:255-422) against the real reference's output."""
    model = make_model(default_state, 'fp32', device)
    entry = golden_default['packed']
    inputs = oracle.synthetic_inputs(2, entry['frames'], seed=entry['input_seed'])
    with torch.inference_mode():
        packed = model.pack_features(
            inputs[0].to(device), inputs[1][:, None].to(device),
            inputs[2][:, None].to(device), *on(device, inputs[3:]))
        assert packed.shape == entry['packed'].shape == (2, 53, entry['frames'])
        # band means of ~-100 dB values: one fp32 ulp is 7.6e-6
        assert max_abs(packed, entry['packed']) < 2e-5
        audio = model.packed_inference(entry['packed'].to(device))
        unpacked = model.unpack_features(entry['packed'].to(device))
    assert audio.dtype == torch.float32
    assert audio.shape == entry['audio'].shape
    check(max_abs(audio, entry['audio']), TIGHT['fp32'], 'packed_interface:fp32')
    assert len(model.labels()) == entry['labels'] == 53
    assert torch.equal(unpacked[4].cpu(), inputs[4])
    # the export-time self test of the reference (generator.py:363-368)
    with torch.inference_mode():
        zeros = model.packed_inference(torch.zeros(1, 53, 32, device=device))
    assert tuple(zeros.shape) == (1, 1, 8192) and torch.isfinite(zeros).all()


@pytest.mark.parametrize('dtype', ['checkpoint', 'bf16'])
def test_packed_inference_graph(device, default_state, dtype):
    """The low-latency schedule of the streaming (nn~) use, generator.py:
    334-343,363-368: `packed_inference(x, graph=True)` replays one captured
    hipGraph per input shape. Bit-identical to the eager call at the nn~ chunk
    sizes, for inputs other than the captured ones; the graph owns its
    workspace (a larger eager forward in between does not disturb it) and is
    re-captured when the weights change."""
    model = make_model(default_state, dtype, device)
    gen = torch.Generator().manual_seed(77)

    def packed(batch, frames):
        inputs = oracle.synthetic_inputs(batch, frames, seed=frames + batch)
        with torch.inference_mode():
            return model.pack_features(
                inputs[0].to(device), inputs[1][:, None].to(device),
                inputs[2][:, None].to(device), *on(device, inputs[3:])).clone()

    for frames in (8, 16, 32, 64):
        first, second = packed(1, frames), packed(1, frames + 1)[..., :frames]
        with torch.inference_mode():
            eager = [model.packed_inference(t) for t in (first, second)]
        replay = [model.packed_inference(t, graph=True) for t in (first, second)]
        assert replay[0].shape == (1, 1, frames * 256)
        for want, got in zip(eager, replay):
            assert torch.equal(want, got), frames
        assert not torch.equal(replay[0], replay[1])
    assert len(model._packed_graphs) == 4
    # under inference mode too, and a batch of 2
    pair = packed(2, 16)
    with torch.inference_mode():
        assert torch.equal(
            model.packed_inference(pair, graph=True),
            model.packed_inference(pair))
    # a shape captured INSIDE inference mode replays outside it and the other
    # way round (the static input of a graph is updated in place by both)
    inside, outside = packed(1, 24), packed(1, 40)
    with torch.inference_mode():
        first_inside = model.packed_inference(inside, graph=True)
        want_outside = model.packed_inference(outside)
    assert torch.equal(
        model.packed_inference(inside.clone(), graph=True), first_inside)
    first_outside = model.packed_inference(outside.clone(), graph=True)
    with torch.inference_mode():
        assert torch.equal(
            model.packed_inference(outside, graph=True), first_outside)
    assert torch.equal(first_outside, want_outside)
    # a much larger eager forward re-allocates the module's shared workspace
    chunk = packed(1, 32)
    before = model.packed_inference(chunk, graph=True)
    with torch.inference_mode():
        model.packed_inference(packed(3, 500))
    torch.cuda.synchronize()
    assert torch.equal(model.packed_inference(chunk, graph=True), before)
    # new weights: the engine is rebuilt, stale graphs are dropped
    state = {k: v.clone() for k, v in default_state.items()}
    state['model.model.5.weight'] = state['model.model.5.weight'] * .5
    model.load_state_dict(state)
    with torch.inference_mode():
        want = model.packed_inference(chunk)
    got = model.packed_inference(chunk, graph=True)
    assert torch.equal(want, got) and not torch.equal(got, before)
    assert len(model._packed_graphs) == 1
    # the reference's export-time self test shape (generator.py:363-368)
    zeros = model.packed_inference(
        torch.zeros(1, 53, 32, device=device), graph=True)
    assert tuple(zeros.shape) == (1, 1, 8192) and torch.isfinite(zeros).all()


@pytest.mark.parametrize('dtype', ['bf16', 'checkpoint'])
def test_full_size_shift_equivariance(device, default_state, dtype):
    """A size-independent property at BASELINE.json's full size (batch 32 x
    861 frames), no oracle needed: the generator is convolutional
    (hifigan.py:63-70; prepare_features is per frame), so conditioning shifted
    by k frames gives the same audio shifted by 256 k samples - BIT FOR BIT
    away from the edges (the receptive field is 14 frames a side),
    whatever tile, segment or walk step a column falls into after the shift:
    every tiling computes a column with the same arithmetic."""
    model = make_model(default_state, dtype, device)
    frames, shift, field = 861, 37, 16
    inputs = on(device, oracle.synthetic_inputs(32, frames + shift, seed=77))
    early = [t[..., :frames].contiguous() if t.ndim >= 2 else t for t in inputs]
    late = [t[..., shift:].contiguous() if t.ndim >= 2 else t for t in inputs]
    with torch.inference_mode():
        a = model(*early, None)
        b = model(*late, None)
    assert a.shape == b.shape == (32, 1, frames * 256)
    # frames [shift + field, frames - field) of `early` = frames
    # [field, frames - shift - field) of `late`
    lo, hi = (shift + field) * 256, (frames - field) * 256
    same = torch.equal(a[..., lo:hi], b[..., lo - shift * 256:hi - shift * 256])
    worst = (a[..., lo:hi] - b[..., lo - shift * 256:hi - shift * 256]).abs().max()
    print(f'shift equivariance {dtype}: interior of {hi - lo} samples x 32 '
          f'utterances, max difference {worst.item():.3e}')
    assert same
    # and the edges DO differ (the test would pass trivially on constant audio)
    assert not torch.equal(a[..., :lo], b[..., :lo])


def test_full_size_stft_parseval(device):
    """Size-independent property of the STFT at the benchmarked size, against
    nothing but the input: Parseval for the real 1024-point transform of a
    windowed frame, sum_n (w x)^2 = (|X_0|^2 + |X_512|^2 + 2 sum_{0<k<512}
    |X_k|^2) / 1024, for every one of the 27 552 frames (the kernel's
    magnitude is sqrt(|X|^2 + 1e-6): removed before summing)."""
    import promonet_amd
    gen = torch.Generator().manual_seed(123)
    batch, frames = 32, 861
    audio = torch.randn(batch, frames * 256, generator=gen) * .1
    audio *= (10. ** (-(torch.arange(batch) % 5) / 2.))[:, None]
    spec = promonet_amd.preprocess.spectrogram.from_audio(
        audio.to(device)[:, None]).double().cpu()
    power = spec ** 2 - 1e-6
    weight = torch.full((513, 1), 2., dtype=torch.float64)
    weight[0] = weight[512] = 1.
    spectral = (power * weight).sum(1) / 1024.
    padded = torch.nn.functional.pad(
        audio.double()[:, None], (384, 384), mode='reflect')[:, 0]
    window = torch.hann_window(1024, dtype=torch.float64)
    temporal = (padded.unfold(-1, 1024, 256) * window).pow(2).sum(-1)
    assert spectral.shape == temporal.shape == (batch, frames)
    relative = ((spectral - temporal).abs() / temporal).max().item()
    print(f'Parseval over {batch * frames} frames: worst relative {relative:.3e}')
    # (measured 1.4e-7)
    check(relative, 5e-7, 'stft_parseval_full_size')
