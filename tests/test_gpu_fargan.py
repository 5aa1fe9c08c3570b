"""GPU parity of the FARGAN path (config/fargan.py) against goldens computed
by the REAL reference and against the CPU oracle on longer sequences."""
import pytest
import torch

import restatement as oracle
from util import check, max_abs

pytestmark = pytest.mark.gpu

# fp32 math everywhere; f16 / mixed only change the STORAGE of the streamed
# weights ('mixed': GRU cells and GLU gates f16, the other layers fp32).
# Gates <= 3x what MI355X measures (profiles/r04/pytest_fargan_mixed.log:
# fp32 3.9e-7, f16 7.6e-5, mixed 6.0e-6 on the reference's goldens)
GATE = {'fp32': 1.2e-6, 'f16': 2.3e-4, 'mixed': 1.8e-5}


@pytest.fixture()
def fargan_model(golden_fargan, golden_default, device):
    import promonet_amd
    state = oracle.random_state_fargan(seed=golden_fargan['seed'])
    state['pitch_distribution'] = golden_default['pitch_distribution'].clone()
    models = {}

    def build(dtype):
        if dtype not in models:
            promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=dtype)
            try:
                model = promonet_amd.model.Generator()
                model.load_state_dict(state)
                models[dtype] = model.to(device).eval()
            finally:
                promonet_amd.configure(
                    MODEL='hifigan', FARGAN_WEIGHT_DTYPE='fp32')
        return models[dtype]
    build.state = state
    return build


def on(device, inputs):
    return [t.to(device) for t in inputs]


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'mixed'])
def test_matches_reference_golden(
    device, golden_fargan, fargan_model, dtype, mode
):
    """mode 1: one persistent workgroup per utterance; mode 2: clusters of 8
    workgroups exchanging layer outputs through global memory."""
    model = fargan_model(dtype)
    model.model.kernel_mode = mode
    for name in ('b2_t8', 'b1_t60', 'b3_t25'):
        entry = golden_fargan[name]
        inputs = oracle.synthetic_inputs(
            entry['batch'], entry['frames'], seed=entry['input_seed'])
        with torch.inference_mode():
            got = model(*on(device, inputs), entry['previous'].to(device))
            features = model.prepare_features(*on(device, inputs[:4]))
        assert got.shape == entry['audio'].shape
        assert features.shape[1] == 114
        assert max_abs(features[:, -1], entry['period']) < 1e-4
        error = max_abs(got, entry['audio'])
        print(f'fargan {dtype} {name}: max-abs {error:.3e} '
              f'(abs-max {entry["audio"].abs().max().item():.3f})')
        check(error, GATE[dtype], f'fargan_golden:{dtype}', name)
    model.model.kernel_mode = 0


@pytest.mark.parametrize(
    'batch,dtype', [(37, 'fp32'), (70, 'fp32'), (135, 'fp32'), (70, 'mixed')])
def test_cluster_waves_and_determinism(device, fargan_model, batch, dtype):
    """More utterances than clusters: 2 (batch 37, one padded slot) or 4
    (batch 70, two padded slots) utterances advance in lockstep per cluster,
    and at 135 the 32 clusters walk a second wave with their epochs still
    counting. Repeated runs bit-identical, and the two kernels agree."""
    model = fargan_model(dtype)
    inputs = on(device, oracle.synthetic_inputs(batch, 6, seed=15))
    with torch.inference_mode():
        model.model.kernel_mode = 2
        clustered = model(*inputs, None)
        again = model(*inputs, None)
        model.model.kernel_mode = 1
        single = model(*inputs, None)
        model.model.kernel_mode = 2
        alone = model(*[t[5:6] for t in inputs], None)
    model.model.kernel_mode = 0
    assert torch.equal(clustered, again)
    assert max_abs(clustered, single) < 2e-6
    # a lockstep group's utterance == its stand-alone run (one utterance per
    # cluster: LDS-resident short slices, eight-member polls in one round)
    assert torch.equal(clustered[5], alone[0])


def test_long_sequence_vs_oracle(device, fargan_model):
    """Autoregression over 2 s (688 dependent steps) stays on the oracle's
    trajectory; batch-1 previous samples / globals broadcast."""
    model = fargan_model('fp32')
    inputs = oracle.synthetic_inputs(2, 172, seed=9)
    with torch.inference_mode():
        want = oracle.fargan_generator_forward(*inputs, fargan_model.state)
        got = model(*on(device, inputs), None)
    assert got.shape == (2, 1, 172 * 256)
    # (measured 3e-7)
    check(max_abs(got, want), 1e-6, 'fargan_long_sequence:fp32')


# whole-utterance max-abs gates, <= 3x the measured values (fp32 2.7e-7, f16
# 6.6e-5, mixed 6.5e-6 over 3 444 dependent steps; three other weight seeds
# read the same, profiles/r04/fargan_storage_seeds.txt). The north star's 1e-4
# holds for fp32 and mixed storage with a 15x margin; f16 STORAGE of every
# weight sits at 0.7 of it, which is why it is not a default.
FULL_GATE = {'fp32': 1e-6, 'f16': 1e-4, 'mixed': 2e-5}
_CONFIG5 = {}   # the oracle's audio of the two checked utterances


@pytest.mark.parametrize(
    'dtype,mode',
    [('fp32', 1), ('fp32', 2), ('f16', 2), ('mixed', 2), ('mixed', 1)])
def test_full_size_config5(device, fargan_model, dtype, mode):
    """BASELINE.json configs[4] at full size: batch 32 x 861 frames (3 444
    dependent sub-frame steps - where autoregressive drift would show).
    Utterances 3 and 30 are checked over their WHOLE length against the CPU
    oracle at the 1e-4 gate, in both kernel modes; the rest through the
    size-independent property (modes agree, runs are deterministic). With
    f16-STORED weights (fp32 arithmetic) the error is 7e-5 and does not grow
    along the 3 444 autoregressive steps: same gate."""
    model = fargan_model(dtype)
    inputs = oracle.synthetic_inputs(32, 861, seed=55)
    pick = [3, 30]
    with torch.inference_mode():
        # (one oracle run serves every parametrisation: same inputs, same
        # weights; 3 444 dependent steps of small mat-vecs - a few torch
        # threads, the default of one per host core costs 15x the time)
        if 'want' not in _CONFIG5:
            threads = torch.get_num_threads()
            torch.set_num_threads(min(threads, 8))
            try:
                _CONFIG5['want'] = oracle.fargan_generator_forward(
                    *[t[pick] for t in inputs], fargan_model.state)
            finally:
                torch.set_num_threads(threads)
        want = _CONFIG5['want']
        model.model.kernel_mode = mode
        got = model(*on(device, inputs), None)
        model.model.kernel_mode = 0
    assert got.shape == (32, 1, 861 * 256) and torch.isfinite(got).all()
    error = max_abs(got[pick], want)
    tail = max_abs(got[pick][..., -256 * 40:], want[..., -256 * 40:])
    print(f'fargan full size {dtype} mode {mode}: max-abs {error:.3e} (last 40 '
          f'frames {tail:.3e}; abs-max {want.abs().max().item():.3f})')
    check(error, FULL_GATE[dtype], f'fargan_full_size:{dtype}', mode)


def test_module_seam(device, fargan_model):
    """FARGAN.forward(features (B,114,T), global (B,258,1), previous)."""
    model = fargan_model('fp32')
    inputs = oracle.synthetic_inputs(2, 12, seed=10)
    state = fargan_model.state
    features = oracle.prepare_features(
        *inputs[:4], state['pitch_distribution'],
        state['pitch_embedding.weight'], state['ppg_threshold'])
    period = oracle.SAMPLE_RATE / torch.clip(inputs[1], 50., 550.)
    features = torch.cat((features, period[:, None]), dim=1)
    glob = oracle.prepare_global_features(
        *inputs[4:7], state['speaker_embedding.weight'])
    previous = torch.zeros(2, 1, 512)
    want = oracle.fargan_forward(features, glob, previous, state)
    with torch.inference_mode():
        got = model.model(
            features.to(device), glob.to(device), previous.to(device))
    check(max_abs(got, want), 1.5e-6, 'fargan_module_seam:fp32')
    with pytest.raises(ValueError):
        model.model(features[:, :-1].to(device), glob.to(device), None)


def test_deterministic_and_batch_independent(device, fargan_model):
    model = fargan_model('fp32')
    inputs = on(device, oracle.synthetic_inputs(5, 40, seed=12))
    with torch.inference_mode():
        full = model(*inputs, None)
        again = model(*inputs, None)
        single = model(*[t[3:4] for t in inputs], None)
    assert torch.equal(full, again)
    assert torch.equal(single[0], full[3])


@pytest.mark.parametrize(
    'dtype,mode', [('fp32', 1), ('fp32', 2), ('mixed', 2), ('f16', 2)])
def test_ragged_batch_is_exact(device, fargan_model, dtype, mode):
    """Zero-padded utterances of different lengths in one batch (more of them
    than clusters, so several advance in lockstep): the model is causal, each
    equals its stand-alone synthesis bit for bit, the tails are zero. With
    'mixed' / f16 storage the stand-alone run (one utterance per cluster)
    reads its short weight slices from their LDS-resident copies, the
    lockstep batch streams them: same bits."""
    model = fargan_model(dtype)
    lengths = [9, 1, 14, 5, 14, 3] * 6 + [7]          # 37 utterances
    frames = max(lengths)
    inputs = on(device, oracle.synthetic_inputs(len(lengths), frames, seed=33))
    for item, length in enumerate(lengths):           # garbage past the end
        for tensor in inputs[:4]:
            tensor[item, ..., length:] = 7.
    model.model.kernel_mode = mode
    with torch.inference_mode():
        ragged = model(*inputs, None, lengths=lengths)
        for item in (0, 1, 2, 5, 36):
            length = lengths[item]
            single = model(
                *[t[item:item + 1, ..., :length] if t.ndim >= 2
                  else t[item:item + 1] for t in inputs], None)
            assert torch.equal(ragged[item, :, :length * 256], single[0]), item
            if length < frames:
                assert ragged[item, :, length * 256:].abs().max().item() == 0.
    model.model.kernel_mode = 0


def test_auto_mode_timeout_fallback_and_rearm(
    device, fargan_model, monkeypatch
):
    """Auto mode (kernel_mode 0) on a GPU that cannot keep the clusters
    resident: the bounded exchange reports PM_ETIMEOUT (injected here through
    pm_fargan_check while the cluster kernel is selected), the module retries
    once, falls back to the one-workgroup-per-utterance kernel for RETRY_AFTER
    calls and then tries the clusters again INSIDE the guarded path - the call
    on which the counter reaches zero must not raise."""
    import warnings
    from promonet_amd import _lib
    model = fargan_model('fp32')
    fargan = model.model
    inputs = on(device, oracle.synthetic_inputs(2, 6, seed=21))
    with torch.inference_mode():
        want = model(*inputs, None)
    real = _lib.lib()
    state = {'mode': 0, 'cluster_checks': 0, 'fail': True}

    class Shim:
        def __getattr__(self, name):
            return getattr(real, name)

        def pm_fargan_set_mode(self, engine, mode):
            state['mode'] = mode
            return real.pm_fargan_set_mode(engine, mode)

        def pm_fargan_check(self, *args):
            code = real.pm_fargan_check(*args)
            if state['mode'] != 1:
                state['cluster_checks'] += 1
                if state['fail']:
                    return _lib.PM_ETIMEOUT
            return code

    monkeypatch.setattr(_lib, 'lib', lambda: Shim())
    fargan.kernel_mode, fargan._fallback_calls = 0, 0
    monkeypatch.setattr(type(fargan), 'RETRY_AFTER', 3)
    try:
        with torch.inference_mode(), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            got = model(*inputs, None)              # 2 cluster tries, fallback
            assert state['cluster_checks'] == 2 and fargan._fallback_calls == 3
            assert max_abs(got, want) < 5e-6
            for left in (2, 1, 0):                  # fallback kernel, no raise
                got = model(*inputs, None)
                assert fargan._fallback_calls == left
                assert state['cluster_checks'] == 2
            # counter at zero, GPU still shared: guarded retry, re-armed
            got = model(*inputs, None)
            assert state['cluster_checks'] == 4 and fargan._fallback_calls == 3
            assert max_abs(got, want) < 5e-6
            state['fail'] = False
            for _ in range(3):
                model(*inputs, None)
            got = model(*inputs, None)              # clusters again, they hold
            assert state['cluster_checks'] == 5 and fargan._fallback_calls == 0
            assert max_abs(got, want) < 5e-6
    finally:
        fargan.kernel_mode, fargan._fallback_calls = 0, 0
