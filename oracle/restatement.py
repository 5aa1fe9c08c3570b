"""CPU oracle: a restatement of promonet's synthesis hot path in plain PyTorch.

TEST INFRASTRUCTURE ONLY. Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module. The product
(`promonet_amd`) never does: its compute runs in `libpromonet_hip.so`.

Every function cites the reference file:line it follows (paths relative to
`/root/reference`). Pinning status:

* Generator / HiFi-GAN / prepare_features / prepare_global_features /
  spectrogram.from_audio: PINNED. `oracle/make_golden.py` imports the real
  reference in the build container, checks this restatement against it
  (<= 1e-6) and commits the vectors under `tests/golden/`.
* `ppgs.sparsify` (third-party `ppgs`, unpinned version, `setup.py:21`),
  `librosa.filters.mel`, `librosa.stft`, `librosa.amplitude_to_db`,
  `librosa.A_weighting` (third-party `librosa`, unpinned, `setup.py:17`):
  PARITY UNPINNED. Neither package is installed or vendored; the functions
  below restate their published algorithms and are anchored on the
  reference's call sites only. Second sources (tests/test_cpu_oracle.py):
  numpy.quantile, scipy.signal.freqs + the IEC 61672 table, and
  transformers.audio_utils (mel filter bank, STFT, amplitude_to_db).

All math is fp32 on CPU (`torch` ATen), the same op sequence the reference
executes: un-fused conv1d / conv_transpose1d / leaky_relu / add.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

###############################################################################
# Constants (promonet/config/defaults.py, promonet/config/static.py)
###############################################################################

SAMPLE_RATE = 22050          # defaults.py:49
HOPSIZE = 256                # defaults.py:31
NUM_FFT = 1024               # defaults.py:43
WINDOW_SIZE = 1024           # defaults.py:52
NUM_MELS = 80                # defaults.py:40
FMIN = 50.                   # defaults.py:27
FMAX = 550.                  # defaults.py:28
MIN_DB = -100.               # defaults.py:37
REF_DB = 20.                 # defaults.py:46
LOUDNESS_BANDS = 8           # defaults.py:90
PITCH_BINS = 256             # defaults.py:96
PITCH_EMBEDDING_SIZE = 64    # defaults.py:99
PPG_CHANNELS = 40
SPARSE_PPG_THRESHOLD = 0.85  # defaults.py:117
LRELU_SLOPE = .1             # defaults.py:216
RESBLOCK_KERNEL_SIZES = (3, 7, 11)             # defaults.py:250
RESBLOCK_DILATION_SIZES = ((1, 3, 5),) * 3     # defaults.py:253
UPSAMPLE_INITIAL_SIZE = 512                    # defaults.py:256
UPSAMPLE_KERNEL_SIZES = (16, 16, 4, 4)         # defaults.py:259
UPSAMPLE_RATES = (8, 8, 2, 2)                  # defaults.py:262
SPEAKER_CHANNELS = 256                         # defaults.py:265
NUM_SPEAKERS = 109                             # static.py:62
NUM_FEATURES = 113                             # static.py:48-53
GLOBAL_CHANNELS = 258                          # static.py:42-45


###############################################################################
# Feature preparation
###############################################################################


def sparsify(ppg, method='percentile', threshold=SPARSE_PPG_THRESHOLD):
    """`ppgs.sparsify` (third-party; called at model/generator.py:144-147).

    PARITY UNPINNED: restated from the published interactiveaudiolab/ppgs
    package. threshold = per-frame quantile (linear interpolation) over the
    channel axis; entries <= threshold are zeroed; renormalise with
    softmax(log(p + 1e-8)).
    """
    if not torch.is_tensor(threshold):
        threshold = torch.tensor(threshold, dtype=ppg.dtype)
    if method == 'percentile':
        q = torch.quantile(ppg, threshold.to(ppg.dtype), dim=-2, keepdim=True)
        ppg = torch.where(ppg > q, ppg, torch.zeros_like(ppg))
    elif method == 'constant':
        ppg = torch.where(ppg > threshold, ppg, torch.zeros_like(ppg))
    elif method == 'topk':
        # keep the k largest entries per frame (ties: lower channel first,
        # a stable descending sort), zero the rest
        order = torch.sort(ppg, dim=-2, descending=True, stable=True).indices
        keep = torch.zeros_like(ppg, dtype=torch.bool)
        keep.scatter_(-2, order[..., :int(threshold), :], True)
        ppg = torch.where(keep, ppg, torch.zeros_like(ppg))
    else:
        raise ValueError(f'Sparsify method {method} is not defined')
    return torch.softmax(torch.log(ppg + 1e-8), -2)


def band_average(loudness, bands=LOUDNESS_BANDS):
    """preprocess/loudness.py:84-111 (and model/generator.py:172-180)."""
    if bands is None:
        return loudness
    if bands == 1:
        return loudness.mean(dim=-2, keepdim=True)
    step = loudness.shape[-2] / bands
    rows = [
        loudness[..., int(b * step):int((b + 1) * step), :].mean(dim=-2)
        for b in range(int(bands))]
    return torch.stack(rows, dim=-2)


def normalize(loudness):
    """preprocess/loudness.py:144-146."""
    return (loudness - MIN_DB) / (REF_DB - MIN_DB)


def pitch_bins(pitch, pitch_distribution):
    """model/generator.py:152-157: clip, searchsorted (right=False), clip."""
    hz = torch.clip(pitch, FMIN, FMAX)
    bins = torch.searchsorted(pitch_distribution, hz)
    return torch.clip(bins, 0, PITCH_BINS - 1)


def prepare_features(
    loudness, pitch, periodicity, ppg, pitch_distribution, pitch_embedding,
    ppg_threshold=SPARSE_PPG_THRESHOLD, sparse_method='percentile'
):
    """model/generator.py:137-197 (hifigan branch; SPARSE_PPG_METHOD
    'percentile' is the default configuration, None skips sparsify :140-147).

    loudness (B, 8|513, T) dB; pitch (B, T) Hz; periodicity (B, T);
    ppg (B, 40, T). Returns (B, 113, T) =
    [ppg 0:40 | pitch-embedding 40:104 | loudness 104:112 | periodicity 112].
    """
    features = ppg if sparse_method is None else sparsify(
        ppg, sparse_method, ppg_threshold)
    bins = pitch_bins(pitch, pitch_distribution)
    embedded = F.embedding(bins, pitch_embedding).permute(0, 2, 1)
    features = torch.cat((features, embedded), dim=1)
    averaged = band_average(loudness, LOUDNESS_BANDS)
    normalized = normalize(averaged)
    if normalized.ndim == 2:
        normalized = normalized[None]
    features = torch.cat((features, normalized), dim=1)
    return torch.cat((features, periodicity[:, None]), dim=1)


def prepare_global_features(
    speakers, spectral_balance_ratios, loudness_ratios, speaker_embedding,
    speaker_bias=None, augment_pitch=True, augment_loudness=True
):
    """model/generator.py:49-70. Default: Embedding lookup + both ratios;
    with `speaker_bias` the ZERO_SHOT Linear over x-vectors (:35-38)."""
    if speaker_bias is None:
        g = F.embedding(speakers, speaker_embedding).unsqueeze(-1)
    else:
        g = F.linear(speakers, speaker_embedding, speaker_bias).unsqueeze(-1)
    if augment_pitch:
        g = torch.cat((g, spectral_balance_ratios[:, None, None]), dim=1)
    if augment_loudness:
        g = torch.cat((g, loudness_ratios[:, None, None]), dim=1)
    return g


###############################################################################
# HiFi-GAN generator from a reference-keyed state dict
###############################################################################


def get_padding(kernel_size, dilation=1, stride=1):
    """model/core.py:9-11."""
    return int((kernel_size * dilation - dilation - stride + 1) / 2)


def fold_weight_norm(g, v):
    """torch.nn.utils.weight_norm, dim=0: w = g * v / ||v|| over dims != 0.

    For ConvTranspose1d dim 0 is the INPUT channel (weight_g (C_in, 1, 1)).
    Reference: model/core.py:43-45, model/hifigan.py:100-106 (hooks recompute
    this every forward; model/generator.py:203-206 removes them on export).
    """
    norm = torch.linalg.vector_norm(v, ord=2, dim=(1, 2), keepdim=True)
    return v * (g / norm)


def folded_state(state):
    """Return {name: tensor} with every weight_g/weight_v pair folded."""
    out = {}
    for key, value in state.items():
        if key.endswith('.weight_g'):
            base = key[:-len('.weight_g')]
            out[base + '.weight'] = fold_weight_norm(
                value, state[base + '.weight_v'])
        elif key.endswith('.weight_v'):
            continue
        else:
            out[key] = value
    return out


def hifigan_config(state):
    """Derive (initial_channels, rates, kernel sizes) from tensor shapes."""
    initial = state['model.input_feature_conv.weight'].shape[0]
    rates, kernels = [], []
    i = 0
    while f'model.model.{i}.model.1.bias' in state:
        key = f'model.model.{i}.model.1.weight'
        w = state[key] if key in state else state[key + '_v']
        kernels.append(w.shape[2])
        i += 1
    # stride is not recoverable from shapes alone; k == 2 * r in every
    # reference config (defaults.py:259-262)
    rates = [k // 2 for k in kernels]
    return initial, tuple(rates), tuple(kernels)


def block(x, w, prefix, kernel_size, dilations, trace=None):
    """model/hifigan.py:198-210."""
    for n, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(
            xt, w[f'{prefix}.convs1.{n}.weight'], w[f'{prefix}.convs1.{n}.bias'],
            padding=get_padding(kernel_size, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(
            xt, w[f'{prefix}.convs2.{n}.weight'], w[f'{prefix}.convs2.{n}.bias'],
            padding=get_padding(kernel_size, 1))
        x = xt + x
        if trace is not None:
            trace[f'{prefix}.iter{n}'] = x
    return x


def residual_block(x, w, prefix, trace=None):
    """model/hifigan.py:141-145: sum of the three Blocks / 3."""
    xs = None
    for j, (k, d) in enumerate(
            zip(RESBLOCK_KERNEL_SIZES, RESBLOCK_DILATION_SIZES)):
        y = block(x, w, f'{prefix}.model.{j}', k, d, trace)
        xs = y if xs is None else xs + y
    return xs / len(RESBLOCK_KERNEL_SIZES)


def hifigan_forward(features, global_features, state, trace=None):
    """model/hifigan.py:63-70 with weights from a reference-keyed state dict.

    `state` keys are prefixed `model.` (the Generator's attribute name for
    the vocoder, model/generator.py:22). Accepts weight-normed or folded
    tensors. features (B, 113, T), global_features (B|1, 258, 1) ->
    (B, 1, 256 T).
    """
    w = folded_state(state)
    _, rates, kernels = hifigan_config(w)
    x = F.conv1d(
        features, w['model.input_feature_conv.weight'],
        w['model.input_feature_conv.bias'], padding=3)
    x = x + F.conv1d(
        global_features, w['model.input_speaker_conv.weight'],
        w['model.input_speaker_conv.bias'])
    if trace is not None:
        trace['input'] = x
    for i, (r, k) in enumerate(zip(rates, kernels)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(
            x, w[f'model.model.{i}.model.1.weight'],
            w[f'model.model.{i}.model.1.bias'], stride=r,
            padding=(k - r) // 2)
        if trace is not None:
            trace[f'upsample{i}'] = x
        x = residual_block(x, w, f'model.model.{i}.model.2', trace)
        if trace is not None:
            trace[f'mrf{i}'] = x
    n = len(rates)
    x = F.leaky_relu(x, LRELU_SLOPE)
    x = F.conv1d(x, w[f'model.model.{n + 1}.weight'], None, padding=3)
    return torch.tanh(x)


def generator_forward(
    loudness, pitch, periodicity, ppg, speakers, spectral_balance_ratios,
    loudness_ratios, state, trace=None
):
    """model/generator.py:116-135 (`previous_samples` is unused by HiFi-GAN,
    model/hifigan.py:63)."""
    features = prepare_features(
        loudness, pitch, periodicity, ppg, state['pitch_distribution'],
        state['pitch_embedding.weight'], state['ppg_threshold'])
    global_features = prepare_global_features(
        speakers, spectral_balance_ratios, loudness_ratios,
        state['speaker_embedding.weight'])
    if trace is not None:
        trace['features'] = features
        trace['global_features'] = global_features
    return hifigan_forward(features, global_features, state, trace)


def from_features(
    loudness, pitch, periodicity, ppg, state, speaker=0,
    spectral_balance_ratio=1., loudness_ratio=1.
):
    """synthesize/core.py:18-59 + generate :209-281 on CPU: returns item 0
    of the batch, shape (1, 256 T), float32."""
    if loudness.ndim == 2:
        loudness = loudness[None]
    speakers = torch.full((1,), speaker, dtype=torch.long)
    sbr = torch.tensor([spectral_balance_ratio], dtype=torch.float)
    lr = torch.tensor([loudness_ratio], dtype=torch.float)
    with torch.inference_mode():
        return generator_forward(
            loudness, pitch, periodicity, ppg, speakers, sbr, lr,
            state)[0].to(torch.float32)


###############################################################################
# Random-init weights exactly as the reference constructs them
###############################################################################


def random_state(seed=0, initial_channels=UPSAMPLE_INITIAL_SIZE,
                 pitch_distribution=None, folded=False):
    """Build a reference-keyed state dict with the reference's init scheme.

    NOT bit-identical to `torch.manual_seed(seed); Generator()` (module
    construction order consumes the RNG differently); the goldens that need
    the reference's exact tensors carry them. Distributions follow
    model/hifigan.py:19-61,220-223 (weight-normed convs: v ~ N(0, 0.01),
    g = ||v||; torch defaults elsewhere).
    """
    gen = torch.Generator().manual_seed(seed)

    def normal(*shape, std=1.):
        return torch.randn(*shape, generator=gen) * std

    def uniform(*shape, bound):
        return (torch.rand(*shape, generator=gen) * 2 - 1) * bound

    state = {}
    state['default_previous_samples'] = torch.zeros(1, 1, 1)
    state['ppg_threshold'] = torch.tensor(SPARSE_PPG_THRESHOLD)
    if pitch_distribution is None:
        pitch_distribution = torch.exp(torch.linspace(
            math.log(55.9431), math.log(547.3264), PITCH_BINS))
    state['pitch_distribution'] = pitch_distribution.clone().float()
    state['speaker_embedding.weight'] = normal(NUM_SPEAKERS, SPEAKER_CHANNELS)
    state['pitch_embedding.weight'] = normal(PITCH_BINS, PITCH_EMBEDDING_SIZE)

    c0 = initial_channels
    bound = 1 / math.sqrt(NUM_FEATURES * 7)
    state['model.input_feature_conv.weight'] = uniform(
        c0, NUM_FEATURES, 7, bound=bound)
    state['model.input_feature_conv.bias'] = uniform(c0, bound=bound)
    bound = 1 / math.sqrt(GLOBAL_CHANNELS)
    state['model.input_speaker_conv.weight'] = uniform(
        c0, GLOBAL_CHANNELS, 1, bound=bound)
    state['model.input_speaker_conv.bias'] = uniform(c0, bound=bound)

    def weight_normed(prefix, v, bias_bound):
        g = torch.linalg.vector_norm(v, ord=2, dim=(1, 2), keepdim=True)
        state[prefix + '.bias'] = uniform(v.shape[0] if 'convs' in prefix
                                          else v.shape[1], bound=bias_bound)
        state[prefix + '.weight_g'] = g
        state[prefix + '.weight_v'] = v

    for i, (r, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNEL_SIZES)):
        cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
        # ConvTranspose1d fan_in = weight.size(1) * k = cout * k
        weight_normed(
            f'model.model.{i}.model.1', normal(cin, cout, k, std=.01),
            1 / math.sqrt(cout * k))
        for j, ks in enumerate(RESBLOCK_KERNEL_SIZES):
            for name in ('convs1', 'convs2'):
                for n in range(3):
                    weight_normed(
                        f'model.model.{i}.model.2.model.{j}.{name}.{n}',
                        normal(cout, cout, ks, std=.01),
                        1 / math.sqrt(cout * ks))
    n = len(UPSAMPLE_RATES)
    cl = c0 // 2 ** n
    state[f'model.model.{n + 1}.weight'] = uniform(
        1, cl, 7, bound=1 / math.sqrt(cl * 7))
    return folded_state(state) if folded else state


def synthetic_inputs(batch, frames, seed=1234, loudness_rows=8):
    """SURVEY.md section 8(d) / BASELINE.md section 4 synthetic workload."""
    gen = torch.Generator().manual_seed(seed)
    loudness = torch.rand(
        batch, loudness_rows, frames, generator=gen) * 80. - 100.
    pitch = torch.exp(
        torch.rand(batch, frames, generator=gen) *
        (math.log(550.) - math.log(50.)) + math.log(50.))
    periodicity = torch.rand(batch, frames, generator=gen)
    ppg = torch.softmax(
        3. * torch.randn(batch, PPG_CHANNELS, frames, generator=gen), dim=1)
    speakers = torch.arange(batch) % NUM_SPEAKERS
    ones = torch.ones(batch)
    return loudness, pitch, periodicity, ppg, speakers, ones, ones.clone()


###############################################################################
# STFT / mel / loudness preprocessing
###############################################################################


def spectrogram(audio, mels=False, threshold=None):
    """preprocess/spectrogram.py:15-60. audio (B,1,N)|(1,N) ->
    (B,513,N//256) (squeeze(0) when B == 1)."""
    window = torch.hann_window(WINDOW_SIZE, dtype=audio.dtype)
    size = (NUM_FFT - HOPSIZE) // 2
    audio = F.pad(audio, (size, size), mode='reflect')
    stft = torch.stft(
        audio.squeeze(1), NUM_FFT, hop_length=HOPSIZE, window=window,
        center=False, normalized=False, onesided=True, return_complex=True)
    stft = torch.view_as_real(stft)
    spec = torch.sqrt(stft.pow(2).sum(-1) + 1e-6)
    if mels:
        spec = linear_to_mel(spec, threshold)
    return spec.squeeze(0)


def spectrogram_dft(audio):
    """Same quantity as `spectrogram` via an explicit float64 framed DFT
    (independent check of the framing/reflect-pad/window conventions)."""
    a = audio.reshape(-1, audio.shape[-1]).double()
    size = (NUM_FFT - HOPSIZE) // 2
    a = F.pad(a[:, None], (size, size), mode='reflect')[:, 0]
    frames = a.unfold(-1, NUM_FFT, HOPSIZE)
    window = torch.hann_window(WINDOW_SIZE, dtype=torch.float64)
    n = torch.arange(NUM_FFT, dtype=torch.float64)
    k = torch.arange(NUM_FFT // 2 + 1, dtype=torch.float64)
    angle = 2 * math.pi * k[:, None] * n[None] / NUM_FFT
    re = torch.einsum('btn,kn->bkt', frames * window, torch.cos(angle))
    im = torch.einsum('btn,kn->bkt', frames * window, -torch.sin(angle))
    return torch.sqrt(re * re + im * im + 1e-6)


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide='ignore', invalid='ignore'):
        log_t = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_t, mels)


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(
        m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)),
        freqs)


def mel_basis(sr=SAMPLE_RATE, n_fft=NUM_FFT, n_mels=NUM_MELS):
    """`librosa.filters.mel(sr, n_fft, n_mels)` defaults (fmin 0, fmax sr/2,
    htk False, norm 'slaney', float32); called at
    preprocess/spectrogram.py:118-121. PARITY UNPINNED (librosa absent)."""
    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(
        hz_to_mel_slaney(0.0), hz_to_mel_slaney(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return torch.from_numpy(weights.astype(np.float32))


def linear_to_mel(spec, threshold=None):
    """preprocess/spectrogram.py:111-133: log(mel_basis @ spec)."""
    mel = torch.log(torch.matmul(mel_basis().to(spec.dtype), spec))
    if threshold is not None:
        return torch.clamp(mel, min=threshold)
    return mel


def a_weighting(frequencies, min_db=-80.0):
    """`librosa.A_weighting` (IEC 61672), PARITY UNPINNED."""
    f_sq = np.asanyarray(frequencies, dtype=np.float64) ** 2.0
    const = np.array([12194.217, 20.598997, 107.65265, 737.86223]) ** 2.0
    with np.errstate(divide='ignore'):
        weights = 2.0 + 20.0 * (
            np.log10(const[0]) + 2 * np.log10(f_sq) -
            np.log10(f_sq + const[0]) - np.log10(f_sq + const[1]) -
            0.5 * np.log10(f_sq + const[2]) - 0.5 * np.log10(f_sq + const[3]))
    return np.maximum(min_db, weights)


def perceptual_weights():
    """preprocess/loudness.py:149-160: A_weighting(fft_frequencies) - REF_DB,
    shape (513, 1), float64 as numpy returns it."""
    freqs = np.linspace(0, SAMPLE_RATE / 2, 1 + WINDOW_SIZE // 2)
    return a_weighting(freqs)[:, None] - float(REF_DB)


def loudness(audio, bands=1):
    """preprocess/loudness.py:17-55. audio (1, N) -> (bands|513, N // 256).

    librosa.stft(center=False, hann periodic) -> complex64;
    amplitude_to_db(|S|): 20 log10(max(1e-5, |S|)) floored at max - 80 over
    the whole utterance; + A-weights - 20; floor at MIN_DB.
    """
    padding = (WINDOW_SIZE - HOPSIZE) // 2
    x = F.pad(audio[None], (padding, padding), mode='reflect').squeeze(0)
    x = x.detach().cpu().numpy().squeeze(0)
    window = torch.hann_window(WINDOW_SIZE, dtype=torch.float64).numpy()
    frames = np.lib.stride_tricks.sliding_window_view(
        x, WINDOW_SIZE)[::HOPSIZE]
    dtype = np.complex64 if x.dtype == np.float32 else np.complex128
    stft = np.fft.rfft(
        frames * window.astype(x.dtype), axis=-1).T.astype(dtype)
    magnitude = np.abs(stft)
    # librosa.amplitude_to_db(S) = power_to_db(|S|**2, ref=1, amin=1e-10,
    # top_db=80): 10 log10(max(amin, |S|^2)) - 10 log10(max(amin, 1))
    amin, top_db = 1e-5, 80.0
    power = np.square(magnitude)
    db = 10.0 * np.log10(np.maximum(amin ** 2, power))
    db -= 10.0 * np.log10(np.maximum(amin ** 2, 1.0))
    db = np.maximum(db, db.max() - top_db)
    weighted = db + perceptual_weights()
    weighted[weighted < MIN_DB] = MIN_DB
    out = torch.from_numpy(weighted).float()
    return band_average(out, bands) if bands is not None else out


###############################################################################
# FARGAN (config/fargan.py): frame-autoregressive GRU vocoder
###############################################################################

FARGAN_SUBFRAMES = 4                            # defaults.py:244
FARGAN_SUBFRAME_SIZE = HOPSIZE // 4             # defaults.py:247
FARGAN_PREVIOUS_SAMPLES = HOPSIZE * 2           # static.py:71-72


def fold_weight_norm_linear(g, v):
    """weight_norm on a Linear (dim=0): per output row, g (O, 1), v (O, I)."""
    return v * (g / torch.linalg.vector_norm(v, dim=1, keepdim=True))


def fargan_weights(state):
    """Folded FARGAN weights from a reference-keyed Generator state dict."""
    p = 'model.subframe_network.'
    w = {}
    for i, key in enumerate((0, 2, 4)):
        w[f'cond{i}'] = state[f'model.conditioning_network.{key}.weight']

    def normed(prefix):
        if prefix + '.weight' in state:
            return state[prefix + '.weight']
        return fold_weight_norm_linear(
            state[prefix + '.weight_g'], state[prefix + '.weight_v'])

    w['fwconv'] = normed(p + 'framewise_convolution.model.0')
    w['fwconv_glu'] = normed(p + 'framewise_convolution.model.2.gate')
    for n in (1, 2, 3):
        w[f'gru{n}_ih'] = state[p + f'gru{n}.weight_ih']
        w[f'gru{n}_hh'] = state[p + f'gru{n}.weight_hh']
        w[f'gru{n}_glu'] = normed(p + f'gru{n}_glu.gate')
    w['skip_glu'] = normed(p + 'skip_glu.gate')
    w['skip'] = state[p + 'skip_dense.weight']
    w['out'] = state[p + 'output_layer.weight']
    return w


def _glu(x, gate):
    """model/fargan.py:375-388"""
    return x * torch.sigmoid(F.linear(x, gate))


def _gru_cell(x, h, w_ih, w_hh):
    """torch.nn.GRUCell(bias=False) (model/fargan.py:212-223)."""
    gi, gh = F.linear(x, w_ih), F.linear(h, w_hh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def fargan_subframe(w, features, previous_samples, period, states):
    """SubframeNetwork.forward in eval mode, FARGAN_GAIN_NORMALIZATION off
    (model/fargan.py:199-335). previous_samples (B, 512)."""
    size = FARGAN_SUBFRAME_SIZE
    total = previous_samples.shape[-1]
    index = total - period[:, None] + torch.arange(size + 4)[None] - 2
    index = index - period[:, None] * (index >= total)          # :233-237
    lookback = torch.gather(previous_samples, 1, index)         # :238-241
    previous_subframe = previous_samples[:, -size:]             # :244-246
    subframe_input = torch.cat(
        (features, previous_subframe, lookback), dim=1)         # :254-256
    fwconv = _glu(
        torch.tanh(F.linear(
            torch.cat((subframe_input, states[3]), -1), w['fwconv'])),
        w['fwconv_glu'])                                        # :257-259, :366
    lookback = lookback[:, 2:-2]                                # :260
    outs, new_states = [], []
    x = fwconv
    for n in (1, 2, 3):                                         # :269-314
        h = _gru_cell(
            torch.cat((x, lookback, previous_subframe), dim=1),
            states[n - 1], w[f'gru{n}_ih'], w[f'gru{n}_hh'])
        x = _glu(h, w[f'gru{n}_glu'])
        outs.append(x)
        new_states.append(h)
    skip = torch.cat(
        outs + [fwconv, lookback, previous_subframe], dim=1)    # :317-326
    skip = _glu(torch.tanh(F.linear(skip, w['skip'])), w['skip_glu'])
    output = torch.tanh(F.linear(skip, w['out']))               # :333
    return output, (*new_states, subframe_input)


def fargan_forward(features, global_features, previous_samples, state,
                   return_states=False):
    """FARGAN.forward (model/fargan.py:21-59) + step (:65-131).

    features (B, 114, T) with the pitch period (in samples) as the last
    channel, global_features (B, 258, 1), previous_samples (B, 1, 512) ->
    (B, 1, 256 T)."""
    w = fargan_weights(state)
    batch = features.shape[0]
    prev = previous_samples[:, 0].expand(batch, -1).clone()
    glob = global_features.squeeze(2).expand(batch, -1)
    states = (
        torch.zeros(batch, HOPSIZE), torch.zeros(batch, HOPSIZE),
        torch.zeros(batch, HOPSIZE),
        torch.zeros(batch, 4 * FARGAN_SUBFRAME_SIZE + 4))       # :406-415
    frames = []
    for frame in features.permute(2, 0, 1):
        period = torch.round(frame[:, -1]).to(torch.long)       # :94
        x = torch.cat((frame[:, :-1], glob), dim=1)
        for i in range(3):                                      # :139-160
            x = torch.tanh(F.linear(x, w[f'cond{i}']))
        # reshape(B, 128, 4).permute(2, 0, 1): sub-frame s takes x[:, s::4]
        subframes = x.reshape(batch, 2 * FARGAN_SUBFRAME_SIZE,
                              FARGAN_SUBFRAMES).permute(2, 0, 1)
        for sub in subframes:
            out, states = fargan_subframe(w, sub, prev, period, states)
            frames.append(out)
            prev = torch.cat((prev[:, FARGAN_SUBFRAME_SIZE:], out), dim=1)
    signal = torch.cat(frames, dim=1).unsqueeze(1)
    if return_states:
        return signal, states, prev
    return signal


def fargan_generator_forward(
    loudness, pitch, periodicity, ppg, speakers, spectral_balance_ratios,
    loudness_ratios, state, previous_samples=None
):
    """Generator.forward with MODEL == 'fargan' (model/generator.py:116-135,
    period channel :191-195)."""
    features = prepare_features(
        loudness, pitch, periodicity, ppg, state['pitch_distribution'],
        state['pitch_embedding.weight'], state['ppg_threshold'])
    period = SAMPLE_RATE / torch.clip(pitch, FMIN, FMAX)
    features = torch.cat((features, period[:, None]), dim=1)
    global_features = prepare_global_features(
        speakers, spectral_balance_ratios, loudness_ratios,
        state['speaker_embedding.weight'])
    if previous_samples is None:
        previous_samples = torch.zeros(
            features.shape[0], 1, FARGAN_PREVIOUS_SAMPLES)
    return fargan_forward(features, global_features, previous_samples, state)


def random_state_fargan(seed=0, pitch_distribution=None):
    """Reference-keyed FARGAN Generator state dict with the reference's init
    scheme (orthogonal Linear weights, model/fargan.py:418-424; GRUCell
    default U(-1/sqrt(H), 1/sqrt(H)); weight-normed layers g = ||v||)."""
    gen = torch.Generator().manual_seed(seed)

    def orthogonal(rows, cols):
        return torch.nn.init.orthogonal_(torch.empty(rows, cols), generator=gen)

    state = {
        'default_previous_samples': torch.zeros(1, 1, FARGAN_PREVIOUS_SAMPLES),
        'ppg_threshold': torch.tensor(SPARSE_PPG_THRESHOLD)}
    if pitch_distribution is None:
        pitch_distribution = torch.exp(torch.linspace(
            math.log(55.9431), math.log(547.3264), PITCH_BINS))
    state['pitch_distribution'] = pitch_distribution.clone().float()
    channels = NUM_FEATURES + GLOBAL_CHANNELS
    state['model.conditioning_network.0.weight'] = orthogonal(channels, channels)
    state['model.conditioning_network.2.weight'] = orthogonal(channels, channels)
    state['model.conditioning_network.4.weight'] = orthogonal(
        2 * HOPSIZE, channels)
    p = 'model.subframe_network.'

    def normed(prefix, rows, cols):
        v = orthogonal(rows, cols)
        state[prefix + '.weight_g'] = torch.linalg.vector_norm(
            v, dim=1, keepdim=True)
        state[prefix + '.weight_v'] = v

    normed(p + 'framewise_convolution.model.0', HOPSIZE,
           2 * (4 * FARGAN_SUBFRAME_SIZE + 4))
    normed(p + 'framewise_convolution.model.2.gate', HOPSIZE, HOPSIZE)
    bound = 1. / math.sqrt(HOPSIZE)
    for n in (1, 2, 3):
        state[p + f'gru{n}.weight_ih'] = (torch.rand(
            3 * HOPSIZE, HOPSIZE + 2 * FARGAN_SUBFRAME_SIZE,
            generator=gen) * 2 - 1) * bound
        state[p + f'gru{n}.weight_hh'] = (torch.rand(
            3 * HOPSIZE, HOPSIZE, generator=gen) * 2 - 1) * bound
    for name in ('gru1_glu', 'gru2_glu', 'gru3_glu', 'skip_glu'):
        normed(p + name + '.gate', HOPSIZE, HOPSIZE)
    state[p + 'skip_dense.weight'] = orthogonal(
        HOPSIZE, 4 * HOPSIZE + 2 * FARGAN_SUBFRAME_SIZE)
    state[p + 'output_layer.weight'] = orthogonal(
        FARGAN_SUBFRAME_SIZE, HOPSIZE)
    state['speaker_embedding.weight'] = torch.randn(
        NUM_SPEAKERS, SPEAKER_CHANNELS, generator=gen)
    state['pitch_embedding.weight'] = torch.randn(
        PITCH_BINS, PITCH_EMBEDDING_SIZE, generator=gen)
    return state



###############################################################################
# promonet.edit
###############################################################################


def grid_sample(sequence, grid, method='linear'):
    """edit/grid.py:12-45 (PINNED against the reference by make_golden)."""
    if method == 'linear':
        xp = torch.arange(sequence.shape[-1])
        i = torch.searchsorted(xp, grid, side='right')
        fp = F.pad(sequence, (0, 1), mode='replicate')
        xp = torch.cat((xp, xp[-1:] + 1))
        return fp[..., i - 1] * (xp[i] - grid) + fp[..., i] * (grid - xp[i - 1])
    if method == 'nearest':
        return sequence[..., torch.round(grid).to(torch.long)]
    raise ValueError(f'Grid sampling method {method} is not defined')


def grid_of_length(tensor, length):
    """`ppgs.edit.grid.of_length` (third-party, absent: PARITY UNPINNED)."""
    return torch.linspace(0., tensor.shape[-1] - 1., int(length))


def grid_constant(tensor, ratio):
    """`ppgs.edit.grid.constant` (third-party, absent: PARITY UNPINNED)."""
    return grid_of_length(tensor, round(tensor.shape[-1] / ratio + 1e-4))


# Phoneme inventory behind the selective time-stretch (third-party ppgs /
# pypar constants read at edit/core.py:57-76; PARITY UNPINNED - restated)
PHONEMES = [
    'aa', 'ae', 'ah', 'ao', 'aw', 'ay', 'b', 'ch', 'd', 'dh', 'eh', 'er', 'ey',
    'f', 'g', 'hh', 'ih', 'iy', 'jh', 'k', 'l', 'm', 'n', 'ng', 'ow', 'oy', 'p',
    'r', 's', 'sh', 't', 'th', 'uh', 'uw', 'v', 'w', 'y', 'z', 'zh', '<silent>']
VOICED = [
    'aa', 'ae', 'ah', 'ao', 'aw', 'ay', 'b', 'd', 'dh', 'eh', 'er', 'ey', 'g',
    'ih', 'iy', 'jh', 'l', 'm', 'n', 'ng', 'ow', 'oy', 'r', 'uh', 'uw', 'v',
    'w', 'y', 'z', 'zh']
SILENCE = '<silent>'


def stretched_phonemes(stretch_unvoiced, stretch_silence):
    """edit/core.py:57-76: indices of the phonemes that are stretched."""
    index = {p: i for i, p in enumerate(PHONEMES)}
    indices = [index[p] for p in VOICED]
    if stretch_silence:
        indices.append(index[SILENCE])
    if stretch_unvoiced:
        indices.extend(list(
            index[p] for p in PHONEMES if p not in VOICED and p != SILENCE))
    return indices


def grid_selective(ppg, ratio, indices):
    """edit/core.py:77-110: the sequential recurrence, in float32 like the
    reference (whose loop variables are 0-dim float32 tensors from the first
    step on)."""
    selected = ppg[torch.tensor(indices)].sum(dim=0)
    target = round(ppg.shape[-1] / ratio)
    total = selected.sum()
    effective = (target - (ppg.shape[-1] - total)) / total
    grid = torch.zeros(target)
    i = torch.zeros(())
    for j in range(1, target):
        left = min(int(math.floor(i)), len(selected) - 1)
        if left + 1 < len(selected):
            offset = i - left
            probability = offset * selected[left + 1] + \
                (1 - offset) * selected[left]
        else:
            probability = selected[left]
        step = 1. / (probability * effective + (1 - probability))
        grid[j] = grid[j - 1] + step
        i = i + step
    return grid


def edit_from_features(
    loudness, pitch, periodicity, ppg, pitch_shift_cents=None,
    time_stretch_ratio=None, loudness_scale_db=None, grid=None,
    stretch_unvoiced=True, stretch_silence=True
):
    """edit/core.py:17-132."""
    if time_stretch_ratio is not None:
        if grid is None:
            if stretch_unvoiced and stretch_silence:
                grid = grid_constant(ppg, time_stretch_ratio)
            else:
                grid = grid_selective(
                    ppg, time_stretch_ratio,
                    stretched_phonemes(stretch_unvoiced, stretch_silence))
        pitch = 2 ** grid_sample(torch.log2(pitch), grid)
        periodicity = grid_sample(periodicity, grid)
        loudness = grid_sample(loudness, grid)
        ppg = grid_sample(ppg, grid, 'linear')
    if pitch_shift_cents is not None:
        pitch = torch.clip(
            pitch.clone() * 2 ** (pitch_shift_cents / 1200), FMIN, FMAX)
    if loudness_scale_db is not None:
        loudness = loudness + loudness_scale_db
    return loudness, pitch, periodicity, ppg
