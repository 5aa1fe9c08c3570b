"""Audio -> (loudness, pitch, periodicity, ppg) orchestration.

API of `promonet.preprocess` (promonet/preprocess/core.py). Loudness runs on
the HIP kernels. Pitch / periodicity (`penn`) and PPGs (`ppgs`) are
third-party neural networks outside the hot-path scope: they are imported
lazily and used as the reference uses them when installed.
"""
from pathlib import Path

import torch

import promonet_amd


def from_audio(
    audio,
    sample_rate=None,
    gpu=None,
    features=None,
    loudness_bands=None,
    max_harmonics=3
):
    """Preprocess audio (preprocess/core.py:17-126)."""
    sample_rate = sample_rate or promonet_amd.SAMPLE_RATE
    features = list(features or promonet_amd.INPUT_FEATURES)
    if loudness_bands is None:
        loudness_bands = promonet_amd.LOUDNESS_BANDS
    if gpu is None:
        raise RuntimeError(
            'promonet_amd preprocessing runs on an AMD GPU: pass gpu=<index>')
    device = torch.device(f'cuda:{gpu}')
    if sample_rate != promonet_amd.SAMPLE_RATE:
        raise ValueError(
            f'resample to {promonet_amd.SAMPLE_RATE} Hz first '
            '(promonet_amd.load.audio does)')
    result = []
    if 'loudness' in features:
        result.append(promonet_amd.preprocess.loudness.from_audio(
            audio.to(device), loudness_bands))
    if 'pitch' in features or 'periodicity' in features:
        try:
            import penn
        except ImportError as error:
            raise ImportError(
                'pitch / periodicity extraction needs the third-party `penn` '
                'package (out of scope of promonet_amd)') from error
        pitch, periodicity = penn.from_audio(
            audio, sample_rate=sample_rate,
            hopsize=promonet_amd.convert.samples_to_seconds(
                promonet_amd.HOPSIZE),
            fmin=promonet_amd.FMIN, fmax=promonet_amd.FMAX, batch_size=2048,
            center='half-hop', decoder='viterbi', gpu=gpu)
        if 'pitch' in features:
            result.append(pitch)
        if 'periodicity' in features:
            result.append(periodicity)
    if 'ppg' in features:
        try:
            import ppgs
        except ImportError as error:
            raise ImportError(
                'PPG extraction needs the third-party `ppgs` package (out of '
                'scope of promonet_amd)') from error
        ppg = ppgs.from_audio(audio, sample_rate, gpu=gpu)
        frames = promonet_amd.convert.samples_to_frames(audio.shape[-1])
        if ppg.shape[-1] != frames:
            grid = ppgs.edit.grid.of_length(ppg, frames)
            ppg = ppgs.edit.grid.sample(ppg, grid, 'linear')
        ppg = torch.softmax(torch.log(ppg + 1e-8), -2)
        result.append(ppg)
    unsupported = set(features) - {'loudness', 'pitch', 'periodicity', 'ppg'}
    if unsupported:
        raise ValueError(
            f'features {sorted(unsupported)} are evaluation-only in the '
            'reference and out of scope here')
    return tuple(result) if len(result) != 1 else result[0]


def from_file(file, gpu=None, features=None, loudness_bands=None):
    """preprocess/core.py:129-166"""
    return from_audio(
        promonet_amd.load.audio(file), gpu=gpu, features=features,
        loudness_bands=loudness_bands)


def from_file_to_file(
    file,
    output_prefix=None,
    gpu=None,
    features=None,
    loudness_bands=None
):
    """Preprocess and save `{prefix}-loudness.pt`, `{prefix}[-viterbi]-pitch.pt`,
    `...-periodicity.pt`, `{prefix}-ppg.pt` (preprocess/core.py:169-224)."""
    file = Path(file)
    features = list(features or promonet_amd.INPUT_FEATURES)
    if output_prefix is None:
        output_prefix = file.parent / file.stem
    outputs = from_file(file, gpu, features, loudness_bands)
    if not isinstance(outputs, tuple):
        outputs = (outputs,)
    viterbi = '-viterbi' if promonet_amd.VITERBI_DECODE_PITCH else ''
    names = {
        'loudness': '-loudness.pt',
        'pitch': f'{viterbi}-pitch.pt',
        'periodicity': f'{viterbi}-periodicity.pt',
        'ppg': '-ppg.pt'}
    ordered = [f for f in ('loudness', 'pitch', 'periodicity', 'ppg')
               if f in features]
    for feature, output in zip(ordered, outputs):
        torch.save(output.cpu(), f'{output_prefix}{names[feature]}')


def from_files_to_files(
    files,
    output_prefixes=None,
    gpu=None,
    features=None,
    loudness_bands=None
):
    """preprocess/core.py:227-319"""
    if output_prefixes is None:
        output_prefixes = [None] * len(files)
    for file, prefix in zip(files, output_prefixes):
        from_file_to_file(file, prefix, gpu, features, loudness_bands)
