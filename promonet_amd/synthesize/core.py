"""Speech synthesis from features, on the MI355X HIP engine.

API of `promonet.synthesize` (promonet/synthesize/core.py): identical names,
positional order, defaults and return conventions, plus a batched entry
point the reference lacks (`from_features_batched`).
"""
import contextlib
import os
import time
from pathlib import Path
from typing import List, Optional, Union

import numpy as np
import torch

import promonet_amd


###############################################################################
# Timers (the two torchutil.time contexts of synthesize/core.py:222,250)
###############################################################################


class timer:
    """Wall-clock accumulators keyed like the reference's torchutil timers
    ('load', 'generate'); read by `timer.results()` (evaluate/core.py:125)."""
    seconds = {}

    @classmethod
    @contextlib.contextmanager
    def context(cls, name):
        start = time.perf_counter()
        try:
            yield
        finally:
            cls.seconds[name] = (
                cls.seconds.get(name, 0.) + time.perf_counter() - start)

    @classmethod
    def reset(cls):
        cls.seconds.clear()

    @classmethod
    def results(cls):
        return dict(cls.seconds)


###############################################################################
# Editing API
###############################################################################


def from_features(
    loudness: torch.Tensor,
    pitch: torch.Tensor,
    periodicity: torch.Tensor,
    ppg: torch.Tensor,
    speaker: Union[int, torch.Tensor] = 0,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None
) -> torch.Tensor:
    """Synthesize one utterance from its features (synthesize/core.py:18-59).

    loudness (8|513, T) or (B, ., T) dB, pitch (B, T) Hz, periodicity (B, T),
    ppg (B, 40, T); `speaker` index, the two augmentation ratios (> 1 raises
    the spectral balance / the loudness), optional generator checkpoint file
    or directory, and the GPU index (required: there is no CPU path).
    Returns (1, 256 T) float32 - utterance 0 of the batch only, as the
    reference does (core.py:281).
    """
    device = _device(gpu, pitch)
    if loudness.ndim == 2:
        loudness = loudness[None]
    return generate(
        loudness.to(device),
        pitch.to(device),
        periodicity.to(device),
        ppg.to(device),
        speaker,
        spectral_balance_ratio,
        loudness_ratio,
        checkpoint
    ).to(torch.float32)


def from_features_batched(
    loudness: torch.Tensor,
    pitch: torch.Tensor,
    periodicity: torch.Tensor,
    ppg: torch.Tensor,
    speakers: Union[int, torch.Tensor, List[int]] = 0,
    spectral_balance_ratios: Union[float, torch.Tensor] = 1.,
    loudness_ratios: Union[float, torch.Tensor] = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None,
    lengths: Optional[Union[torch.Tensor, List[int]]] = None
) -> torch.Tensor:
    """Batched synthesis: (B, 8|513, T), (B, T), (B, T), (B, 40, T) ->
    (B, 1, 256 T). Per-utterance speakers and ratios; not in the reference,
    whose public API returns utterance 0 only. `lengths` (B,) frames: the
    utterances are zero-padded to T and each is synthesised exactly as if
    alone (audio past 256 * lengths[b] is zero)."""
    device = _device(gpu, pitch)
    batch = pitch.shape[0]

    def per_item(value, dtype):
        if isinstance(value, torch.Tensor):
            return value.to(device=device, dtype=dtype).reshape(-1).expand(
                batch).contiguous()
        if isinstance(value, (list, tuple)):
            return torch.tensor(value, dtype=dtype, device=device)
        return torch.full((batch,), value, dtype=dtype, device=device)

    model = _cached_model(checkpoint, device)
    if model.zero_shot:
        speakers = speakers.to(device=device, dtype=torch.float)
        speakers = speakers.reshape(-1, speakers.shape[-1]).expand(
            batch, -1).contiguous()
    else:
        _check_speakers(speakers)
        speakers = per_item(speakers, torch.long)
    with timer.context('generate'), torch.inference_mode():
        return model(
            loudness.to(device), pitch.to(device), periodicity.to(device),
            ppg.to(device), speakers,
            per_item(spectral_balance_ratios, torch.float),
            per_item(loudness_ratios, torch.float),
            model.default_previous_samples, lengths)


def _load_features(loudness_file, pitch_file, periodicity_file, ppg_file):
    """torch.load the four feature files of one utterance; the PPG is
    resampled to the pitch's frame count and gets its batch axis."""
    pitch = torch.load(pitch_file)
    ppg = promonet_amd.load.ppg(ppg_file, resample_length=pitch.shape[-1])
    return (
        torch.load(loudness_file), pitch, torch.load(periodicity_file),
        ppg[None])


def from_file(
    loudness_file: Union[str, os.PathLike],
    pitch_file: Union[str, os.PathLike],
    periodicity_file: Union[str, os.PathLike],
    ppg_file: Union[str, os.PathLike],
    speaker: Union[int, torch.Tensor, Path, str] = 0,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None
) -> torch.Tensor:
    """`from_features` on features stored as .pt files (core.py:62-111);
    returns (1, samples) float32 on the GPU."""
    device = _device(gpu)
    features = [
        tensor.to(device) for tensor in _load_features(
            loudness_file, pitch_file, periodicity_file, ppg_file)]
    return from_features(
        *features, speaker, spectral_balance_ratio, loudness_ratio,
        checkpoint, gpu)


def from_file_to_file(
    loudness_file: Union[str, os.PathLike],
    pitch_file: Union[str, os.PathLike],
    periodicity_file: Union[str, os.PathLike],
    ppg_file: Union[str, os.PathLike],
    output_file: Union[str, os.PathLike],
    speaker: Union[int, torch.Tensor, Path, str] = 0,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None
) -> None:
    """`from_file`, then write a 22.05 kHz wav (core.py:114-155)."""
    audio = from_file(
        loudness_file, pitch_file, periodicity_file, ppg_file, speaker,
        spectral_balance_ratio, loudness_ratio, checkpoint, gpu)
    target = Path(output_file)
    target.parent.mkdir(exist_ok=True, parents=True)
    save_audio(target, audio.cpu())


def from_files_to_files(
    loudness_files: List[Union[str, os.PathLike]],
    pitch_files: List[Union[str, os.PathLike]],
    periodicity_files: List[Union[str, os.PathLike]],
    ppg_files: List[Union[str, os.PathLike]],
    output_files: List[Union[str, os.PathLike]],
    speakers: Optional[Union[List[int], torch.Tensor, Path, str]] = None,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None
) -> None:
    """One `from_file_to_file` per utterance, in order - the reference's
    sequential behaviour (core.py:158-201). `from_files_to_files_batched`
    below is the throughput path."""
    count = len(pitch_files)
    speakers = [0] * count if speakers is None else speakers
    for index in range(count):
        from_file_to_file(
            loudness_files[index], pitch_files[index],
            periodicity_files[index], ppg_files[index], output_files[index],
            speakers[index], spectral_balance_ratio, loudness_ratio,
            checkpoint, gpu)


def from_files_to_files_batched(
    loudness_files: List[Union[str, os.PathLike]],
    pitch_files: List[Union[str, os.PathLike]],
    periodicity_files: List[Union[str, os.PathLike]],
    ppg_files: List[Union[str, os.PathLike]],
    output_files: List[Union[str, os.PathLike]],
    speakers: Optional[List[int]] = None,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint: Optional[Union[str, os.PathLike]] = None,
    gpu: Optional[int] = None,
    batch_size: int = 32
) -> None:
    """`from_files_to_files` with the files synthesised `batch_size` at a
    time (sorted by length, zero-padded, ragged-exact) instead of the
    reference's one-utterance loop (synthesize/core.py:158-201). Same files,
    same audio as the sequential path. SURVEY.md 8(f) item 2."""
    device = _device(gpu)
    if speakers is None:
        speakers = [0] * len(pitch_files)
    items = []
    for index in range(len(pitch_files)):
        pitch = torch.load(pitch_files[index])
        items.append((
            pitch.shape[-1], index, torch.load(loudness_files[index]), pitch,
            torch.load(periodicity_files[index]),
            promonet_amd.load.ppg(ppg_files[index], pitch.shape[-1])))
    items.sort(key=lambda item: item[0])
    for start in range(0, len(items), batch_size):
        group = items[start:start + batch_size]
        frames = max(item[0] for item in group)

        def padded(tensors):
            out = torch.zeros(
                (len(tensors),) + tuple(tensors[0].shape[:-1]) + (frames,))
            for row, tensor in zip(out, tensors):
                row[..., :tensor.shape[-1]] = tensor
            return out.to(device)

        audio = from_features_batched(
            padded([item[2].reshape(-1, item[0]) for item in group]),
            padded([item[3].reshape(item[0]) for item in group]),
            padded([item[4].reshape(item[0]) for item in group]),
            padded([item[5].reshape(-1, item[0]) for item in group]),
            [speakers[item[1]] for item in group], spectral_balance_ratio,
            loudness_ratio, checkpoint, gpu,
            lengths=[item[0] for item in group]).cpu()
        for row, item in zip(audio, group):
            output_file = Path(output_files[item[1]])
            output_file.parent.mkdir(exist_ok=True, parents=True)
            save_audio(
                output_file, row[:, :item[0] * promonet_amd.HOPSIZE])


###############################################################################
# Pipeline
###############################################################################


def generate(
    loudness,
    pitch,
    periodicity,
    ppg,
    speaker=0,
    spectral_balance_ratio: float = 1.,
    loudness_ratio: float = 1.,
    checkpoint=None
) -> torch.Tensor:
    """Generate speech from phoneme and prosody features (core.py:209-281)"""
    device = pitch.device
    model = _cached_model(checkpoint, device)

    with timer.context('generate'):
        if model.zero_shot:                 # core.py:253-254: an x-vector
            speakers = speaker.to(device=device, dtype=torch.float).reshape(
                1, -1)
        else:
            _check_speakers(speaker)
            speakers = torch.full(
                (1,), int(speaker), dtype=torch.long, device=device)
        spectral_balance_ratio = torch.tensor(
            [spectral_balance_ratio], dtype=torch.float, device=device)
        loudness_ratio = torch.tensor(
            [loudness_ratio], dtype=torch.float, device=device)
        batch = pitch.shape[0]
        if batch > 1:
            # The reference broadcasts one speaker over the batch through the
            # (1, 512, 1) speaker-conv output (hifigan.py:68)
            speakers = speakers.expand(
                batch, *speakers.shape[1:]).contiguous()
            spectral_balance_ratio = \
                spectral_balance_ratio.expand(batch).contiguous()
            loudness_ratio = loudness_ratio.expand(batch).contiguous()
        with torch.inference_mode():
            return model(
                loudness,
                pitch,
                periodicity,
                ppg,
                speakers,
                spectral_balance_ratio,
                loudness_ratio,
                model.default_previous_samples
            )[0]


###############################################################################
# Utilities
###############################################################################


def _check_speakers(speakers):
    """Host-side ids (ints, lists, CPU tensors) are validated like the
    reference's torch.nn.Embedding does (IndexError); device tensors are not
    read back (no sync) - the kernel turns an out-of-range id into NaN audio -
    unless `promonet_amd.configure(CHECK_DEVICE_SPEAKERS=True)` asks for the
    reference's behaviour at the price of one device sync per call."""
    if isinstance(speakers, torch.Tensor):
        if speakers.is_cuda:
            if not promonet_amd.CHECK_DEVICE_SPEAKERS:
                return
            bad = (speakers < 0) | (speakers >= promonet_amd.NUM_SPEAKERS)
            if bool(bad.any()):
                raise IndexError(
                    f'speaker {int(speakers[bad][0])} is out of range '
                    f'[0, {promonet_amd.NUM_SPEAKERS})')
            return
        values = speakers.reshape(-1).tolist()
    elif isinstance(speakers, (list, tuple)):
        values = list(speakers)
    else:
        values = [speakers]
    for value in values:
        if not 0 <= int(value) < promonet_amd.NUM_SPEAKERS:
            raise IndexError(
                f'speaker {int(value)} is out of range '
                f'[0, {promonet_amd.NUM_SPEAKERS})')


def _device(gpu, like=None):
    if gpu is None:
        if like is not None and like.is_cuda:
            return like.device
        raise RuntimeError(
            'promonet_amd.synthesize runs on an AMD GPU only: pass gpu=<index> '
            '(the reference falls back to CPU PyTorch; this package does not)')
    return torch.device(f'cuda:{gpu}')


def _cached_model(checkpoint, device):
    """Model cache on function attributes (core.py:225-248). Unlike the
    reference, a `checkpoint=None` call does not reload on every call (the
    reference compares the downloaded path with None, core.py:227,247)."""
    with timer.context('load'):
        if (
            not hasattr(generate, 'model') or
            generate.checkpoint != checkpoint or
            generate.device != device
        ):
            model = promonet_amd.model.Generator()
            file = checkpoint
            if file is None:
                import huggingface_hub
                file = huggingface_hub.hf_hub_download(
                    'maxrmorrison/promonet',
                    f'generator-00{promonet_amd.STEPS}.pt')
            else:
                file = Path(file)
                if file.is_dir():
                    files = sorted(file.glob('generator-*.pt'))
                    if not files:
                        raise FileNotFoundError(
                            f'no generator-*.pt in {file}')
                    file = files[-1]
            load_checkpoint(file, model)
            generate.model = model.to(device).eval()
            generate.checkpoint = checkpoint
            generate.device = device
    return generate.model


def load_checkpoint(file, model):
    """`torchutil.checkpoint.load(file, model)`: a torch.save'd dict with the
    module state under 'model' (a bare state dict is accepted too)."""
    state = torch.load(file, map_location='cpu')
    if isinstance(state, dict) and 'model' in state and \
            isinstance(state['model'], dict):
        state = state['model']
    model.load_state_dict(state)
    return model


def set_model(model, device=None):
    """Install an already constructed Generator as the cached model (tests,
    benchmarks, random-init runs without a checkpoint file)."""
    if device is None:
        device = next(model.parameters()).device
    generate.model = model.to(device).eval()
    generate.checkpoint = None
    generate.device = torch.device(device)


def save_audio(file, audio):
    """32-bit float wav, as torchaudio.save writes a float tensor
    (core.py:155)."""
    import scipy.io.wavfile
    scipy.io.wavfile.write(
        str(file), promonet_amd.SAMPLE_RATE,
        audio.detach().cpu().to(torch.float32).numpy().T.astype(np.float32))
