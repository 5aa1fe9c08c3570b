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
    batch_size: int = 32,
    num_workers: Optional[int] = 0
) -> None:
    """`from_files_to_files` with the files synthesised `batch_size` at a
    time (sorted by length, zero-padded, ragged-exact) instead of the
    reference's one-utterance loop (synthesize/core.py:158-201). Same files,
    same audio as the sequential path. SURVEY.md 8(f) item 2.

    `num_workers` CPU processes (OPT-IN only - nothing in this package passes
    it, the CLI mirrors the reference's flags and its sequential loop; default
    0: everything in this process, so that a caller script needs no
    `if __name__ == '__main__':` guard; a natural value is
    `promonet_amd.NUM_WORKERS`, the size of the pools the reference forks for
    its file-level preprocessing, defaults.py:387; the `configure()` overrides
    travel to the workers as `Pool` initargs and therefore have to pickle)
    unpickle the feature files, pad the batches and write the wav
    files while the GPU synthesises: the serial loop spends three quarters of
    its time in `torch.load` and the wav writer. The pool is started on first
    use and kept (`shutdown_workers()` ends it; it is keyed on the
    `configure()` overrides, which the workers re-apply); small jobs (fewer
    than 4 batches) do not start one and run in this process. As with any
    `spawn` pool - the reference's included - a calling SCRIPT then needs its
    `if __name__ == '__main__':` guard."""
    device = _device(gpu)
    count = len(pitch_files)
    if count == 0:
        return
    if speakers is None:
        speakers = [0] * count
    if num_workers is None:
        num_workers = 0
    num_workers = min(int(num_workers), os.cpu_count() or 1)
    hop, rate = promonet_amd.HOPSIZE, promonet_amd.SAMPLE_RATE
    files = (loudness_files, pitch_files, periodicity_files, ppg_files)
    if num_workers < 1 or (count < 4 * batch_size and
                           getattr(_worker_pool, 'pool', None) is None):
        lengths = _load_lengths(pitch_files)
        # (longest first: the engine's workspace is sized once, not per batch)
        order = sorted(range(count), key=lambda index: (-lengths[index], index))
        for start in range(0, count, batch_size):
            group = order[start:start + batch_size]
            batch = _load_group([[f[i] for i in group] for f in files])
            audio = _synthesize_group(
                batch, [speakers[i] for i in group], spectral_balance_ratio,
                loudness_ratio, checkpoint, gpu).cpu()
            _write_group(
                audio, 0, [output_files[i] for i in group], batch[0], hop,
                rate)
        return

    import collections
    pool = _worker_pool(num_workers)
    chunk = -(-count // num_workers)
    with timer.context('files/lengths'):
        lengths = sum(pool.map(
            _load_lengths,
            [pitch_files[i:i + chunk] for i in range(0, count, chunk)]), [])
    # (longest first: the engine's workspace and every buffer of the caching
    # allocator are sized by the first batch instead of growing 32 times)
    order = sorted(range(count), key=lambda index: (-lengths[index], index))
    groups = [order[i:i + batch_size] for i in range(0, count, batch_size)]
    depth = 2 * num_workers           # batches in flight on either side
    loads = collections.deque()
    # Audio leaves through a ring of host buffers in SHARED memory, allocated
    # once: a tensor pickled to a worker travels as a handle, whereas a fresh
    # 28 MB tensor per batch costs its page faults again in every process
    # (measured: 36 ms per batch, more than its synthesis).
    ring = getattr(_worker_pool, 'ring', None)
    if ring is None or ring[0].shape[0] < batch_size or \
            ring[0].shape[-1] < max(lengths) * hop:
        # (kept with the pool: 4 x 28 MB of fresh shared pages are 0.15 s)
        _release_ring()
        with timer.context('files/ring'):
            ring = _worker_pool.ring = [
                torch.empty(batch_size, 1, max(lengths) * hop).share_memory_()
                for _ in range(4)]
            # page-lock the shared pages: the D2H copy of a batch (28 MB) is
            # then a DMA at link speed instead of a staged pageable copy
            try:
                runtime = torch.cuda.cudart()
                for buffer in ring:
                    runtime.cudaHostRegister(
                        buffer.data_ptr(), buffer.numel() * 4, 0)
            except Exception:
                pass
    busy = [[] for _ in ring]         # write tasks still reading a slot
    copy_stream = torch.cuda.Stream(device)
    submitted, previous = 0, None

    def flush(item, slot):
        """D2H of a finished batch on the side stream into ring slot `slot`,
        then off to the writer processes, a few files per task."""
        audio, done, names, frames = item
        with timer.context('files/wait for writers'):
            for task in busy[slot]:
                task.get(WORKER_TIMEOUT)
            busy[slot] = []
        rows, width = audio.shape[0], audio.shape[-1]
        # a CONTIGUOUS (rows, 1, width) view of the slot's first rows x width
        # floats: a strided destination would make the copy stage through a
        # fresh pageable buffer (the page faults the ring exists to avoid)
        target = ring[slot].view(-1)[:rows * width].view(rows, 1, width)
        with timer.context('files/audio to host'):
            copy_stream.wait_event(done)
            with torch.cuda.stream(copy_stream):
                target.copy_(audio, non_blocking=True)
            copy_stream.synchronize()
        busy[slot] = [
            pool.apply_async(_write_group, (
                ring[slot], first, names[first:first + 8],
                frames[first:first + 8], hop, rate, (rows, width)))
            for first in range(0, len(names), 8)]

    upload_stream = torch.cuda.Stream(device)

    def refill():
        nonlocal submitted
        while submitted < len(groups) and len(loads) < depth:
            loads.append(pool.apply_async(_load_group, (
                [[f[i] for i in groups[submitted]] for f in files],)))
            submitted += 1

    def upload(number):
        """Batch `number` from its loader to the device on a side stream: a
        copy out of pageable memory is stream-ordered AND blocks the host, so
        on the compute stream it would wait for the forward in front of it."""
        refill()
        with timer.context('files/wait for loaders'):
            batch = loads.popleft().get(WORKER_TIMEOUT)
        ids = [speakers[i] for i in groups[number]]
        _check_speakers(ids)
        with torch.cuda.stream(upload_stream):
            tensors = [t.to(device, non_blocking=True) for t in batch[1:]]
            tensors.append(torch.tensor(ids, dtype=torch.long).to(
                device, non_blocking=True))
            tensors.append(torch.tensor(batch[0], dtype=torch.int32).to(
                device, non_blocking=True))
            ready = torch.cuda.Event()
            ready.record(upload_stream)
        return batch[0], tensors, ready

    try:
        staged = upload(0)
        for number, group in enumerate(groups):
            frames, tensors, ready = staged
            compute = torch.cuda.current_stream(device)
            compute.wait_event(ready)
            for tensor in tensors:
                tensor.record_stream(compute)
            audio = _synthesize_group(
                (tensors[5], *tensors[:4]), tensors[4], spectral_balance_ratio,
                loudness_ratio, checkpoint, gpu)
            done = torch.cuda.Event()
            done.record(compute)
            # (the next batch arrives and the previous one leaves the device
            # while this one computes)
            if number + 1 < len(groups):
                staged = upload(number + 1)
            if previous is not None:
                flush(previous, (number - 1) % len(ring))
            previous = (audio, done, [output_files[i] for i in group], frames)
        if previous is not None:
            flush(previous, (len(groups) - 1) % len(ring))
        with timer.context('files/wait for writers'):
            for tasks in busy:
                for task in tasks:
                    task.get(WORKER_TIMEOUT)
    except BaseException:
        # writers / loaders of this job may still hold ring slots (or a worker
        # died: mp.Pool does not notice, hence the timeouts): the next call
        # must not inherit them - end the pool, drop the ring
        shutdown_workers()
        raise


# seconds one loader / writer task may take before the job fails instead of
# hanging on a worker the OS killed (mp.Pool never reports a dead worker)
WORKER_TIMEOUT = 600.


def _worker_pool(num_workers):
    """The IO processes, started once and kept (ten `spawn`ed interpreters
    importing torch take seconds - more than a thousand files' worth of
    synthesis): cached on the function like the model (`generate.model`),
    replaced when another size is asked for, ended by `shutdown_workers()` or
    at interpreter exit."""
    overrides = dict(promonet_amd.config.OVERRIDES)
    key = (num_workers, repr(sorted(overrides.items(), key=lambda kv: kv[0])))
    cached = getattr(_worker_pool, 'pool', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    shutdown_workers()
    import atexit
    import torch.multiprocessing as mp
    # ('spawn': the children never touch the GPU, but a fork of a process that
    # has initialised HIP inherits its runtime threads' locks)
    pool = mp.get_context('spawn').Pool(
        num_workers, initializer=_worker_init, initargs=(overrides,))
    _worker_pool.pool = (key, pool)
    if not getattr(_worker_pool, 'registered', False):
        atexit.register(shutdown_workers)
        _worker_pool.registered = True
    return pool


def _worker_init(overrides=None):
    # (unpickling and padding are single-threaded work: ten processes with
    # one intra-op thread pool per host core each would only fight)
    torch.set_num_threads(1)
    # a spawned interpreter imported the package with its DEFAULT constants:
    # the parent's configure() calls are replayed here
    if overrides:
        promonet_amd.configure(**overrides)


def shutdown_workers():
    """End the IO worker processes of `from_files_to_files_batched`."""
    cached = getattr(_worker_pool, 'pool', None)
    if cached is not None:
        cached[1].terminate()
        cached[1].join()
        _worker_pool.pool = None
    _release_ring()


def _release_ring():
    """Undo the page-locking of the shared audio ring before it is dropped."""
    for buffer in getattr(_worker_pool, 'ring', None) or []:
        try:
            torch.cuda.cudart().cudaHostUnregister(buffer.data_ptr())
        except Exception:
            pass
    _worker_pool.ring = None


def _load_lengths(pitch_files):
    """Frames of every utterance (worker side)."""
    return [int(torch.load(file).shape[-1]) for file in pitch_files]


def _load_group(files):
    """One zero-padded batch from the four feature files of its utterances
    (worker side: CPU only). Returns (frames per utterance, loudness
    (B, bands, T), pitch (B, T), periodicity (B, T), ppg (B, 40, T))."""
    loudness_files, pitch_files, periodicity_files, ppg_files = files
    items = []
    for index in range(len(pitch_files)):
        pitch = torch.load(pitch_files[index])
        frames = int(pitch.shape[-1])
        items.append((
            frames, torch.load(loudness_files[index]).reshape(-1, frames),
            pitch.reshape(frames),
            torch.load(periodicity_files[index]).reshape(frames),
            promonet_amd.load.ppg(ppg_files[index], frames).reshape(
                -1, frames)))
    longest = max(item[0] for item in items)

    def padded(column):
        first = items[0][column]
        out = torch.zeros(
            (len(items),) + tuple(first.shape[:-1]) + (longest,),
            dtype=torch.float32)
        for row, item in zip(out, items):
            row[..., :item[0]] = item[column]
        return out
    return ([item[0] for item in items], padded(1), padded(2), padded(3),
            padded(4))


def _synthesize_group(
    batch, speakers, spectral_balance_ratio, loudness_ratio, checkpoint, gpu
):
    frames, loudness, pitch, periodicity, ppg = batch
    return from_features_batched(
        loudness, pitch, periodicity, ppg, speakers, spectral_balance_ratio,
        loudness_ratio, checkpoint, gpu, lengths=frames)


def _write_group(
    audio, first, output_files, frames, hop=None, sample_rate=None, shape=None
):
    """Rows first ... of the host audio -> one wav per utterance, cut to its
    length (worker side). `audio` is (B, 1, samples), or a ring slot whose
    first rows x width floats hold a contiguous `shape` = (rows, width) batch.
    `hop` / `sample_rate` come from the parent: the worker's own module
    constants are only as good as the overrides replayed into it."""
    hop = promonet_amd.HOPSIZE if hop is None else hop
    if shape is not None:
        rows, width = shape
        audio = audio.view(-1)[:rows * width].view(rows, 1, width)
    for offset, (file, count) in enumerate(zip(output_files, frames)):
        file = Path(file)
        file.parent.mkdir(exist_ok=True, parents=True)
        save_audio(file, audio[first + offset, :, :count * hop], sample_rate)


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


def save_audio(file, audio, sample_rate=None):
    """32-bit float wav, as torchaudio.save writes a float tensor
    (core.py:155)."""
    import scipy.io.wavfile
    scipy.io.wavfile.write(
        str(file),
        promonet_amd.SAMPLE_RATE if sample_rate is None else sample_rate,
        audio.detach().cpu().to(torch.float32).numpy().T.astype(np.float32))
