"""Throughput of the preprocessing kernels (GPU box): magnitude spectrogram,
log-mel and A-weighted 8-band loudness of a batch of 32 x 10 s waveforms, with
the CPU port (oracle/restatement.py) timed beside them on a bounded sample."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402

device = torch.device('cuda:0')
batch, seconds = 32, 10
samples = promonet_amd.convert.seconds_to_frames(seconds) * promonet_amd.HOPSIZE
torch.manual_seed(0)
audio = (torch.rand(batch, 1, samples) * 2 - 1) * .5
audio_gpu = audio.to(device)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / reps


result = {}
cases = {
    'spectrogram': lambda: promonet_amd.preprocess.spectrogram.from_audio(audio_gpu),
    'log_mel': lambda: promonet_amd.preprocess.spectrogram.from_audio(audio_gpu, True),
    'loudness_8_bands': lambda: promonet_amd.preprocess.loudness.from_audio(audio_gpu[:, 0], 8),
}
frames = samples // promonet_amd.HOPSIZE
bins = promonet_amd.NUM_FFT // 2 + 1
dft_flops = 2. * promonet_amd.NUM_FFT * 2 * bins * batch * frames
with torch.inference_mode():
    for name, fn in cases.items():
        seconds_per = timed(fn)
        result[name] = {
            'ms': seconds_per * 1e3,
            'audio_seconds_per_second': batch * seconds / seconds_per,
            'dft_tflops': dft_flops / seconds_per / 1e12}
    # kernels alone: the C ABI on preallocated buffers (no allocation, no
    # Python wrapper), both FFT workgroup shapes
    from promonet_amd import _lib
    lib = _lib.lib()
    flat = audio_gpu[:, 0].contiguous()
    spec = torch.empty(batch, bins, frames, device=device)
    mel = torch.empty(batch, 80, frames, device=device)
    loud = torch.empty(batch, 8, frames, device=device)
    basis = promonet_amd.preprocess.spectrogram.mel_basis().to(device).contiguous()
    weights = promonet_amd.preprocess.loudness.perceptual_weights_tensor(device)
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device=device)
    stream = _lib.stream()
    prepared = promonet_amd.preprocess.spectrogram._prepared_mel_basis(device)
    abi = {
        'spectrogram': lambda: lib.pm_stft_magnitude(
            _lib.ptr(flat), _lib.ptr(spec), batch, samples, None, 0, stream),
        'log_mel': lambda: lib.pm_stft_mel(
            _lib.ptr(flat), prepared.data_ptr(), _lib.ptr(mel), batch, samples,
            80, 0, 0., stream),
        'loudness_8_bands': lambda: lib.pm_loudness(
            _lib.ptr(flat), _lib.ptr(weights), _lib.ptr(loud), batch, samples,
            8, -100., scratch.data_ptr(), scratch.numel(), stream),
    }
    in_bytes, frame_bytes = 4. * batch * samples, 4. * batch * frames
    algorithmic = {'spectrogram': in_bytes + bins * frame_bytes,
                   'log_mel': in_bytes + 80 * frame_bytes,
                   'loudness_8_bands': 2 * in_bytes + 8 * frame_bytes}
    for group in (16, 32):
        _lib.check(lib.pm_stft_set_frames_per_group(group))
        for name, fn in abi.items():
            seconds_per = timed(fn, reps=200)
            result[f'{name}_abi_group{group}'] = {
                'ms': seconds_per * 1e3,
                'algorithmic_gbs': algorithmic[name] / seconds_per / 1e9,
                'frac_of_8tbs': algorithmic[name] / seconds_per / 8e12}
    _lib.check(lib.pm_stft_set_frames_per_group(16))
    # the opt-in optimistic schedule of the 8-band loudness beside the default
    # (steady-level input: no group is transformed twice)
    _lib.check(lib.pm_stft_set_loudness_passes(1))
    seconds_per = timed(abi['loudness_8_bands'], reps=200)
    result['loudness_8_bands_optimistic_abi_group16'] = {
        'ms': seconds_per * 1e3}
    _lib.check(lib.pm_stft_set_loudness_passes(2))
    # CPU port on 2 utterances
    cpu_audio = audio[:2]
    cpu = {}
    cpu_cases = {
        'spectrogram': lambda: oracle.spectrogram(cpu_audio),
        'loudness_8_bands': lambda: [oracle.loudness(a, 8) for a in cpu_audio],
    }
    for name, fn in cpu_cases.items():
        fn()
        start = time.perf_counter()
        fn()
        cpu[name] = {'ms_for_2_utterances': (time.perf_counter() - start) * 1e3}
        cpu[name]['audio_seconds_per_second'] = \
            2 * seconds / (cpu[name]['ms_for_2_utterances'] * 1e-3)
    result['cpu_port'] = cpu
for key, value in result.items():
    print(key, value)
print(json.dumps(result))
