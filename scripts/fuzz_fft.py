"""GPU box: random shapes through the three FFT transforms against the CPU
oracle (spectrogram, log-mel, loudness 8 / 1 / None bands), both workgroup
shapes: utterance lengths from the minimum (385 samples) up, batches 1..5,
lengths that are not hop multiples, one-frame and two-frame utterances.
usage: python scripts/fuzz_fft.py [cases] [seed]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402
from promonet_amd import _lib  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator().manual_seed(seed)
device = torch.device('cuda:0')
lib = _lib.lib()
worst = {'stft': 0., 'mel': 0., 'loud': 0.}
special = [385, 386, 511, 512, 513, 767, 768, 1023, 1024, 4096, 4097, 256 * 17 - 1,
           256 * 16, 256 * 33 + 255]
for case in range(cases):
    batch = int(torch.randint(1, 6, (1,), generator=gen))
    samples = special[case] if case < len(special) else \
        int(torch.randint(385, 60000, (1,), generator=gen))
    group = 16 if case % 2 == 0 else 32
    _lib.check(lib.pm_stft_set_frames_per_group(group))
    audio = torch.randn(batch, samples, generator=gen) * \
        10. ** (-torch.rand(batch, 1, generator=gen) * 3.)
    if case % 3 == 0:
        audio[:, samples // 3:samples // 2] *= 1e-5
    dev = audio.to(device)
    want = oracle.spectrogram(audio[:, None]).reshape(batch, 513, -1)
    got = promonet_amd.preprocess.spectrogram.from_audio(dev[:, None]).reshape(
        batch, 513, -1).cpu()
    assert got.shape == want.shape, (samples, got.shape, want.shape)
    ratio = ((got - want).abs() / (2e-5 + 1e-5 * want.abs())).max().item()
    worst['stft'] = max(worst['stft'], ratio)
    mel_want = oracle.linear_to_mel(want)
    mel = promonet_amd.preprocess.spectrogram.from_audio(
        dev[:, None], mels=True).reshape(batch, 80, -1).cpu()
    ratio_mel = ((mel - mel_want).abs() / (2e-5 + 1e-5 * mel_want.abs())).max().item()
    worst['mel'] = max(worst['mel'], ratio_mel)
    ratio_loud = 0.
    for bands in (8, 1, 5):
        loud = promonet_amd.preprocess.loudness.from_audio(dev, bands)
        loud = loud.reshape(batch, bands, -1).cpu()
        for item in range(batch):
            ref = oracle.loudness(audio[item:item + 1], bands)
            ratio_loud = max(ratio_loud, (
                (loud[item] - ref).abs() / (1e-4 + 1e-5 * ref.abs())).max().item())
    worst['loud'] = max(worst['loud'], ratio_loud)
    flag = '' if max(ratio, ratio_mel, ratio_loud) < 1. else '  <-- OUTSIDE THE GATE'
    print(f'case {case}: B {batch} N {samples} (T {samples // 256}) group {group}: '
          f'stft {ratio:.3f} mel {ratio_mel:.3f} loudness {ratio_loud:.3f} of the gate{flag}')
_lib.check(lib.pm_stft_set_frames_per_group(16))
print('worst / gate:', worst)
assert max(worst.values()) < 1.
print('fuzz_fft: all inside the gates')
