"""GPU box: per-bin (bands=None) loudness of the 32 x 10 s batch against a
FLOAT64 run of the oracle - ours and the float32 oracle (numpy's pocketfft)
side by side: the conditioning factor kappa of tests/test_gpu_preprocess_full.py
(excess over the plain 1e-4 dB + 1e-5 rel gate in units of 8.686 eps x frame
amplitude / bin amplitude, over the bins >= 50 dB under their frame) for both,
and where each is worst."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / 'oracle', ROOT / 'tests'):
    sys.path.insert(0, str(p))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402
import test_gpu_preprocess_full as t  # noqa: E402

audio = t.audio_batch(*t.FULL, 41)
got = promonet_amd.preprocess.loudness.from_audio(audio.cuda(), None).cpu()
eps32 = 2. ** -23
kappa = {'ours': 0., 'oracle32': 0., 'ours-vs-oracle32': 0.}
outside = {k: 0 for k in kappa}
worst = []
for item in range(audio.shape[0]):
    w32 = oracle.loudness(audio[item:item + 1], None)
    w64 = oracle.loudness(audio[item:item + 1].double(), None)
    spec = oracle.spectrogram(audio[item:item + 1, None].double()).squeeze()
    assert spec.shape == w64.shape, (spec.shape, w64.shape)
    power = (spec ** 2 - 1e-6).clamp_min(1e-20)
    conditioning = (power.sum(0, keepdim=True) / power).sqrt()
    gate = 1e-4 + 1e-5 * w64.abs()
    for name, a, b in (('ours', got[item].double(), w64),
                       ('oracle32', w32.double(), w64),
                       ('ours-vs-oracle32', got[item].double(), w32.double())):
        d = (a - b).abs()
        weak = conditioning > 300.      # (bins >= 50 dB under their frame)
        kappa[name] = max(kappa[name], ((d - gate).clamp_min(0.) / (
            8.686 * eps32 * conditioning))[weak].max().item())
        outside[name] += (d > gate).sum().item()
    d = (got[item].double() - w64).abs()
    index = int(d.argmax())
    k, f = divmod(index, d.shape[1])
    worst.append((d.max().item(), item, k, f, (w32.double() - w64).abs()[k, f].item(),
                  (w32.double() - w64).abs().max().item()))
print('conditioning factor kappa against float64:', kappa, 'bins outside the plain gate:', outside)
for row in sorted(worst, reverse=True)[:5]:
    print('ours-vs-f64 %.2e dB at utt %d bin %d frame %d (oracle32 there %.2e; '
          'oracle32 worst of the utterance %.2e)' % row)
