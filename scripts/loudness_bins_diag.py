"""GPU box: where the per-bin (bands=None) loudness of the 32 x 10 s batch
differs most from the CPU oracle, and how both sit against a float64 run."""
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
worst = []
for item in range(audio.shape[0]):
    w32 = oracle.loudness(audio[item:item + 1], None)
    w64 = oracle.loudness(audio[item:item + 1].double(), None)
    spec = oracle.spectrogram(audio[item:item + 1, None].double())
    power = (spec ** 2 - 1e-6).clamp_min(1e-20)
    d = (got[item] - w32).abs()
    index = int(d.argmax())
    k, f = divmod(index, d.shape[1])
    worst.append((d.max().item(), item, k, f, w32[k, f].item(),
                  (got[item] - w64).abs()[k, f].item(),
                  (w32 - w64).abs()[k, f].item(),
                  10 * torch.log10(power[k, f] / power[:, f].max()).item(),
                  (got[item] - w64).abs().max().item(),
                  (w32 - w64).abs().max().item()))
for row in sorted(worst, reverse=True)[:8]:
    print('err %.2e utt %d bin %d frame %d want %.2f | ours-vs-f64 %.2e '
          'oracle32-vs-f64 %.2e | bin power %.1f dB under frame max | whole '
          'utterance: ours-vs-f64 %.2e oracle32-vs-f64 %.2e' % row)
