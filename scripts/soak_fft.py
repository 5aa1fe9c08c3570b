import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(5)
audio = (torch.randn(32, 861 * 256, generator=gen) * .1).to(dev)
ref = None
for i in range(300):
    out = (promonet_amd.preprocess.spectrogram.from_audio(audio[:, None]),
           promonet_amd.preprocess.spectrogram.from_audio(audio[:, None], True),
           promonet_amd.preprocess.loudness.from_audio(audio, 8))
    if ref is None:
        ref = [t.clone() for t in out]
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, out)), i
torch.cuda.synchronize()
print('soak_fft: 300 x (spectrogram, log-mel, loudness) of batch 32 x 10 s bit-identical')
