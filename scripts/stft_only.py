"""The three preprocessing launches alone (spectrogram, log-mel, 8-band
loudness of a batch of 32 x 10 s waveforms), a few calls each: the command
rocprofv3 wraps for the FFT kernels' stats / PMC passes
(scripts/profile_stft.sh, profiles/r04/stft_pmc.txt)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import promonet_amd  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 5
device = torch.device('cuda:0')
samples = promonet_amd.convert.seconds_to_frames(10) * promonet_amd.HOPSIZE
torch.manual_seed(0)
audio = ((torch.rand(32, 1, samples) * 2 - 1) * .5).to(device)
with torch.inference_mode():
    for _ in range(calls):
        promonet_amd.preprocess.spectrogram.from_audio(audio)
        promonet_amd.preprocess.spectrogram.from_audio(audio, True)
        promonet_amd.preprocess.loudness.from_audio(audio[:, 0], 8)
    torch.cuda.synchronize()
