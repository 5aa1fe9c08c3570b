import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import torch
import promonet_amd
from promonet_amd import _lib
import test_gpu_preprocess_full as t
audio = t.audio_batch(*t.FULL, 41).cuda()
lib = _lib.lib()
_lib.check(lib.pm_stft_set_loudness_passes(2)); two = promonet_amd.preprocess.loudness.from_audio(audio, 8)
_lib.check(lib.pm_stft_set_loudness_passes(1)); one = promonet_amd.preprocess.loudness.from_audio(audio, 8)
d = (one - two).abs()
print('max diff', d.max().item(), 'count', (d > 0).sum().item(), 'of', d.numel())
idx = (d > 0).nonzero()
print(idx[:20].tolist())
for b, band, f in idx[:8].tolist():
    print(b, band, f, one[b, band, f].item(), two[b, band, f].item())
