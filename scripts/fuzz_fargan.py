"""Determinism check of the FARGAN cluster kernel on the GPU box: the same
batch synthesised repeatedly must come out bit-identical (a race between the
members' LDS reduction buffers or the granule exchange would show as run-to-run
differences), for one, two and four utterances per cluster and both weight
storages. usage: python scripts/fuzz_fargan.py [repeats]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import promonet_amd  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 6
device = torch.device('cuda:0')
bad = 0
for dtype in ('fp32', 'mixed', 'f16'):
    promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=dtype)
    torch.manual_seed(0)
    model = promonet_amd.model.Generator().to(device).eval()
    for batch, frames in ((32, 172), (48, 60), (96, 40), (7, 90)):
        inputs = synthetic_inputs(batch, frames, 77 + batch, device)
        with torch.inference_mode():
            first = model(*inputs, None).clone()
            finite = bool(torch.isfinite(first).all())
            same = True
            for _ in range(repeats - 1):
                same &= torch.equal(model(*inputs, None), first)
        status = 'ok' if same and finite else 'MISMATCH'
        bad += status != 'ok'
        print(f'{dtype} weights, batch {batch} x {frames} frames, '
              f'{repeats} runs: {status} (abs-max {first.abs().max().item():.3f})',
              flush=True)
promonet_amd.configure(MODEL='hifigan', FARGAN_WEIGHT_DTYPE='fp32')
print('fuzz_fargan:', 'FAILED' if bad else 'deterministic')
sys.exit(1 if bad else 0)
