"""Stress check on the GPU box: random ragged batches through the batched
forward must equal the stand-alone synthesis of every utterance bit for bit
(wide and narrow tile variants, every utterance-edge / tile-edge alignment the
buffer-descriptor addressing has to get right), tails must be zero.
usage: python scripts/fuzz_ragged.py [trials] [seed] [max batch] [force]
(batches of 20+ utterances of 100+ frames run the wide tile variants; max
batch <= 4 switches to long utterances, which run the walked kernels; `force`
= 1: the batched run is forced onto the walked / skewed kernels with 1 ... 5
segments per utterance (pm_debug_force / pm_debug_skew), the stand-alone runs
onto the plain tilings)"""
import random
import os
import sys
os.environ.setdefault('PROMONET_HIP_DEBUG', '1')   # the library's test hooks
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import restatement as oracle  # noqa: E402  (test infrastructure: inputs, weights)
import promonet_amd  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
max_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 6
force = len(sys.argv) > 4 and sys.argv[4] == '1'
from promonet_amd import _lib  # noqa: E402
device = torch.device('cuda:0')
state = oracle.random_state(seed=0)
models = {}
for dtype in ('bf16', 'f16', 'fp32', 'checkpoint'):
    promonet_amd.configure(COMPUTE_DTYPE=dtype)
    model = promonet_amd.model.Generator()
    model.load_state_dict(state)
    models[dtype] = model.to(device).eval()
promonet_amd.configure(COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)

bad = 0
for trial in range(trials):
    dtype = ('bf16', 'f16', 'fp32', 'checkpoint')[trial % 4]
    model = models[dtype]
    batch = rng.randint(max(1, max_batch // 2), max_batch)
    # (a few long utterances in a small batch: the walked kernels with many
    # short, uneven segments against the stand-alone tiling of batch 1)
    top = rng.choice((3, 20, 60, 150, 300) if max_batch > 4 else (900, 2500))
    lengths = [rng.randint(1, top) for _ in range(batch)]
    frames = max(lengths)
    inputs = [t.to(device) for t in
              oracle.synthetic_inputs(batch, frames, seed=100 + trial)]
    for item, length in enumerate(lengths):
        for tensor in inputs[:4]:
            tensor[item, ..., length:] = 7.          # garbage past the end
    with torch.inference_mode():
        if force:
            nseg = rng.randint(1, 5)
            _lib.check(_lib.lib().pm_debug_force(nseg, 0))
            _lib.check(_lib.lib().pm_debug_skew(rng.choice((0, 1))))
        ragged = model(*inputs, None, lengths=lengths)
        if force:
            _lib.check(_lib.lib().pm_debug_force(0, 0))
            _lib.check(_lib.lib().pm_debug_skew(-1))
        for item, length in enumerate(lengths):
            single = model(
                *[t[item:item + 1, ..., :length] if t.ndim >= 2
                  else t[item:item + 1] for t in inputs], None)
            same = torch.equal(ragged[item, :, :length * 256], single[0])
            tail = ragged[item, :, length * 256:]
            clean = tail.numel() == 0 or tail.abs().max().item() == 0.
            finite = bool(torch.isfinite(single).all())
            if not (same and clean and finite):
                bad += 1
                print(f'MISMATCH trial {trial} {dtype} lengths {lengths} '
                      f'item {item}: same {same} tail-zero {clean} '
                      f'finite {finite}')
    print(f'trial {trial} {dtype} lengths {lengths}: ok', flush=True)
if force:
    _lib.check(_lib.lib().pm_debug_skew(0))
print('fuzz_ragged:', 'FAILED %d' % bad if bad else 'all %d trials exact' % trials)
sys.exit(1 if bad else 0)
