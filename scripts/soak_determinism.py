"""Soak on the GPU box: the full-size forward (batch 32 x 861 frames) repeated,
every output compared bit for bit with the first one - with and without a
second stream hammering HBM beside it (the skewed walk's scratch hand-overs,
the walked kernels' LDS carries and the buffer-descriptor edges must not depend
on timing). usage: python scripts/soak_determinism.py [repeats]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import restatement as oracle  # noqa: E402  (test infrastructure: inputs, weights)
import promonet_amd  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 200
device = torch.device('cuda:0')
state = oracle.random_state(seed=0)
inputs = [t.to(device) for t in oracle.synthetic_inputs(32, 861, seed=1234)]
bad = 0
for dtype in ('bf16', 'f16', 'checkpoint'):
    promonet_amd.configure(COMPUTE_DTYPE=dtype)
    model = promonet_amd.model.Generator()
    model.load_state_dict(state)
    model = model.to(device).eval()
    side = torch.cuda.Stream()
    junk = torch.empty(1 << 28, dtype=torch.uint8, device=device)
    with torch.inference_mode():
        first = model(*inputs, None).clone()
        for noisy in (False, True):
            for i in range(repeats):
                if noisy and i % 2 == 0:
                    with torch.cuda.stream(side):
                        junk[:1 << 27].copy_(junk[1 << 27:])
                out = model(*inputs, None)
                if not torch.equal(out, first):
                    bad += 1
                    diff = (out - first).abs()
                    print(f'{dtype} noisy {noisy} repeat {i}: differs, max '
                          f'{diff.max().item():.3e} at '
                          f'{(diff.amax(-1).flatten() > 0).nonzero()[:4].tolist()}')
            torch.cuda.synchronize()
            print(f'{dtype} noisy {noisy}: {repeats} repeats done', flush=True)
promonet_amd.configure(COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
print('soak_determinism:', 'FAILED %d' % bad if bad else 'every output identical')
sys.exit(1 if bad else 0)
