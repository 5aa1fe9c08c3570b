"""'mixed' / f16 / fp32 FARGAN weight storage against the fp32 oracle for other
weight seeds than the tests' (GPU box): whole 10 s utterances, max-abs."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402

device = torch.device('cuda:0')
torch.set_num_threads(8)
for seed in (1, 2, 3):
    state = oracle.random_state_fargan(seed=seed)
    state['pitch_distribution'] = promonet_amd.load.pitch_distribution()
    inputs = oracle.synthetic_inputs(2, 861, seed=100 + seed)
    with torch.inference_mode():
        want = oracle.fargan_generator_forward(*inputs, state)
    line = [f'seed {seed}: abs-max {want.abs().max().item():.3f}']
    for storage in ('fp32', 'mixed', 'f16'):
        promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=storage)
        model = promonet_amd.model.Generator()
        model.load_state_dict(state)
        model = model.to(device).eval()
        with torch.inference_mode():
            got = model(*[t.to(device) for t in inputs], None).cpu()
        line.append(f'{storage} {(got - want).abs().max().item():.3e}')
    print(' | '.join(line), flush=True)
promonet_amd.configure(MODEL='hifigan', FARGAN_WEIGHT_DTYPE='fp32')
