"""Which operand schedule for the library default ('checkpoint')? GPU box.
All 7 053 312 samples of batch 32 x 861 frames at a trained checkpoint's
output scale (output conv rescaled so that the audio peaks at 0.99), each
candidate against the fp32 CPU oracle (the rescaled reference is
tanh(f atanh(audio)) of the oracle's, as tests/test_gpu_model.py does), with the
step time of the same model beside it.
usage: python scripts/checkpoint_schedule.py [candidate ...]"""
import math
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402

device = torch.device('cuda:0')
candidates = sys.argv[1:] or [
    'f16+f16+f16+f16x3', 'f16+f16+f16ux+f16x3', 'f16+f16+f16+f16a2',
    'f16+f16+f16ux+f16a2', 'f16+f16+f16+f16']
batch, frames = 32, 861
golden = torch.load(ROOT / 'tests/golden/generator_default.pt')
state = oracle.random_state(seed=golden['seed'])
state['pitch_distribution'] = golden['pitch_distribution'].clone()
inputs = oracle.synthetic_inputs(batch, frames, seed=1234)
torch.set_num_threads(8)
start = time.perf_counter()
with torch.inference_mode():
    want = torch.cat([
        oracle.generator_forward(*[t[i:i + 4] for t in inputs], state)
        for i in range(0, batch, 4)])
print(f'oracle: {time.perf_counter() - start:.0f} s', flush=True)
peak = want.abs().max().item()
factor = math.atanh(.99) / math.atanh(peak)
scaled = dict(state)
scaled['model.model.5.weight'] = scaled['model.model.5.weight'] * factor
want_scaled = torch.tanh(factor * torch.atanh(want.double()))
args = [t.to(device) for t in inputs]
for name in candidates:
    promonet_amd.configure(COMPUTE_DTYPE=name)
    model = promonet_amd.model.Generator()
    model.load_state_dict(scaled)
    model = model.to(device).eval()
    with torch.inference_mode():
        got = model(*args, None)
        for _ in range(2):
            model(*args, None)
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(8):
            model(*args, None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - start) / 8 * 1e3
    diff = (got.cpu().double() - want_scaled).abs()
    print(f'{name:24s} max-abs {diff.max().item():.3e}  rms '
          f'{diff.pow(2).mean().sqrt().item():.3e}  step {ms:.2f} ms',
          flush=True)
    del model
