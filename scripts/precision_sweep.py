"""Precision of the MFMA operand types at random-init AND at trained-checkpoint
output scale (GPU box). VERDICT r02 #6: the 1e-4 gate is met on the random-init
default model only because its output peaks at 0.017; a trained generator's
audio peaks near 1. Here the output conv is rescaled so that the audio peaks at
~0.5 (same arithmetic everywhere else, so the relative error of the trunk is
unchanged) and every operand-type choice - including per-stage mixes - is
compared, on every sample, with the fp32 CPU oracle on the same weights.
Round 4: a second trained-scale column at peak 0.99 and the split-f16 modes
('checkpoint' = f16, f16x3 in the last stage; 'f16x3' everywhere).

usage: python scripts/precision_sweep.py [batch] [frames]"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import promonet_amd  # noqa: E402
import restatement as oracle  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 861
device = torch.device('cuda:0')
VARIANTS = ['fp32', 'checkpoint', 'f16x3', 'f16', 'bf16', 'bf16+bf16+f16+f16',
            'bf16+f16+f16+f16', 'bf16+bf16+bf16+f16', 'bf16+bf16+bf16+f16x3',
            'f16+f16+bf16+bf16']


def reference(inputs, state):
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, threads))
    try:
        with torch.inference_mode():
            return torch.cat([
                oracle.generator_forward(*[t[i:i + 4] for t in inputs], state)
                for i in range(0, batch, 4)])
    finally:
        torch.set_num_threads(threads)


def run(state, inputs, dtype, time_it):
    promonet_amd.configure(COMPUTE_DTYPE=dtype)
    model = promonet_amd.model.Generator()
    model.load_state_dict(state)
    model = model.to(device).eval()
    args = [t.to(device) for t in inputs]
    with torch.inference_mode():
        got = model(*args, None)
        ms = None
        if time_it:
            for _ in range(2):
                model(*args, None)
            torch.cuda.synchronize()
            start = time.perf_counter()
            for _ in range(5):
                model(*args, None)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - start) / 5 * 1e3
    return got.cpu(), ms


golden = torch.load(ROOT / 'tests/golden/generator_default.pt')
state = oracle.random_state(seed=golden['seed'])
state['pitch_distribution'] = golden['pitch_distribution'].clone()
inputs = oracle.synthetic_inputs(batch, frames, seed=1234)
result = {'batch': batch, 'frames': frames, 'cases': {}}
base_state = state
for label, peak_target in (
        ('random_init', None), ('trained_scale', .5), ('trained_scale_0.99', .99)):
    if peak_target is not None:
        # audio = tanh(conv(...)): scale the output conv so that it peaks at
        # `peak_target` (0.99: the pre-tanh amplitude is 4.8x that of 0.5)
        peak = result['cases']['random_init']['reference_abs_max']
        factor = float(torch.atanh(torch.tensor(peak_target))) / float(
            torch.atanh(torch.tensor(peak)))
        state = dict(base_state)
        key = [k for k in state if k.endswith('model.5.weight')][0]
        state[key] = state[key] * factor
        result['cases'][label] = {'output_conv_scale': factor}
    else:
        result['cases'][label] = {}
    want = reference(inputs, state)
    entry = result['cases'][label]
    entry['reference_abs_max'] = want.abs().max().item()
    entry['variants'] = {}
    for dtype in VARIANTS:
        got, ms = run(state, inputs, dtype, label == 'random_init')
        diff = (got - want).abs()
        row = {
            'max_abs': diff.max().item(),
            'rms': diff.pow(2).mean().sqrt().item(),
            'max_abs_over_output_abs_max':
                diff.max().item() / entry['reference_abs_max']}
        if ms is not None:
            row['ms_per_step'] = ms
        entry['variants'][dtype] = row
        print(label, dtype, row, flush=True)
promonet_amd.configure(COMPUTE_DTYPE=promonet_amd.config.DEFAULT_COMPUTE_DTYPE)
print(json.dumps(result))
