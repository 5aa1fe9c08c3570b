"""Which FARGAN weights feel their storage type (CPU, a few minutes).

The oracle synthesises a whole 10 s utterance (3 444 dependent sub-frame
steps) with one group of matrices at a time rounded to f16 (everything else
fp32) and prints the max-abs / rms deviation from the all-fp32 audio; then the
'mixed' storage of promonet_amd (FARGAN_WEIGHT_DTYPE='mixed': GRU cells and
GLU gates f16, the other layers fp32) for three weight seeds, also with the
output layer scaled up (a louder, strongly fed-back random model - where any
perturbation, fp32 summation order included, is amplified). Test
infrastructure: runs the oracle only.
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'oracle'))
import restatement as R
torch.set_num_threads(8)
torch.manual_seed(0)
state = R.random_state_fargan(seed=0)
B, T = 2, 861
inp = R.synthetic_inputs(B, T, seed=1234)
def run(st):
    with torch.no_grad():
        return R.fargan_generator_forward(*inp, st)
t0 = time.time(); ref = run(state); print('ref', time.time() - t0, 's peak', ref.abs().max().item())
p = 'model.subframe_network.'
groups = {
 'cond': [f'model.conditioning_network.{k}.weight' for k in (0, 2, 4)],
 'fwconv': [p + 'framewise_convolution.model.0'],
 'fwglu': [p + 'framewise_convolution.model.2.gate'],
 'gru_ih': [p + f'gru{n}.weight_ih' for n in (1, 2, 3)],
 'gru_hh': [p + f'gru{n}.weight_hh' for n in (1, 2, 3)],
 'gru_glu': [p + f'gru{n}_glu.gate' for n in (1, 2, 3)],
 'skip': [p + 'skip_dense.weight'],
 'skip_glu': [p + 'skip_glu.gate'],
 'out': [p + 'output_layer.weight'],
}
def quant(wt, mode):
    if mode == 'f16': return wt.half().float()
    if mode == 'bf16': return wt.bfloat16().float()
    if mode.startswith('m'):  # keep N explicit mantissa bits (fp32 exponent), round to nearest
        n = int(mode[1:]); drop = 23 - n
        i = wt.view(torch.int32).clone()
        i = ((i + (1 << (drop - 1))) >> drop) << drop
        return i.view(torch.float32)
def with_quant(names, mode):
    st = dict(state)
    for name in names:
        if name in st:
            st[name] = quant(st[name], mode)
        else:
            # weight-normed: quantize the folded weight, store as plain weight
            folded = R.fold_weight_norm_linear(st[name + '.weight_g'], st[name + '.weight_v'])
            st = {k: v for k, v in st.items() if not k.startswith(name + '.weight_')}
            st[name + '.weight'] = quant(folded, mode)
    return st
allnames = [n for g in groups.values() for n in g]
for mode in ('f16', 'm15', 'm14', 'm13', 'm12'):
    d = (run(with_quant(allnames, mode)) - ref).abs()
    print(f'all {mode}: max {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}', flush=True)
for g, names in groups.items():
    d = (run(with_quant(names, 'f16')) - ref).abs()
    print(f'{g} f16 only: max {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e}', flush=True)
print('--- mixed')
f16_groups = ['gru_ih', 'gru_hh', 'gru_glu', 'fwglu', 'skip_glu']
for seed in (0, 1, 2):
    state = R.random_state_fargan(seed=seed)
    for boost in (1., 6.):
        state2 = dict(state)
        state2[p + 'output_layer.weight'] = state[p + 'output_layer.weight'] * boost
        state_save = state
        state = state2
        ref = run(state)
        names = [n for g in f16_groups for n in groups[g]]
        d = (run(with_quant(names, 'f16')) - ref).abs()
        d2 = (run(with_quant(allnames, 'f16')) - ref).abs()
        print(f'seed {seed} boost {boost}: peak {ref.abs().max().item():.3f} mixed max {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} | all-f16 max {d2.max().item():.3e}', flush=True)
        state = state_save
