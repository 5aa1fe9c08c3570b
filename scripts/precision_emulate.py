"""CPU emulation of the MFMA operand roundings (no GPU): which operand format in
which stage holds the 1e-4 max-abs gate at TRAINED-checkpoint output scale?

Every conv of the generator is evaluated as the kernels evaluate it - operands
rounded to the 16-bit type, products and sums in fp32 - with the rounding chosen
per upsampling stage:
  f16 / bf16   one rounding of activations and weights (one MFMA per k16 step)
  <t>w2        weights as hi + lo (two MFMAs sharing the activation fragment)
  <t>a2        activations as hi + lo (two MFMAs sharing the weight fragment)
  <t>x3        both split, the lo x lo product dropped (three MFMAs)
  <t>x3c2      x3 on the conv2 of every Block iteration only (x3c1 / a2c1: on
               the conv1 - and the upsampler - only)
The output conv is rescaled so that the audio peaks at `peak` (every rounding
inside the trunk stays what it is). Used to choose what to build
(DESIGN.md section 3); the GPU numbers come from scripts/precision_sweep.py.

usage: python scripts/precision_emulate.py [peak] [batch] [frames]"""
import math
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'oracle'))
import restatement as oracle  # noqa: E402

peak = float(sys.argv[1]) if len(sys.argv) > 1 else 1.
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 120


def split(x, fmt):
    """(hi, lo, use_lo_a, use_lo_w) pieces of an operand in format `fmt`."""
    base = torch.float16 if fmt.startswith('f16') else torch.bfloat16
    hi = x.to(base).to(torch.float32)
    lo = (x - hi).to(base).to(torch.float32)
    return hi, lo


def conv(x, w, bias, fmt, second=False, role=None, **kw):
    """One convolution with operand format `fmt` (None: exact fp32).
    `fmt` may carry a per-conv schedule: 'f16:up=x3,c1=x3,c2=a2,c1k3=a2' - the
    split of the upsampler / the conv1s / the conv2s of the stage, optionally
    of one kernel size only (role = ('c1', 11)); unnamed convs: one rounding."""
    op = kw.pop('op', F.conv1d)
    if fmt is None or fmt == 'fp32':
        return op(x, w, bias, **kw)
    xh, xl = split(x, fmt)
    wh, wl = split(w, fmt)
    if ':' in fmt:
        table = dict(item.split('=') for item in fmt.split(':')[1].split(','))
        which, k = role
        suffix = table.get(f'{which}k{k}', table.get(which, ''))
        out = op(xh, wh, bias, **kw)
        if suffix in ('a2', 'x3'):
            out = out + op(xl, wh, None, **kw)
        if suffix in ('w2', 'x3'):
            out = out + op(xh, wl, None, **kw)
        return out
    suffix = fmt[3:] if fmt.startswith('f16') else fmt[4:]
    if suffix == 'x3c2':
        suffix = 'x3' if second else ''
    if suffix == 'x3c1':
        suffix = '' if second else 'x3'
    if suffix == 'a2c1':
        suffix = '' if second else 'a2'
    out = op(xh, wh, bias, **kw)
    if suffix in ('a2', 'x3'):
        out = out + op(xl, wh, None, **kw)
    if suffix in ('w2', 'x3'):
        out = out + op(xh, wl, None, **kw)
    return out


def forward(features, glob, w, formats):
    _, rates, kernels = oracle.hifigan_config(w)
    x = conv(features, w['model.input_feature_conv.weight'],
             w['model.input_feature_conv.bias'], formats[0], role=('in', 7),
             padding=3)
    x = x + F.conv1d(glob, w['model.input_speaker_conv.weight'],
                     w['model.input_speaker_conv.bias'])
    for i, (r, k) in enumerate(zip(rates, kernels)):
        fmt = formats[i]
        x = F.leaky_relu(x, .1)
        x = conv(x, w[f'model.model.{i}.model.1.weight'],
                 w[f'model.model.{i}.model.1.bias'], fmt, role=('up', 0),
                 op=F.conv_transpose1d, stride=r, padding=(k - r) // 2)
        xs = None
        for j, (ks, dil) in enumerate(zip(
                oracle.RESBLOCK_KERNEL_SIZES, oracle.RESBLOCK_DILATION_SIZES)):
            y = x
            prefix = f'model.model.{i}.model.2.model.{j}'
            for n, d in enumerate(dil):
                t = F.leaky_relu(y, .1)
                t = conv(t, w[f'{prefix}.convs1.{n}.weight'],
                         w[f'{prefix}.convs1.{n}.bias'], fmt, role=('c1', ks),
                         padding=oracle.get_padding(ks, d), dilation=d)
                t = F.leaky_relu(t, .1)
                t = conv(t, w[f'{prefix}.convs2.{n}.weight'],
                         w[f'{prefix}.convs2.{n}.bias'], fmt, second=True,
                         role=('c2', ks), padding=oracle.get_padding(ks, 1))
                y = y + t
            xs = y if xs is None else xs + y
        x = xs / 3
    x = F.leaky_relu(x, .1)
    x = F.conv1d(x, w[f'model.model.{len(rates) + 1}.weight'], None, padding=3)
    return torch.tanh(x)


golden = torch.load(ROOT / 'tests/golden/generator_default.pt')
state = oracle.random_state(seed=golden['seed'])
state['pitch_distribution'] = golden['pitch_distribution'].clone()
inputs = oracle.synthetic_inputs(batch, frames, seed=99)
torch.set_num_threads(8)
with torch.inference_mode():
    raw = oracle.generator_forward(*inputs, state).abs().max()
    key = 'model.model.5.weight'
    state[key] = state[key] * (math.atanh(peak) / math.atanh(float(raw)))
    w = oracle.folded_state(state)
    features = oracle.prepare_features(
        *inputs[:4], state['pitch_distribution'],
        state['pitch_embedding.weight'], state['ppg_threshold'])
    glob = oracle.prepare_global_features(
        *inputs[4:7], state['speaker_embedding.weight'])
    want = forward(features, glob, w, [None] * 4)
    print(f'peak {want.abs().max().item():.3f} (batch {batch} x {frames} frames)')
    variants = [
        'bf16', 'f16', 'bf16+bf16+bf16+f16',
        'f16+f16+f16+f16w2', 'f16+f16+f16+f16a2', 'f16+f16+f16+f16x3c2',
        'f16+f16+f16+f16x3', 'f16+f16+f16x3+f16x3', 'f16x3',
        'bf16+bf16+bf16+bf16x3', 'bf16+bf16+bf16x3+bf16x3', 'bf16x3',
        'f16+f16+f16w2+f16x3', 'f16+f16w2+f16w2+f16x3',
    ] if len(sys.argv) <= 4 else sys.argv[4:]
    for name in variants:
        formats = name.split('+') if '+' in name else [name] * 4
        got = forward(features, glob, w, formats)
        diff = (got - want).abs()
        print(f'{name:28s} max-abs {diff.max().item():.3e}  rms '
              f'{diff.pow(2).mean().sqrt().item():.3e}', flush=True)
