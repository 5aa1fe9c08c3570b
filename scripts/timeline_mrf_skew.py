"""Phase totals of the skewed whole-MRF walk (conv_mrf_skew_kernel) at full
size (debug tool, GPU box, -DPM_TUNING build): shader clocks of wave 0 per phase
and Block, mean over the workgroups, per step. usage: [dtype]"""
import ctypes
import os
import sys
os.environ.setdefault('PROMONET_HIP_DEBUG', '1')
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from promonet_amd import _lib  # noqa: E402

device = torch.device('cuda:0')
lib = _lib.lib()
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16a2'
channels, length, batch = 32, 220416, 32
gen = torch.Generator().manual_seed(0)
order = {'w1': [], 'b1': [], 'w2': [], 'b2': []}
for k in (3, 7, 11):
    for n in range(3):
        for which in (1, 2):
            order[f'w{which}'].append((torch.randn(channels, channels, k, generator=gen) / (channels * k) ** .5).to(device))
            order[f'b{which}'].append((torch.randn(channels, generator=gen) * .1).to(device))
ptrs = lambda name: (ctypes.c_void_p * 9)(*[t.data_ptr() for t in order[name]])
dil = (ctypes.c_int * 3)(1, 3, 5)
ws = torch.empty(9 * lib.pm_op_workspace_bytes(channels, channels, 11) + lib.pm_walk_scratch_bytes(batch), dtype=torch.uint8, device=device)
x = torch.randn(batch, length, channels, device=device)
out = torch.zeros_like(x)
stamps = torch.zeros(3 * 256, 16, dtype=torch.int64, device=device)
NAMES = ('stage', 'conv1', 'epi1', 'barrier1', 'conv2', 'epi2', '-', 'barrier2', 'sum/store')


def run():
    _lib.check(lib.pm_mrf_cl(_lib.DTYPES[dtype], _lib.ptr(x), _lib.ptr(out), ptrs('w1'), ptrs('b1'), ptrs('w2'), ptrs('b2'), dil, 3, batch, length, channels, ws.data_ptr(), ws.numel(), _lib.stream()))


run(); torch.cuda.synchronize()
lib.pm_debug_timeline(stamps.data_ptr())
start, end = torch.cuda.Event(True), torch.cuda.Event(True)
start.record(); run(); end.record(); torch.cuda.synchronize()
lib.pm_debug_timeline(None)
t = stamps.cpu().double().reshape(256, 3, 16)
t = t[t[:, 0, 9] > 0]
print(f'{dtype}: {t.shape[0]} workgroups, launch {start.elapsed_time(end) * 1e3:.0f} us incl. packing')
total = 0
for j, k in enumerate((11, 7, 3)):
    steps = t[:, j, 9].mean().item()
    mean = (t[:, j, :9].mean(0) / steps).tolist()
    total += sum(mean)
    print(f'  k={k}: {steps:.1f} steps, cycles per step {sum(mean):.0f}: ' + ' | '.join(f'{n} {v:.0f}' for n, v in zip(NAMES, mean) if n != '-'))
print(f'  all three: {total:.0f} cycles per 512-column step')
