"""Phase totals of the skewed whole-Block walk at full size (debug tool, GPU
box, -DPM_TUNING build): shader clocks of wave 0 per phase, mean over the
workgroups, per step."""
import ctypes
import os
import sys
os.environ.setdefault('PROMONET_HIP_DEBUG', '1')   # the library's test hooks
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from promonet_amd import _lib  # noqa: E402

device = torch.device('cuda:0')
lib = _lib.lib()
NAMES = ('stage', 'conv1', 'epi1', 'barrier1', 'conv2', 'epi2', 'drain',
         'barrier2', 'store')
# usage: timeline_skew.py [dtype [channels length k ...]] (default: the bf16 shapes)
DTYPE = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
SHAPES = ((128, 55104, 11), (128, 55104, 7), (128, 55104, 3),
          (64, 110208, 11), (64, 110208, 7))
if len(sys.argv) > 2:
    flat = [int(v) for v in sys.argv[2:]]
    SHAPES = tuple(zip(flat[0::3], flat[1::3], flat[2::3]))
for channels, length, k in SHAPES:
    batch = 32
    x = torch.randn(batch, length, channels, device=device)
    out = torch.zeros_like(x)
    w = [torch.randn(channels, channels, k, device=device) * .01 for _ in range(6)]
    bias = [torch.zeros(channels, device=device) for _ in range(6)]
    per = lib.pm_op_workspace_bytes(channels, channels, k)
    ws = torch.empty(3 * per + lib.pm_walk_scratch_bytes(batch),
                     dtype=torch.uint8, device=device)
    stamps = torch.zeros(1 << 12, 16, dtype=torch.int64, device=device)
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    w1, w2, b1, b2 = arr(w[:3]), arr(w[3:]), arr(bias[:3]), arr(bias[3:])
    dil = (ctypes.c_int * 3)(1, 3, 5)

    def run():
        _lib.check(lib.pm_block_cl(
            _lib.DTYPES[DTYPE], _lib.ptr(x), _lib.ptr(out), w1, b1, w2, b2, dil, 3,
            batch, length, channels, k, 2, 1. / 3, ws.data_ptr(), ws.numel(),
            _lib.stream()))

    _lib.check(lib.pm_debug_skew(1))       # every shape on the skewed kernel
    run()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(stamps.data_ptr())
    start, end = torch.cuda.Event(True), torch.cuda.Event(True)
    start.record()
    run()
    end.record()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(None)
    _lib.check(lib.pm_debug_skew(0))
    t = stamps.cpu().double()
    t = t[t[:, 9] > 0]
    steps = t[:, 9].mean().item()
    mean = (t[:, :9].mean(0) / steps).tolist()
    print(f'C={channels} k={k}: {t.shape[0]} workgroups, {steps:.1f} steps each, '
          f'{start.elapsed_time(end) * 1e3:.0f} us incl. packing; cycles per step '
          f'{sum(mean):.0f}')
    print('   ' + ' | '.join(f'{n} {v:.0f}' for n, v in zip(NAMES, mean)))
