"""Phase timeline of the whole-Block kernel at full size (debug tool, GPU box).
Stamps: 0 start, 1 staged, then per iteration conv1 / epilogue1+sync / conv2 /
epilogue2+sync, 14 end (after the HBM store)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from promonet_amd import _lib  # noqa: E402

device = torch.device('cuda:0')
lib = _lib.lib()
import ctypes
for channels, length, k in ((32, 220416, 3), (32, 220416, 7), (32, 220416, 11),
                            (64, 110208, 3), (64, 110208, 11), (128, 55104, 3)):
    batch = 32
    x = torch.randn(batch, length, channels, device=device)
    out = torch.zeros_like(x)
    w = [torch.randn(channels, channels, k, device=device) * .01 for _ in range(6)]
    bias = [torch.zeros(channels, device=device) for _ in range(6)]
    per = lib.pm_op_workspace_bytes(channels, channels, k)
    ws = torch.empty(3 * per, dtype=torch.uint8, device=device)
    stamps = torch.zeros(1 << 16, 16, dtype=torch.int64, device=device)
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    w1, w2, b1, b2 = arr(w[:3]), arr(w[3:]), arr(bias[:3]), arr(bias[3:])
    dil = (ctypes.c_int * 3)(1, 3, 5)

    def run():
        _lib.check(lib.pm_block_cl(
            _lib.PM_F16, _lib.ptr(x), _lib.ptr(out), w1, b1, w2, b2, dil, 3,
            batch, length, channels, k, 2, 1. / 3, ws.data_ptr(), ws.numel(),
            _lib.stream()))

    run()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(None)
    t = stamps.cpu()
    t = t[t[:, 14] > 0].double()
    n = t.shape[0]
    order = list(range(14)) + [14]
    seq = t[:, order]
    phases = (seq[:, 1:] - seq[:, :-1]).mean(0).tolist()
    total = (t[:, 14] - t[:, 0]).mean().item()
    span = (t[:, 14].max() - t[:, 0].min()).item()
    print(f'C={channels} k={k}: {n} blocks ({n / 256:.1f}/CU), span {span:.0f} '
          f'ticks, mean block {total:.0f} ticks (100 MHz ticks)')
    print('   stage %.0f' % phases[0])
    for it in range(3):
        print('   it%d: conv1 %.0f | epi1 %.0f | conv2 %.0f | epi2 %.0f' %
              (it, *phases[1 + 4 * it:5 + 4 * it]))
    print('   store %.0f' % phases[13])
