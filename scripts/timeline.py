"""Phase timeline of the fused pair kernel at full size (debug tool, GPU box).
Stamps: 0 start, 1 first chunk staged, 2 conv1 done, 3 epilogue1+sync,
4 conv2 done, 5 end."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from promonet_amd import _lib  # noqa: E402

device = torch.device('cuda:0')
lib = _lib.lib()
for channels, length, k, d in ((256, 6888, 11, 5), (128, 55104, 11, 5),
                               (128, 55104, 7, 3), (128, 55104, 7, 1)):
    batch = 32
    x = torch.randn(batch, length, channels, device=device)
    out = torch.empty_like(x)
    w1 = torch.randn(channels, channels, k, device=device) * .01
    w2 = torch.randn(channels, channels, k, device=device) * .01
    b1 = torch.zeros(channels, device=device)
    b2 = torch.zeros(channels, device=device)
    ws = torch.empty(lib.pm_op_workspace_bytes(channels, channels, k),
                     dtype=torch.uint8, device=device)
    stamps = torch.zeros(1 << 17, 16, dtype=torch.int64, device=device)

    def run():
        _lib.check(lib.pm_block_iteration_cl(
            _lib.PM_BF16, _lib.ptr(x), _lib.ptr(out), _lib.ptr(w1),
            _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), batch, length, channels,
            k, d, 0, 1., ws.data_ptr(), ws.numel(), _lib.stream()))

    run()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(stamps.data_ptr())
    start, end = torch.cuda.Event(True), torch.cuda.Event(True)
    start.record()
    run()
    end.record()
    torch.cuda.synchronize()
    lib.pm_debug_timeline(None)
    t = stamps.cpu()
    used = t[:, 5] > 0
    t = t[used].double()
    n = t.shape[0]
    phases = (t[:, 1:6] - t[:, 0:5]).mean(0)
    total = (t[:, 5] - t[:, 0]).mean()
    span = (t[:, 5].max() - t[:, 0].min())
    print(f'C={channels} k={k} d={d}: {n} blocks, kernel {start.elapsed_time(end) * 1e3:.0f} us '
          f'(incl. packing), span {span:.0f} ticks, mean block {total:.0f} ticks')
    print('   stage %.0f | conv1 %.0f | epi1 %.0f | conv2 %.0f | epi2 %.0f' %
          tuple(phases.tolist()))
    if channels > 64:
        # stamps 6 / 8: end of chunk 0's MFMAs / after the barrier behind it
        print('   conv1 chunk 0: mma %.0f, barrier %.0f' % (
            (t[:, 6] - t[:, 1]).mean(), (t[:, 8] - t[:, 6]).mean()))
    # shader clock against the constant 100 MHz clock; idle gap between
    # consecutive workgroups of one CU (the place = XCC id, HW_ID[15:8])
    wall = (t[:, 13] - t[:, 12]) * 10.          # ns
    ghz = ((t[:, 5] - t[:, 0]) / wall).mean()
    raw = stamps.cpu()[used.nonzero().squeeze(1)]
    place = ((raw[:, 14] >> 32) << 8) | ((raw[:, 14] >> 8) & 0xff)
    gaps, busy = [], []
    for cu in place.unique().tolist():
        rows = raw[place == cu]
        rows = rows[rows[:, 12].argsort()]
        gaps.append(((rows[1:, 12] - rows[:-1, 13]).double() * 10.))
        busy.append(((rows[:, 13] - rows[:, 12]).double().sum() * 10.) /
                    ((rows[-1, 13] - rows[0, 12]).double() * 10.))
    # lockstep check: spread over the CUs of the start time of a CU's k-th tile
    kth = {}
    for cu in place.unique().tolist():
        rows = raw[place == cu]
        starts = rows[:, 12].sort().values
        for k in (1, 5, 10, 20):
            if k < starts.numel():
                kth.setdefault(k, []).append(starts[k].item())
    spread = ' '.join('k=%d: %.1f us' % (k, torch.tensor(v).double().std() * 1e-2)
                      for k, v in sorted(kth.items()))
    print('   std over CUs of the start of a CU\'s k-th tile: ' + spread)
    gaps = torch.cat(gaps)
    print('   shader clock %.2f GHz | %d CUs seen | gap between workgroups of a CU: mean %.0f ns, median %.0f ns | CU occupied %.1f %% of its span' % (
        ghz, place.unique().numel(), gaps.mean(), gaps.median(), 100 * torch.tensor(busy).mean()))
    print('   blocks/CU %.2f -> sum of block time per CU %.0f ticks' %
          (n / 256, n / 256 * total))
