"""Phase timeline of one sub-frame step of the FARGAN cluster kernel (debug
tool for a -DPM_TUNING build, GPU box): s_memtime stamps of member 1 of
cluster 0 at frame 7, sub-frame 1, in cycles of the shader clock (~2.1 GHz
under this load; FARGAN_WEIGHT_DTYPE picks the weight storage)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd  # noqa: E402
from promonet_amd import _lib  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

device = torch.device('cuda:0')
import os  # noqa: E402
promonet_amd.configure(
    MODEL='fargan',
    FARGAN_WEIGHT_DTYPE=os.environ.get('FARGAN_WEIGHT_DTYPE', 'fp32'))
torch.manual_seed(0)
model = promonet_amd.model.Generator().to(device).eval()
lib = _lib.lib()
for batch in (32, 64):
    inputs = synthetic_inputs(batch, 40, 1234, device)
    stamps = torch.zeros(64, dtype=torch.int64, device=device)
    with torch.inference_mode():
        model(*inputs, None)
        torch.cuda.synchronize()
        _lib.check(lib.pm_debug_timeline(stamps.data_ptr()))
        model(*inputs, None)
        torch.cuda.synchronize()
        lib.pm_debug_timeline(None)
    t = stamps.cpu().tolist()
    names = {1: 'fwconv rows', 2: 'fwconv_glu cols', 3: 'exchange sum'}
    for n in range(3):
        names.update({4 + 4 * n: f'gru{n} ih+hh rows', 5 + 4 * n: f'gru{n} cell',
                      6 + 4 * n: f'glu{n} cols', 7 + 4 * n: f'exchange sum'})
    names.update({16: '-', 17: 'skip rows', 18: 'exchange vec',
                  19: 'skip_glu rows', 20: 'out cols', 21: 'exchange sum',
                  22: 'state update'})
    total = t[22] - t[0]
    print(f'batch {batch}: step total {total} cycles')
    exchanges = 0
    for i in range(1, 23):
        print(f'  {names[i]:18s} {t[i] - t[i - 1]:6d} cycles')
        if names[i].startswith('exchange'):
            exchanges += t[i] - t[i - 1]
    print(f'  exchanges: {exchanges} cycles = {exchanges / total:.2f} of the step')
