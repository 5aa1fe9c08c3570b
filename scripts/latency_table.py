"""GPU box: where a streaming chunk's time goes - per-kernel table of ONE
packed_inference call at B = 1, T = 8 / 64 frames (engine profile report)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import promonet_amd  # noqa: E402
from promonet_amd import _lib  # noqa: E402

device = torch.device('cuda:0')
torch.manual_seed(0)
model = promonet_amd.model.Generator().to(device).eval()
lib = _lib.lib()
for frames in (8, 64):
    x = torch.rand(1, 53, frames, device=device)
    with torch.inference_mode():
        for _ in range(3):
            model.packed_inference(x)
        engine = model.model.engine()
        lib.pm_hifigan_profile_only(engine, None)
        lib.pm_hifigan_profile_reset(engine)
        lib.pm_hifigan_profile_enable(engine, 1)
        for _ in range(10):
            model.packed_inference(x)
        lib.pm_hifigan_profile_enable(engine, 0)
        lib.pm_hifigan_profile_collect(engine)
        report = lib.pm_hifigan_profile_report(engine).decode()
    total = 0.
    print(f'--- B 1, T {frames} ({promonet_amd.COMPUTE_DTYPE}) us per call')
    for line in report.strip().splitlines():
        label, count, ms, flops, nbytes = line.split()
        us = float(ms) * 1e3 / 10
        total += us
        print(f'{label:18s} x{int(count) // 10}  {us:8.1f} us')
    print(f'sum of kernels {total:.1f} us')
