"""FARGAN throughput (config 5): batch x 10 s utterances. Run on the GPU box.
PM_FARGAN=single|cluster forces a kernel, BATCHES=1,32,... picks the sizes."""
import os
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import synthetic_inputs  # noqa: E402

device = torch.device('cuda:0')
frames = 861
results = {}
for dtype in ('fp32', 'mixed', 'f16'):
    promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=dtype)
    torch.manual_seed(0)
    model = promonet_amd.model.Generator().to(device).eval()
    for batch in [int(b) for b in os.environ.get('BATCHES', '1,32,256').split(',')]:
        inputs = synthetic_inputs(batch, frames, 1234, device)
        with torch.inference_mode():
            model(*inputs, None)
            torch.cuda.synchronize()
            start = time.perf_counter()
            steps = 2
            for _ in range(steps):
                model(*inputs, None)
            torch.cuda.synchronize()
        seconds = (time.perf_counter() - start) / steps
        samples = batch * frames * 256
        results[f'{dtype}_b{batch}'] = {
            'ms_per_step': seconds * 1e3, 'samples_per_s': samples / seconds,
            'rtf': samples / 22050 / seconds,
            'us_per_subframe_step': seconds / (frames * 4) * 1e6}
        print(dtype, batch, results[f'{dtype}_b{batch}'])
promonet_amd.configure(MODEL='hifigan', FARGAN_WEIGHT_DTYPE='fp32')
print(json.dumps(results))
