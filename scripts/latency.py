"""Single-utterance latency of promonet_amd.synthesize.from_features (the
reference's batch-1 public API): eager launches vs a captured HIP graph."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

device = torch.device('cuda:0')
torch.manual_seed(0)
model = promonet_amd.model.Generator().to(device).eval()
promonet_amd.synthesize.set_model(model, device)
for seconds in (2, 10):
    frames = promonet_amd.convert.seconds_to_frames(seconds)
    inputs = synthetic_inputs(1, frames, 1, device)
    args = (inputs[0][0], inputs[1], inputs[2], inputs[3])
    for _ in range(3):
        promonet_amd.synthesize.from_features(*args, gpu=0)
    torch.cuda.synchronize()
    n = 20
    start = time.perf_counter()
    for _ in range(n):
        out = promonet_amd.synthesize.from_features(*args, gpu=0)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - start) / n
    # graph capture of Generator.forward
    static = [t.clone() for t in inputs]
    graph = torch.cuda.CUDAGraph()
    with torch.inference_mode():
        model(*static, None)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            captured = model(*static, None)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - start) / n
    with torch.inference_mode():
        want = model(*static, None)
    same = torch.equal(want, captured)
    print(f'{seconds} s utterance ({frames} frames): eager {eager * 1e3:.3f} ms '
          f'(RTF {seconds / eager:.0f}), graph replay {replay * 1e3:.3f} ms '
          f'(RTF {seconds / replay:.0f}), graph output identical: {same}')
