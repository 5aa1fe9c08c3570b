"""Probe: does the chip do more when the batch runs as two out-of-phase halves?

Batch 32 x 10 s as ONE forward on one stream (what bench.py times) against two
16-utterance forwards on two streams (two model instances: a module's
workspace serves one stream), the second started half a step late, so that one
half's HBM-bound launches (upsamplers, output layer: 8 % of a step) meet the
other half's matrix-bound ones. Run on the GPU box; prints ms per 32 utterances.
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

device = torch.device('cuda:0')
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
promonet_amd.configure(COMPUTE_DTYPE=dtype)
frames = 861


def build():
    torch.manual_seed(0)
    return promonet_amd.model.Generator().to(device).eval()


def timed(step, steps=20, warmup=3):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / steps * 1e3


with torch.inference_mode():
    whole = build()
    inputs = synthetic_inputs(32, frames, 1234, device)
    for r in range(2):
        print(f'one stream, batch 32: {timed(lambda: whole(*inputs, None)):.3f} ms')
    halves = [build(), build()]
    parts = [[t[:16].contiguous() for t in inputs],
             [t[16:].contiguous() for t in inputs]]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def both():
        for model, part, stream in zip(halves, parts, streams):
            with torch.cuda.stream(stream):
                model(*part, None)

    # half a step of head start for the first stream
    with torch.cuda.stream(streams[0]):
        for _ in range(1):
            halves[0](*parts[0], None)
    for r in range(3):
        print(f'two streams, 2 x batch 16: {timed(both):.3f} ms')
    one = halves[0]
    print(f'one stream, batch 16: {timed(lambda: one(*parts[0], None)):.3f} ms '
          '(x 2 for 32 utterances)')
