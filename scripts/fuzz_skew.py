"""Stress check on the GPU box: the skewed whole-Block walk (forced, random
segment counts) against the other tilings of the same Block, bit for bit, for
random channel counts, kernel sizes, dilation sets (any the ABI admits, not
only the reference's 1 / 3 / 5), iteration counts, lengths, batch sizes and
store modes. usage: python scripts/fuzz_skew.py [trials] [seed]"""
import ctypes
import random
import os
import sys
os.environ.setdefault('PROMONET_HIP_DEBUG', '1')   # the library's test hooks
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from promonet_amd import _lib  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WIDE = len(sys.argv) > 3 and sys.argv[3] == 'wide'   # fp32 / f16x3 shapes
device = torch.device('cuda:0')
lib = _lib.lib()
bad = skipped = 0
for trial in range(trials):
    channels = rng.choice((64, 128, 128, 256, 50, 100))
    k = rng.choice((3, 7, 11)) if channels <= 128 else rng.choice((3, 7))
    if channels in (64, 50) and k == 3:
        k = 7                       # (C = 64 k 3 has a 4-wave tiling only)
    if WIDE:
        channels = rng.choice((32, 64, 20, 50))
        k = rng.choice((3, 7, 11))
    niter = rng.randint(1, 3)
    h2 = k // 2
    dmax = min(5, 30 // h2 - 1)
    dil = [rng.randint(1, dmax) for _ in range(niter)]
    batch = rng.randint(1, 3)
    length = rng.choice((rng.randint(1, 200), rng.randint(200, 3000),
                         rng.randint(3000, 9000)))
    nseg = rng.randint(1, 5)
    mode = rng.randint(0, 2)
    gen = torch.Generator().manual_seed(trial)
    std = 1. / (channels * k) ** .5
    w = [[(torch.randn(channels, channels, k, generator=gen) * std).to(device)
          for _ in range(niter)] for _ in range(2)]
    b = [[(torch.randn(channels, generator=gen) * .1).to(device)
          for _ in range(niter)] for _ in range(2)]
    arr = lambda ts: (ctypes.c_void_p * niter)(*[t.data_ptr() for t in ts])
    cpad = (channels + 31) // 32 * 32
    x = torch.zeros(batch, length, cpad)
    x[..., :channels] = torch.randn(batch, length, channels, generator=gen)
    prev = torch.zeros(batch, length, cpad)
    prev[..., :channels] = torch.randn(batch, length, channels, generator=gen)
    x = x.to(device)
    weights = 3 * lib.pm_op_workspace_bytes(channels, channels, k)
    ws = torch.empty(weights + lib.pm_walk_scratch_bytes(batch),
                     dtype=torch.uint8, device=device)
    dtype = rng.choice(('bf16', 'f16'))
    if WIDE:
        # the 4-byte operand layouts (exact fp32, split f16): whole-Block
        # tilings - and the skewed walk - exist for C <= 64 (every k)
        dtype = rng.choice(('fp32', 'f16x3'))
    outs = []
    for skew, size in ((1, ws.numel()), (-1, weights)):
        out = prev.clone().to(device)
        _lib.check(lib.pm_debug_force(nseg, 0))
        _lib.check(lib.pm_debug_skew(skew))
        ws[weights:].fill_(0xff)
        rc = lib.pm_block_cl(
            _lib.DTYPES[dtype], _lib.ptr(x), _lib.ptr(out), arr(w[0]),
            arr(b[0]), arr(w[1]), arr(b[1]), (ctypes.c_int * niter)(*dil),
            niter, batch, length, channels, k, mode, 1 / 3, ws.data_ptr(),
            size, _lib.stream())
        torch.cuda.synchronize()
        outs.append(None if rc else out)
    _lib.check(lib.pm_debug_force(0, 0))
    _lib.check(lib.pm_debug_skew(0))
    tag = (f'trial {trial} {dtype} C {channels} k {k} dil {dil} B {batch} '
           f'L {length} nseg {nseg} mode {mode}')
    if outs[0] is not None and outs[1] is None:
        # no other whole-Block tiling of this shape (C = 128 k 11, C = 256
        # k 7): one fused-pair launch per iteration instead
        src = x
        for n in range(niter):
            last = n + 1 == niter
            dst = prev.clone().to(device) if last else torch.empty_like(x)
            _lib.check(lib.pm_block_iteration_cl(
                _lib.DTYPES[dtype], _lib.ptr(src), _lib.ptr(dst),
                _lib.ptr(w[0][n]), _lib.ptr(b[0][n]), _lib.ptr(w[1][n]),
                _lib.ptr(b[1][n]), batch, length, channels, k, dil[n],
                mode if last else 0, 1 / 3 if last else 1., ws.data_ptr(),
                weights, _lib.stream()))
            src = dst
        torch.cuda.synchronize()
        outs[1] = src
        tag += ' (vs pair launches)'
    if outs[0] is None or outs[1] is None:
        skipped += 1
        print(tag + ': no kernel for this shape on one side, skipped')
        continue
    same = torch.equal(outs[0], outs[1])
    finite = bool(torch.isfinite(outs[0]).all())
    if not (same and finite):
        bad += 1
        diff = (outs[0] - outs[1]).abs()
        print('MISMATCH ' + tag + f': max diff {diff.max().item():.3e} at '
              f'{(diff.amax(-1) > 0).nonzero()[:4].tolist()} finite {finite}')
    else:
        print(tag + ': ok', flush=True)
print('fuzz_skew:', 'FAILED %d' % bad if bad else
      f'all {trials - skipped} compared trials exact ({skipped} skipped)')
sys.exit(1 if bad else 0)
