"""Where does the forced walked whole-MRF launch differ from the stand-alone tiling?"""
import ctypes, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
from util import to_cl, from_cl
from promonet_amd import _lib
device = 'cuda'
channels = 32
gen = torch.Generator().manual_seed(channels)
order = {'w1': [], 'b1': [], 'w2': [], 'b2': []}
for j, k in enumerate((3, 7, 11)):
    std = 1. / (channels * k) ** .5
    for n in range(3):
        for which in (1, 2):
            w = torch.randn(channels, channels, k, generator=gen) * std
            b = torch.randn(channels, generator=gen) * .1
            order[f'w{which}'].append(w.to(device).contiguous())
            order[f'b{which}'].append(b.to(device).contiguous())
def pointers(name):
    return (ctypes.c_void_p * 9)(*[t.data_ptr() for t in order[name]])
dil = (ctypes.c_int * 3)(1, 3, 5)
ws = torch.empty(9 * _lib.lib().pm_op_workspace_bytes(channels, channels, 11), dtype=torch.uint8, device=device)
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
for nseg, length in ((2, 12000), (3, 9973), (2, 700), (1, 5000)):
    x = torch.randn(2, channels, length, generator=gen)
    x_cl = to_cl(x).to(device)
    outs = []
    for force in (nseg, 0):
        _lib.check(_lib.lib().pm_debug_force(force, 0))
        out = torch.full_like(x_cl, 7.)
        _lib.check(_lib.lib().pm_mrf_cl(_lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out), pointers('w1'), pointers('b1'), pointers('w2'), pointers('b2'), dil, 3, 2, length, channels, ws.data_ptr(), ws.numel(), _lib.stream()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    _lib.check(_lib.lib().pm_debug_force(0, 0))
    d = (outs[0] - outs[1]).abs().amax(dim=2)   # (B, L)
    print(f'nseg {nseg} length {length}: max diff {d.max().item():.3e}')
    for b in range(2):
        bad = (d[b] > 0).nonzero().flatten()
        if len(bad):
            # runs
            runs = []; start = prev = bad[0].item()
            for t in bad[1:].tolist():
                if t != prev + 1: runs.append((start, prev)); start = t
                prev = t
            runs.append((start, prev))
            print('  b', b, 'differing column runs:', runs[:40], '...' if len(runs) > 40 else '')
