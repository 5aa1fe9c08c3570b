"""GPU parity, per kernel: HIP path (through the C ABI) vs the CPU oracle on
the same seeded inputs. fp32 operand mode is exact-fp32 MFMA (only the
summation order differs from ATen); 16-bit operands accumulate in fp32. The
gates are set from measurement (TOL* below)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import restatement as oracle
from util import check, from_cl, max_abs, pad32, rel_err, to_cl

pytestmark = pytest.mark.gpu

# Gates, relative to the output's abs-max: <= 3x what MI355X measures
# (profiles/r04/measured_errors.json, the ledger tests/util.py::check keeps) -
# one conv pair: fp32 3.2e-6 / f16 2.5e-4 / bf16 2.1e-3; a whole Block or MRF
# stage: 2.9e-6 / 4.3e-4 / 3.5e-3; the polyphase upsamplers (K = 2 C_in, one
# rounding of a wide sum): 1.8e-6 / 4.3e-4 / 3.6e-3. Split f16 ('f16x3': hi + lo,
# three MFMAs per step) measures like fp32: 2.4e-6 / 1.6e-6 / 1.6e-6; 'f16a2'
# (activations split, weights rounded once: two MFMAs) 1.8e-4 / 2.8e-4 / 2.7e-4 -
# on these random inputs the weight rounding it keeps is half of f16's error.
TOL = {'fp32': 8e-6, 'f16': 6e-4, 'bf16': 5e-3, 'f16x3': 8e-6, 'f16a2': 5e-4}
TOL_BLOCK = {'fp32': 8e-6, 'f16': 1.2e-3, 'bf16': 1e-2, 'f16x3': 5e-6,
             'f16a2': 8e-4}
TOL_MRF = {'fp32': 2.3e-6, 'f16': 7.5e-4, 'bf16': 6e-3, 'f16x3': 2e-6,
           'f16a2': 5e-4}
TOL_UP = {'fp32': 5e-6, 'f16': 1.2e-3, 'bf16': 1e-2, 'f16x3': 5e-6,
          'f16a2': 8e-4}


def lib():
    from promonet_amd import _lib
    return _lib


def workspace(device, c_in, c_out, k):
    size = lib().lib().pm_op_workspace_bytes(c_in, c_out, k)
    return torch.empty(size, dtype=torch.uint8, device=device)


def block_iteration_oracle(x, w1, b1, w2, b2, k, d):
    xt = F.leaky_relu(x, .1)
    xt = F.conv1d(xt, w1, b1, padding=oracle.get_padding(k, d), dilation=d)
    xt = F.leaky_relu(xt, .1)
    xt = F.conv1d(xt, w2, b2, padding=oracle.get_padding(k, 1))
    return xt + x


def run_block_iteration(
    device, dtype, x, w1, b1, w2, b2, k, d, mode=0, scale=1., out_init=None
):
    _lib = lib()
    b, c, l = x.shape
    x_cl = to_cl(x).to(device)
    out = torch.zeros_like(x_cl) if out_init is None \
        else to_cl(out_init).to(device)
    ws = workspace(device, c, c, k)
    tensors = [t.to(device).contiguous() for t in (w1, b1, w2, b2)]
    _lib.check(_lib.lib().pm_block_iteration_cl(
        _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
        *[_lib.ptr(t) for t in tensors], b, l, c, k, d, mode, scale,
        ws.data_ptr(), ws.numel(), _lib.stream()))
    torch.cuda.synchronize()
    padded = out[:, :, c:]
    assert padded.numel() == 0 or padded.abs().max().item() == 0.
    return from_cl(out, c).cpu()


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16', 'f16x3', 'f16a2'])
@pytest.mark.parametrize('channels', [32, 64, 128, 256])
@pytest.mark.parametrize('kernel_size', [3, 7, 11])
def test_block_iteration(device, dtype, channels, kernel_size):
    gen = torch.Generator().manual_seed(channels * 100 + kernel_size)
    for d, length in ((1, 301), (3, 257), (5, 130)):
        x = torch.randn(2, channels, length, generator=gen)
        std = 1. / (channels * kernel_size) ** .5
        w1 = torch.randn(channels, channels, kernel_size, generator=gen) * std
        w2 = torch.randn(channels, channels, kernel_size, generator=gen) * std
        b1 = torch.randn(channels, generator=gen) * .1
        b2 = torch.randn(channels, generator=gen) * .1
        want = block_iteration_oracle(x, w1, b1, w2, b2, kernel_size, d)
        got = run_block_iteration(
            device, dtype, x, w1, b1, w2, b2, kernel_size, d)
        check(rel_err(got, want), TOL[dtype], f'iteration:{dtype}',
              (channels, kernel_size, d, length))


@pytest.mark.parametrize('channels', [4, 8, 16, 48])
def test_block_iteration_padded_channels(device, channels):
    gen = torch.Generator().manual_seed(channels)
    x = torch.randn(1, channels, 97, generator=gen)
    w1 = torch.randn(channels, channels, 7, generator=gen) * .2
    w2 = torch.randn(channels, channels, 7, generator=gen) * .2
    b1 = torch.randn(channels, generator=gen)
    b2 = torch.randn(channels, generator=gen)
    want = block_iteration_oracle(x, w1, b1, w2, b2, 7, 3)
    got = run_block_iteration(device, 'fp32', x, w1, b1, w2, b2, 7, 3)
    check(rel_err(got, want), TOL['fp32'], 'iteration_padded:fp32', channels)


def test_block_iteration_short_and_modes(device):
    """Sequences shorter than the halo, one-sample sequences, and the MRF
    accumulate epilogues (hifigan.py:141-145)."""
    gen = torch.Generator().manual_seed(3)
    c, k = 32, 11
    w1 = torch.randn(c, c, k, generator=gen) * .05
    w2 = torch.randn(c, c, k, generator=gen) * .05
    b1 = torch.randn(c, generator=gen) * .1
    b2 = torch.randn(c, generator=gen) * .1
    for length in (1, 2, 9, 31, 118, 119, 236):
        x = torch.randn(3, c, length, generator=gen)
        want = block_iteration_oracle(x, w1, b1, w2, b2, k, 5)
        got = run_block_iteration(device, 'fp32', x, w1, b1, w2, b2, k, 5)
        check(rel_err(got, want), TOL['fp32'], 'iteration:fp32', length)
    x = torch.randn(2, c, 200, generator=gen)
    prev = torch.randn(2, c, 200, generator=gen)
    want = block_iteration_oracle(x, w1, b1, w2, b2, k, 1)
    got = run_block_iteration(
        device, 'fp32', x, w1, b1, w2, b2, k, 1, mode=1, scale=1 / 3)
    check(rel_err(got, want / 3), TOL['fp32'], 'iteration:fp32', 'mode 1')
    got = run_block_iteration(
        device, 'fp32', x, w1, b1, w2, b2, k, 1, mode=2, scale=1 / 3,
        out_init=prev)
    check(rel_err(got, prev + want / 3), TOL['fp32'], 'iteration:fp32',
          'mode 2')


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16', 'f16x3', 'f16a2'])
@pytest.mark.parametrize('channels', [32, 64, 16, 128])
@pytest.mark.parametrize('kernel_size', [3, 7, 11])
def test_whole_block(device, dtype, channels, kernel_size):
    """All three dilations of a Block fused in one kernel (trunk in
    registers, halo recomputed) vs the oracle's unfused op sequence; lengths
    around the tile size, shorter than the halo, and the MRF epilogues."""
    import ctypes
    _lib = lib()
    gen = torch.Generator().manual_seed(channels + kernel_size)
    std = 1. / (channels * kernel_size) ** .5
    weights = {
        name: [torch.randn(channels, channels, kernel_size, generator=gen) * std
               for _ in range(3)] for name in ('w1', 'w2')}
    biases = {
        name: [torch.randn(channels, generator=gen) * .1 for _ in range(3)]
        for name in ('b1', 'b2')}
    dilations = (1, 3, 5)
    state = {}
    for n in range(3):
        for which, (w, b) in enumerate((('w1', 'b1'), ('w2', 'b2')), 1):
            state[f'p.convs{which}.{n}.weight'] = weights[w][n]
            state[f'p.convs{which}.{n}.bias'] = biases[b][n]
    on_device = {
        name: [t.to(device).contiguous() for t in tensors]
        for name, tensors in {**weights, **biases}.items()}

    def pointers(name):
        return (ctypes.c_void_p * 3)(*[t.data_ptr() for t in on_device[name]])

    if channels == 128 and (kernel_size != 3 or dtype in ('fp32', 'f16x3', 'f16a2')):
        # (C = 128 k 7 exists as a WALKED whole Block only: covered by
        # test_walked_whole_block; k 11 and fp32 run on the pair kernel)
        with pytest.raises(RuntimeError, match='no whole-Block kernel'):
            x_cl = torch.zeros(1, 64, channels, device=device)
            _lib.check(_lib.lib().pm_block_cl(
                _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(x_cl),
                pointers('w1'), pointers('b1'), pointers('w2'), pointers('b2'),
                (ctypes.c_int * 3)(*dilations), 3, 1, 64, channels, kernel_size,
                0, 1., torch.empty(
                    3 * _lib.lib().pm_op_workspace_bytes(
                        channels, channels, kernel_size),
                    dtype=torch.uint8, device=device).data_ptr(),
                3 * _lib.lib().pm_op_workspace_bytes(
                    channels, channels, kernel_size), _lib.stream()))
        return
    dil = (ctypes.c_int * 3)(*dilations)
    ws = torch.empty(
        3 * _lib.lib().pm_op_workspace_bytes(channels, channels, kernel_size),
        dtype=torch.uint8, device=device)
    for length, mode in ((700, 0), (1, 0), (50, 1), (649, 2), (1300, 2)):
        x = torch.randn(2, channels, length, generator=gen)
        prev = torch.randn(2, channels, length, generator=gen)
        want = oracle.block(x, state, 'p', kernel_size, dilations)
        if mode == 1:
            want = want / 3
        elif mode == 2:
            want = prev + want / 3
        x_cl = to_cl(x).to(device)
        out = to_cl(prev).to(device)
        _lib.check(_lib.lib().pm_block_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out), pointers('w1'),
            pointers('b1'), pointers('w2'), pointers('b2'), dil, 3, 2, length,
            channels, kernel_size, mode, 1 / 3, ws.data_ptr(), ws.numel(),
            _lib.stream()))
        torch.cuda.synchronize()
        check(rel_err(from_cl(out, channels), want), TOL_BLOCK[dtype],
              f'block:{dtype}', (channels, kernel_size, length, mode))


def block_fixture(channels, kernel_size, gen, device, prefix='p'):
    std = 1. / (channels * kernel_size) ** .5
    state, tensors = {}, {'w1': [], 'b1': [], 'w2': [], 'b2': []}
    for n in range(3):
        for which in (1, 2):
            w = torch.randn(channels, channels, kernel_size, generator=gen) * std
            b = torch.randn(channels, generator=gen) * .1
            state[f'{prefix}.convs{which}.{n}.weight'] = w
            state[f'{prefix}.convs{which}.{n}.bias'] = b
            tensors[f'w{which}'].append(w.to(device).contiguous())
            tensors[f'b{which}'].append(b.to(device).contiguous())
    return state, tensors


@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
@pytest.mark.parametrize(
    'channels,kernel_size',
    [(64, 3), (64, 7), (64, 11), (128, 3), (128, 7), (256, 3)])
def test_walked_whole_block(device, dtype, channels, kernel_size):
    """The WALKED whole-Block kernels (conv_block3_walk_kernel: a workgroup
    walks its tiles left to right, every layer's last columns carried through
    LDS) at kernel level against the oracle - the production launcher only
    takes them for grids that fill the chip several times (16 % of the
    batch-32 x 10 s step runs in C = 128 k 7 and C = 256 k 3, which exist in
    no other form). Forced through pm_debug_force with 2 and 3 segments per
    utterance: >= 8 tiles per segment, uneven segments, an utterance end
    inside a tile, a segment boundary that is no tile multiple, mode 2."""
    _lib = lib()
    gen = torch.Generator().manual_seed(1000 + channels + kernel_size)
    dilations = (1, 3, 5)
    state, tensors = block_fixture(channels, kernel_size, gen, device)

    def pointers(name):
        return (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tensors[name]])

    dil = (ctypes.c_int * 3)(*dilations)
    size = 3 * _lib.lib().pm_op_workspace_bytes(channels, channels, kernel_size)
    ws = torch.empty(size, dtype=torch.uint8, device=device)
    # columns a walked tile advances by: NC - halo
    columns = {64: 256 if kernel_size == 3 else 512, 128: 256, 256: 128}[channels]
    halo = sum((d + 1) * (kernel_size // 2) for d in dilations)
    step = columns - halo
    try:
        for nseg, length, mode in (
                (2, 16 * step + 37, 0), (3, 25 * step - 5, 2), (2, 61, 1)):
            _lib.check(_lib.lib().pm_debug_force(nseg, 0))
            x = torch.randn(2, channels, length, generator=gen)
            prev = torch.randn(2, channels, length, generator=gen)
            want = oracle.block(x, state, 'p', kernel_size, dilations)
            want = {0: want, 1: want / 3, 2: prev + want / 3}[mode]
            x_cl = to_cl(x).to(device)
            out = to_cl(prev).to(device)
            _lib.check(_lib.lib().pm_block_cl(
                _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
                pointers('w1'), pointers('b1'), pointers('w2'), pointers('b2'),
                dil, 3, 2, length, channels, kernel_size, mode, 1 / 3,
                ws.data_ptr(), ws.numel(), _lib.stream()))
            torch.cuda.synchronize()
            error = rel_err(from_cl(out, channels), want)
            print(f'walked block C {channels} k {kernel_size} {dtype} nseg '
                  f'{nseg} L {length}: rel {error:.3e}')
            check(error, TOL_BLOCK[dtype], f'block:{dtype}',
                  ('walked', channels, kernel_size, nseg, length, mode))
            if channels <= 128 and (channels, kernel_size) != (128, 7):
                # bit-identical to the stand-alone (two-sided halo) tiling
                _lib.check(_lib.lib().pm_debug_force(0, 0))
                plain = to_cl(prev).to(device)
                _lib.check(_lib.lib().pm_block_cl(
                    _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(plain),
                    pointers('w1'), pointers('b1'), pointers('w2'),
                    pointers('b2'), dil, 3, 2, length, channels, kernel_size,
                    mode, 1 / 3, ws.data_ptr(), ws.numel(), _lib.stream()))
                torch.cuda.synchronize()
                assert torch.equal(plain, out), (nseg, length)
    finally:
        _lib.check(_lib.lib().pm_debug_force(0, 0))


SKEW_SHAPES = [(32, 3), (32, 7), (32, 11), (64, 3), (64, 7), (64, 11),
               (128, 3), (128, 7), (128, 11), (256, 3), (256, 7)]


# (the 4-byte operand layouts - exact fp32, split f16 - take the skewed walk
# where they have whole-Block tilings: C <= 64)
@pytest.mark.parametrize(
    'dtype,channels,kernel_size',
    [(dtype, c, k) for dtype in ('f16', 'bf16', 'f16x3', 'f16a2', 'fp32')
     for c, k in SKEW_SHAPES if dtype in ('f16', 'bf16') or c <= 64])
def test_skewed_whole_block(device, dtype, channels, kernel_size):
    """The SKEWED walk of a whole Block (conv_block3_skew_kernel: iteration i
    works 32 i columns behind iteration 0, the trunk moves one tile to the
    right in the register file between iterations - across waves and steps
    through scratch - and nothing is computed twice) against the oracle and,
    where another tiling of the same Block exists, bit for bit against it.
    Taken when the caller hands scratch over behind the workspace; forced here
    with 1, 2 and 3 segments per utterance: uneven segments, a segment shorter
    than a step, an utterance shorter than the skew, every store mode."""
    _lib = lib()
    gen = torch.Generator().manual_seed(2000 + channels + kernel_size)
    dilations = (1, 3, 5)
    state, tensors = block_fixture(channels, kernel_size, gen, device)

    def pointers(name):
        return (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tensors[name]])

    dil = (ctypes.c_int * 3)(*dilations)
    weights = 3 * _lib.lib().pm_op_workspace_bytes(
        channels, channels, kernel_size)
    scratch = _lib.lib().pm_walk_scratch_bytes(2)
    assert scratch > 0
    ws = torch.empty(weights + scratch, dtype=torch.uint8, device=device)
    columns = {32: 512, 64: 256 if kernel_size == 3 else 512, 128: 256,
               256: 128}[channels]
    if dtype in ('fp32', 'f16x3', 'f16a2') and channels == 64:
        columns = 256

    def run(x_cl, out, length, mode, size):
        _lib.check(_lib.lib().pm_block_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
            pointers('w1'), pointers('b1'), pointers('w2'), pointers('b2'),
            dil, 3, 2, length, channels, kernel_size, mode, 1 / 3,
            ws.data_ptr(), size, _lib.stream()))
        torch.cuda.synchronize()

    try:
        for nseg, length, mode in (
                (2, 9 * columns + 37, 0), (3, 13 * columns - 5, 2),
                (1, 3 * columns, 1), (2, 61, 1), (3, 2 * columns + 1, 2)):
            _lib.check(_lib.lib().pm_debug_force(nseg, 0))
            _lib.check(_lib.lib().pm_debug_skew(1))
            x = torch.randn(2, channels, length, generator=gen)
            prev = torch.randn(2, channels, length, generator=gen)
            want = oracle.block(x, state, 'p', kernel_size, dilations)
            want = {0: want, 1: want / 3, 2: prev + want / 3}[mode]
            x_cl = to_cl(x).to(device)
            out = to_cl(prev).to(device)
            ws[weights:].fill_(0xff)    # (NaN patterns: nothing stale is read)
            run(x_cl, out, length, mode, ws.numel())
            error = rel_err(from_cl(out, channels), want)
            print(f'skewed block C {channels} k {kernel_size} {dtype} nseg '
                  f'{nseg} L {length}: rel {error:.3e}')
            check(error, TOL_BLOCK[dtype], f'block:{dtype}',
                  ('skewed', channels, kernel_size, nseg, length, mode))
            if (channels, kernel_size) not in ((128, 11), (256, 7)):
                # the walked / stand-alone tiling of the same Block (no scratch
                # handed over): same arithmetic per column
                other = to_cl(prev).to(device)
                run(x_cl, other, length, mode, weights)
                assert torch.equal(other, out), (nseg, length)
    finally:
        _lib.check(_lib.lib().pm_debug_force(0, 0))
        _lib.check(_lib.lib().pm_debug_skew(0))


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16', 'f16x3', 'f16a2'])
@pytest.mark.parametrize(
    'c_in,c_out,rate',
    [(512, 256, 8), (256, 128, 8), (128, 64, 2), (64, 32, 2), (64, 32, 8),
     (32, 16, 2), (16, 8, 4)])
def test_conv_transpose(device, dtype, c_in, c_out, rate):
    # (the two wide r = 8 shapes run conv_upsample_kernel with 16-bit operands.
    # With inputs this small its launcher gives every 256-row M block its own
    # workgroup; the production path - ONE workgroup streaming the weights of
    # all 4 / 8 M blocks past a staged x tile, the next block's first
    # fragments prefetched - is reached through pm_debug_force(0, groups))
    _lib = lib()
    gen = torch.Generator().manual_seed(c_in + rate)
    k = 2 * rate
    wide = rate == 8 and c_in >= 256 and dtype in ('f16', 'bf16')
    cases = [(length, 0) for length in (1, 5, 130, 300)]
    if wide:
        cases += [(300, 1), (131, 2)]
    for length, groups in cases:
        _lib.check(_lib.lib().pm_debug_force(0, groups))
        x = torch.randn(2, c_in, length, generator=gen)
        w = torch.randn(c_in, c_out, k, generator=gen) / (c_in * 2) ** .5
        bias = torch.randn(c_out, generator=gen) * .1
        want = F.conv_transpose1d(
            F.leaky_relu(x, .1), w, bias, stride=rate,
            padding=(k - rate) // 2)
        x_cl = to_cl(x).to(device)
        out = torch.zeros(
            2, length * rate, pad32(c_out), device=device)
        ws = workspace(device, c_in, c_out, k)
        wd, bd = w.to(device), bias.to(device)
        _lib.check(_lib.lib().pm_conv_transpose_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out), _lib.ptr(wd),
            _lib.ptr(bd), 2, length, c_in, c_out, rate, 1, ws.data_ptr(),
            ws.numel(), _lib.stream()))
        torch.cuda.synchronize()
        _lib.check(_lib.lib().pm_debug_force(0, 0))
        check(rel_err(from_cl(out, c_out), want), TOL_UP[dtype],
              f'conv_transpose:{dtype}', (c_in, c_out, rate, length, groups))


def test_out_conv_tanh(device):
    _lib = lib()
    gen = torch.Generator().manual_seed(1)
    for c, length in ((32, 1000), (8, 77), (32, 1)):
        x = torch.randn(2, c, length, generator=gen)
        w = torch.randn(1, c, 7, generator=gen) * .2
        want = torch.tanh(F.conv1d(F.leaky_relu(x, .1), w, None, padding=3))
        x_cl, wd = to_cl(x).to(device), w.to(device)
        out = torch.empty(2, length, device=device)
        _lib.check(_lib.lib().pm_out_conv_tanh(
            _lib.ptr(x_cl), _lib.ptr(wd), _lib.ptr(out), 2, length, c,
            _lib.stream()))
        torch.cuda.synchronize()
        check(max_abs(out[:, None], want), 6e-6, 'out_conv_tanh:fp32', (c, length))


def test_fold_weight_norm(device):
    _lib = lib()
    gen = torch.Generator().manual_seed(2)
    v = torch.randn(96, 32, 11, generator=gen) * .01
    g = torch.rand(96, 1, 1, generator=gen) + .5
    want = oracle.fold_weight_norm(g, v)
    gd, vd = g.to(device), v.to(device)
    out = torch.empty_like(vd)
    _lib.check(_lib.lib().pm_fold_weight_norm(
        _lib.ptr(gd), _lib.ptr(vd), _lib.ptr(out), 96, 32 * 11,
        _lib.stream()))
    torch.cuda.synchronize()
    assert rel_err(out, want) < 1e-6


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'bf16', 'f16x3', 'f16a2'])
@pytest.mark.parametrize('channels', [32, 20])
def test_whole_mrf(device, dtype, channels):
    """The whole MRF ResidualBlock of the 32-channel stage in one launch
    (Blocks k = 3, 7, 11 back to back on one tile, their sum in registers)
    vs the oracle's residual_block (hifigan.py:141-145)."""
    import ctypes
    _lib = lib()
    gen = torch.Generator().manual_seed(channels)
    kernels, dilations = (3, 7, 11), (1, 3, 5)
    state, order = {}, {'w1': [], 'b1': [], 'w2': [], 'b2': []}
    for j, k in enumerate(kernels):
        std = 1. / (channels * k) ** .5
        for n in range(3):
            for which in (1, 2):
                w = torch.randn(channels, channels, k, generator=gen) * std
                b = torch.randn(channels, generator=gen) * .1
                state[f'p.model.{j}.convs{which}.{n}.weight'] = w
                state[f'p.model.{j}.convs{which}.{n}.bias'] = b
                order[f'w{which}'].append(w.to(device).contiguous())
                order[f'b{which}'].append(b.to(device).contiguous())

    def pointers(name):
        return (ctypes.c_void_p * 9)(*[t.data_ptr() for t in order[name]])

    dil = (ctypes.c_int * 3)(*dilations)
    ws = torch.empty(
        9 * _lib.lib().pm_op_workspace_bytes(channels, channels, 11),
        dtype=torch.uint8, device=device)
    for length in (900, 1, 61, 1500):
        x = torch.randn(2, channels, length, generator=gen)
        want = oracle.residual_block(x, state, 'p')
        x_cl = to_cl(x).to(device)
        out = torch.full_like(x_cl, 7.)
        _lib.check(_lib.lib().pm_mrf_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out), pointers('w1'),
            pointers('b1'), pointers('w2'), pointers('b2'), dil, 3, 2, length,
            channels, ws.data_ptr(), ws.numel(), _lib.stream()))
        torch.cuda.synchronize()
        got = from_cl(out, channels).cpu()
        check(rel_err(got, want), TOL_MRF[dtype], f'mrf:{dtype}',
              (channels, length))
    if dtype != 'fp32':
        # the walked whole-MRF kernel (conv_mrf_walk_kernel), which the
        # launcher only takes from 8 tiles per segment on: forced, 2 and 3
        # segments, ragged segment ends; bit-identical to the stand-alone tiling
        try:
            for nseg, length in ((2, 12000), (3, 9973), (2, 700)):
                x = torch.randn(2, channels, length, generator=gen)
                want = oracle.residual_block(x, state, 'p')
                x_cl = to_cl(x).to(device)
                outs = []
                for force in (nseg, 0):
                    _lib.check(_lib.lib().pm_debug_force(force, 0))
                    out = torch.full_like(x_cl, 7.)
                    _lib.check(_lib.lib().pm_mrf_cl(
                        _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
                        pointers('w1'), pointers('b1'), pointers('w2'),
                        pointers('b2'), dil, 3, 2, length, channels,
                        ws.data_ptr(), ws.numel(), _lib.stream()))
                    torch.cuda.synchronize()
                    outs.append(out)
                got = from_cl(outs[0], channels).cpu()
                check(rel_err(got, want), TOL_MRF[dtype], f'mrf:{dtype}',
                      ('walked', channels, nseg, length))
                assert torch.equal(outs[0], outs[1]), (nseg, length)
        finally:
            _lib.check(_lib.lib().pm_debug_force(0, 0))
    if dtype in ('fp32', 'f16x3', 'f16a2'):
        # the SKEWED whole-MRF walk of the 4-byte operand layouts
        # (conv_mrf_skew_kernel: three skewed Blocks per window, their sum in
        # registers), taken when scratch sits behind the weights; forced with
        # 1, 2 and 3 segments: uneven segments, a segment shorter than a step,
        # an utterance shorter than the skew. Against the oracle and against
        # the stand-alone two-sided tiling of the same launch without scratch
        # (same sum order (B11 + B7 + B3) / 3: bit for bit).
        scratch = _lib.lib().pm_walk_scratch_bytes(2)
        big = torch.empty(ws.numel() + scratch, dtype=torch.uint8, device=device)
        try:
            for nseg, length in ((1, 3000), (2, 9973), (3, 12000), (2, 700),
                                 (1, 40)):
                x = torch.randn(2, channels, length, generator=gen)
                want = oracle.residual_block(x, state, 'p')
                x_cl = to_cl(x).to(device)
                outs = []
                for force, buffer in ((nseg, big), (0, ws)):
                    _lib.check(_lib.lib().pm_debug_force(force, 0))
                    out = torch.full_like(x_cl, 7.)
                    _lib.check(_lib.lib().pm_mrf_cl(
                        _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
                        pointers('w1'), pointers('b1'), pointers('w2'),
                        pointers('b2'), dil, 3, 2, length, channels,
                        buffer.data_ptr(), buffer.numel(), _lib.stream()))
                    torch.cuda.synchronize()
                    outs.append(out)
                got = from_cl(outs[0], channels).cpu()
                check(rel_err(got, want), TOL_MRF[dtype], f'mrf:{dtype}',
                      ('skewed', channels, nseg, length))
                assert torch.equal(outs[0], outs[1]), (nseg, length)
                again = torch.full_like(x_cl, 7.)
                _lib.check(_lib.lib().pm_debug_force(nseg, 0))
                _lib.check(_lib.lib().pm_mrf_cl(
                    _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(again),
                    pointers('w1'), pointers('b1'), pointers('w2'),
                    pointers('b2'), dil, 3, 2, length, channels,
                    big.data_ptr(), big.numel(), _lib.stream()))
                torch.cuda.synchronize()
                assert torch.equal(again, outs[0])
        finally:
            _lib.check(_lib.lib().pm_debug_force(0, 0))
    with pytest.raises(RuntimeError):
        x_cl = torch.zeros(1, 8, 64, device=device)
        _lib.check(_lib.lib().pm_mrf_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(x_cl), pointers('w1'),
            pointers('b1'), pointers('w2'), pointers('b2'), dil, 3, 1, 8, 64,
            ws.data_ptr(), ws.numel(), _lib.stream()))


@pytest.mark.parametrize('dtype', ['fp32', 'f16', 'f16x3', 'f16a2'])
@pytest.mark.parametrize('shape', [(113, 512, 258), (113, 64, 258), (40, 32, 6)])
def test_input_conv(device, dtype, shape):
    """Input feature conv (k 7) + speaker conditioning conv (k 1) as a
    per-utterance bias (hifigan.py:19-30, 67-68), batch-1 globals broadcast."""
    _lib = lib()
    c_in, c_out, G = shape
    gen = torch.Generator().manual_seed(c_in + c_out)
    w = torch.randn(c_out, c_in, 7, generator=gen) / (c_in * 7) ** .5
    bias = torch.randn(c_out, generator=gen) * .1
    sw = torch.randn(c_out, G, 1, generator=gen) / G ** .5
    sb = torch.randn(c_out, generator=gen) * .1
    on_device = [t.to(device).contiguous() for t in (w, bias, sw, sb)]
    for batch, length, gbatch in ((2, 45, 2), (3, 130, 1), (1, 1, 1)):
        x = torch.randn(batch, c_in, length, generator=gen)
        g = torch.randn(gbatch, G, 1, generator=gen)
        want = F.conv1d(x, w, bias, padding=3) + F.conv1d(g, sw, sb)
        x_cl = to_cl(x).to(device)
        out = torch.zeros(batch, length, pad32(c_out), device=device)
        glob = g[:, :, 0].to(device).contiguous()
        size = _lib.lib().pm_op_workspace_bytes(c_in, c_out, 7) + \
            256 * ((batch * pad32(c_out) * 4 + 255) // 256)
        ws = torch.empty(size, dtype=torch.uint8, device=device)
        _lib.check(_lib.lib().pm_input_conv_cl(
            _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out),
            *[_lib.ptr(t) for t in on_device[:2]], _lib.ptr(glob),
            *[_lib.ptr(t) for t in on_device[2:]], gbatch, G, batch, length,
            c_in, c_out, ws.data_ptr(), ws.numel(), _lib.stream()))
        torch.cuda.synchronize()
        got = from_cl(out, c_out).cpu()
        check(rel_err(got, want), TOL[dtype], f'input_conv:{dtype}',
              (shape, batch, length))


def test_f16_operands_saturate(device):
    """Activations beyond the top of the f16 range (65504) - possible with a
    trained checkpoint, never with the random-init weights of the parity
    tests - are saturated when they are converted to MFMA operands, not
    turned into inf (LeakyReLU keeps the negative side ten times further
    from its limit):
    the Block output stays finite and equals the oracle fed the saturated
    activations. Also: realistically large activations (1e3) keep the
    relative error of the f16 path."""
    gen = torch.Generator().manual_seed(3)
    channels, k, d, length = 128, 7, 3, 200
    std = 1. / (channels * k) ** .5
    w1 = torch.randn(channels, channels, k, generator=gen) * std
    w2 = torch.randn(channels, channels, k, generator=gen) * std
    b1 = torch.randn(channels, generator=gen) * .1
    b2 = torch.randn(channels, generator=gen) * .1
    x = torch.randn(1, channels, length, generator=gen) * 1e3
    want = block_iteration_oracle(x, w1, b1, w2, b2, k, d)
    got = run_block_iteration(device, 'f16', x, w1, b1, w2, b2, k, d)
    assert torch.isfinite(got).all()
    check(rel_err(got, want), TOL['f16'], 'iteration:f16', 'large activations')
    x[0, 5, 50] = 3e5                       # lrelu(x) = 3e5 > 65504
    x[0, 9, 120] = -6e5                     # lrelu(x) = -6e4: still finite
    got = run_block_iteration(device, 'f16', x, w1, b1, w2, b2, k, d)
    assert torch.isfinite(got).all()

    def saturated_oracle():
        xt = torch.clamp(F.leaky_relu(x, .1), max=65504.)
        xt = F.conv1d(xt, w1, b1, padding=oracle.get_padding(k, d), dilation=d)
        xt = torch.clamp(F.leaky_relu(xt, .1), max=65504.)
        xt = F.conv1d(xt, w2, b2, padding=oracle.get_padding(k, 1))
        return xt + x
    check(rel_err(got, saturated_oracle()), TOL['f16'], 'iteration:f16',
          'saturated')


@pytest.mark.parametrize('act', ['f16', 'bf16'])
@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
@pytest.mark.parametrize('channels,kernel_size', [(128, 11), (64, 11), (64, 7)])
def test_block_output_as_next_operand(device, dtype, act, channels, kernel_size):
    """The 16-bit hand-over between a stage's last Block launch and the next
    upsampler (Block3Args::act16 / SingleArgs::x16, hifigan.py:100-106 stages
    lrelu(stage output)): the skewed walk writes cvt(lrelu(result)) in the NEXT
    stage's operand type instead of the fp32 tensor - bit for bit what rounding
    the fp32 result gives - and leaves `out` alone; the r = 2 upsampler fed with
    it equals, bit for bit, the one that stages the fp32 tensor itself. Outside
    the skewed walk the request is refused, not ignored."""
    _lib = lib()
    gen = torch.Generator().manual_seed(3000 + channels + kernel_size)
    dilations = (1, 3, 5)
    state, tensors = block_fixture(channels, kernel_size, gen, device)

    def pointers(name):
        return (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tensors[name]])

    dil = (ctypes.c_int * 3)(*dilations)
    weights = 3 * _lib.lib().pm_op_workspace_bytes(
        channels, channels, kernel_size)
    ws = torch.empty(
        weights + _lib.lib().pm_walk_scratch_bytes(2), dtype=torch.uint8,
        device=device)
    columns = 512 if channels == 64 else 256
    length = 7 * columns + 45
    x = torch.randn(2, channels, length, generator=gen)
    prev = torch.randn(2, channels, length, generator=gen)
    x_cl, prev_cl = to_cl(x).to(device), to_cl(prev).to(device)
    torch_type = {'f16': torch.float16, 'bf16': torch.bfloat16}[act]

    def block(out, act16=None, size=None):
        args = [pointers('w1'), pointers('b1'), pointers('w2'), pointers('b2'),
                dil, 3, 2, length, channels, kernel_size, 2, 1 / 3,
                ws.data_ptr(), size or ws.numel(), _lib.stream()]
        if act16 is None:
            code = _lib.lib().pm_block_cl(
                _lib.DTYPES[dtype], _lib.ptr(x_cl), _lib.ptr(out), *args)
        else:
            code = _lib.lib().pm_block_act16_cl(
                _lib.DTYPES[dtype], _lib.DTYPES[act], _lib.ptr(x_cl),
                _lib.ptr(out), act16.data_ptr(), *args)
        torch.cuda.synchronize()
        return code

    try:
        _lib.check(_lib.lib().pm_debug_force(2, 0))
        _lib.check(_lib.lib().pm_debug_skew(1))
        plain = prev_cl.clone()
        _lib.check(block(plain))
        out = prev_cl.clone()
        act16 = torch.full(
            (2, length, channels), 0x7fff, dtype=torch.int16, device=device)
        _lib.check(block(out, act16))
        assert torch.equal(out, prev_cl)            # read, not written
        want = F.leaky_relu(plain, .1)
        if act == 'f16':
            want = want.clamp(max=65504.)
        want = want.to(torch_type).view(torch.int16)
        assert torch.equal(act16, want)
        # the consumer: the next stage's r = 2 upsampler, operand type `act`
        c_out = channels // 2
        w = torch.randn(channels, c_out, 4, generator=gen) / (2 * channels) ** .5
        bias = torch.randn(c_out, generator=gen) * .1
        wd, bd = w.to(device), bias.to(device)
        up_ws = workspace(device, channels, c_out, 4)
        outs = []
        for source in ('fp32', 'x16'):
            up = torch.zeros(2, 2 * length, pad32(c_out), device=device)
            if source == 'fp32':
                _lib.check(_lib.lib().pm_conv_transpose_cl(
                    _lib.DTYPES[act], _lib.ptr(plain), _lib.ptr(up),
                    _lib.ptr(wd), _lib.ptr(bd), 2, length, channels, c_out, 2,
                    1, up_ws.data_ptr(), up_ws.numel(), _lib.stream()))
            else:
                _lib.check(_lib.lib().pm_conv_transpose_x16_cl(
                    _lib.DTYPES[act], act16.data_ptr(), _lib.ptr(up),
                    _lib.ptr(wd), _lib.ptr(bd), 2, length, channels, c_out, 2,
                    up_ws.data_ptr(), up_ws.numel(), _lib.stream()))
            torch.cuda.synchronize()
            outs.append(up)
        assert torch.equal(outs[0], outs[1])
        expected = F.conv_transpose1d(
            F.leaky_relu(from_cl(plain, channels).cpu(), .1), w, bias, stride=2,
            padding=1)
        check(rel_err(from_cl(outs[1], c_out), expected), TOL_UP[act],
              f'conv_transpose:{act}', ('x16', channels))
        # no scratch behind the workspace -> another kernel -> refused
        _lib.check(_lib.lib().pm_debug_skew(-1))
        other = prev_cl.clone()
        if (channels, kernel_size) == (128, 11):
            # (this Block exists as a skewed walk only)
            assert block(other, act16) == _lib.PM_EINVAL
        else:
            assert block(other, act16) == _lib.PM_ESTATE
            assert torch.equal(other, plain)        # the fp32 result instead
    finally:
        _lib.check(_lib.lib().pm_debug_force(0, 0))
        _lib.check(_lib.lib().pm_debug_skew(0))
