"""HiFi-GAN vocoder running on the MI355X HIP engine.

Drop-in for `promonet.model.HiFiGAN` (promonet/model/hifigan.py:13-77):
same constructor arguments, same `forward(x, g, p)` signature, same
`state_dict()` keys / shapes (weight-normed layers keep `weight_g` and
`weight_v`), so a reference `generator-*.pt` loads unchanged. The forward
pass is one call into `libpromonet_hip.so` (`pm_hifigan_forward`); there is
no PyTorch compute path and no CPU fallback.
"""
import ctypes
import math

import torch

import promonet_amd
from promonet_amd import _lib
from .core import attach


def checkpoint_schedule(stages):
    """The operand types 'checkpoint' (config.py) stands for, one per upsampling
    stage: f16 everywhere; in the LAST stage the activations split into hi + lo
    ('f16a2': its Blocks two MFMAs per step, its upsampler fully split), and
    the upsampler in front of the second-to-last stage fully split as well
    ('f16ux'). scripts/checkpoint_schedule.py measures the candidates."""
    if stages == 1:
        return ['f16a2']
    return ['f16'] * (stages - 2) + ['f16ux', 'f16a2']

class HiFiGAN(torch.nn.Module):

    def __init__(self, initial_channel, gin_channels):
        super().__init__()
        self.initial_channel = initial_channel
        self.gin_channels = gin_channels
        self.channels = promonet_amd.HIFIGAN_UPSAMPLE_INITIAL_SIZE
        self.rates = list(promonet_amd.HIFIGAN_UPSAMPLE_RATES)
        self.kernels = list(promonet_amd.HIFIGAN_UPSAMPLE_KERNEL_SIZES)
        self.res_kernels = list(promonet_amd.HIFIGAN_RESBLOCK_KERNEL_SIZES)
        self.res_dilations = [
            list(d) for d in promonet_amd.HIFIGAN_RESBLOCK_DILATION_SIZES]
        self.compute_dtype = promonet_amd.COMPUTE_DTYPE
        self.hopsize = math.prod(self.rates)

        for key, tensor in self._initial_state().items():
            attach(self, key, tensor)

        self._engine = None
        self._engine_key = None
        self._generation = 0     # bumped whenever the engine is dropped
        self._workspace = None
        self._busy = None        # (stream, event) of the last forward
        self.register_load_state_dict_post_hook(
            lambda module, keys: module._invalidate())

    ###########################################################################
    # Parameters (reference init: hifigan.py:19-61, 220-223)
    ###########################################################################

    def _initial_state(self):
        state = {}

        def default_conv(prefix, cout, cin, k, bias=True):
            # torch.nn.Conv1d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            bound = 1. / math.sqrt(cin * k)
            state[f'{prefix}.weight'] = torch.empty(
                cout, cin, k).uniform_(-bound, bound)
            if bias:
                state[f'{prefix}.bias'] = torch.empty(
                    cout).uniform_(-bound, bound)

        def weight_normed(prefix, shape, nbias, fan_in):
            v = torch.randn(*shape) * .01
            bound = 1. / math.sqrt(fan_in)
            state[f'{prefix}.bias'] = torch.empty(nbias).uniform_(-bound, bound)
            state[f'{prefix}.weight_g'] = torch.linalg.vector_norm(
                v, dim=(1, 2), keepdim=True)
            state[f'{prefix}.weight_v'] = v

        c0 = self.channels
        default_conv('input_feature_conv', c0, self.initial_channel, 7)
        default_conv('input_speaker_conv', c0, self.gin_channels, 1)
        for i, (r, k) in enumerate(zip(self.rates, self.kernels)):
            cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
            weight_normed(
                f'model.{i}.model.1', (cin, cout, k), cout, cout * k)
            for j, ks in enumerate(self.res_kernels):
                for name in ('convs1', 'convs2'):
                    for n in range(len(self.res_dilations[j])):
                        weight_normed(
                            f'model.{i}.model.2.model.{j}.{name}.{n}',
                            (cout, cout, ks), cout, cout * ks)
        n = len(self.rates)
        default_conv(f'model.{n + 1}', 1, c0 // 2 ** n, 7, bias=False)
        return state

    ###########################################################################
    # Engine lifetime
    ###########################################################################

    def _invalidate(self):
        self._destroy()

    def _destroy(self):
        if getattr(self, '_engine', None) is not None:
            _lib.lib().pm_hifigan_destroy(self._engine)
            # (captured graphs hold the old engine's weight pointers:
            # Generator.packed_inference(graph=True) keys its cache on this)
            self._generation = getattr(self, '_generation', 0) + 1
        self._engine = None
        self._engine_key = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .half() move the parameters: repack lazily
        result = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return result

    def engine(self):
        """Create the HIP engine and (re)load every tensor when needed."""
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError(
                'promonet_amd.model.HiFiGAN runs on an AMD GPU only; move the '
                'model with .to("cuda:N") (no CPU fallback)')
        key = (first.device, self.compute_dtype)
        if self._engine is not None and self._engine_key == key:
            return self._engine
        self._destroy()
        lib = _lib.lib()
        config = _lib.HifiganConfig()
        config.num_features = self.initial_channel
        config.global_channels = self.gin_channels
        config.initial_channels = self.channels
        config.num_stages = len(self.rates)
        for i, (r, k) in enumerate(zip(self.rates, self.kernels)):
            config.upsample_rates[i] = r
            config.upsample_kernel_sizes[i] = k
        config.num_resblocks = len(self.res_kernels)
        config.num_dilations = len(self.res_dilations[0])
        for j, k in enumerate(self.res_kernels):
            config.resblock_kernel_sizes[j] = k
            if len(self.res_dilations[j]) != config.num_dilations:
                raise ValueError('ragged dilation lists are not supported')
            for n, d in enumerate(self.res_dilations[j]):
                config.resblock_dilations[j][n] = d
        # 'checkpoint' / 'bf16' / 'f16' / 'fp32' / 'f16x3', or one operand type
        # per upsampling stage joined by '+' ('bf16+bf16+f16+f16': the input
        # conv takes the first)
        per_stage = str(self.compute_dtype).split('+')
        if per_stage == ['checkpoint']:
            per_stage = checkpoint_schedule(len(self.rates))
        if len(per_stage) not in (1, len(self.rates)):
            raise ValueError(
                f'COMPUTE_DTYPE {self.compute_dtype!r}: one operand type, or '
                f'one per stage ({len(self.rates)})')
        config.compute_dtype = _lib.DTYPES[per_stage[0]]
        if len(per_stage) > 1:
            for i, name in enumerate(per_stage):
                config.stage_compute_dtype[i] = 1 + _lib.DTYPES[name]
        handle = ctypes.c_void_p()
        with torch.cuda.device(first.device):
            _lib.check(lib.pm_hifigan_create(
                ctypes.byref(config), ctypes.byref(handle)))
            try:
                for name, tensor in self.state_dict().items():
                    tensor = tensor.detach().to(torch.float32).contiguous()
                    _lib.check(lib.pm_hifigan_load_tensor(
                        handle, name.encode(), _lib.ptr(tensor),
                        _lib.shape_array(tensor.shape), tensor.ndim,
                        _lib.stream()))
                _lib.check(lib.pm_hifigan_finalize(handle, _lib.stream()))
            except Exception:
                lib.pm_hifigan_destroy(handle)
                raise
        self._engine = handle
        self._engine_key = key
        return handle

    def workspace(self, batch, frames, device):
        size = _lib.lib().pm_hifigan_workspace_bytes(
            self.engine(), batch, frames)
        if (
            self._workspace is None or self._workspace.numel() < size or
            self._workspace.device != device
        ):
            self._workspace = None
            self._workspace = torch.empty(
                size, dtype=torch.uint8, device=device)
        return self._workspace

    def _claim_workspace(self, device):
        """One workspace per module: a forward on stream B while the previous
        one is still running on stream A would overwrite the activations under
        it. Raise instead (same-stream calls are ordered by the stream; a
        different stream is fine once the previous forward has finished)."""
        stream = torch.cuda.current_stream(device)
        if torch.cuda.is_current_stream_capturing():
            return None
        if self._busy is not None:
            previous, event = self._busy
            if previous != stream and not event.query():
                raise RuntimeError(
                    'promonet_amd.model.HiFiGAN: forward on '
                    f'{stream} while the previous forward is still running on '
                    f'{previous} - the module owns ONE workspace; wait for it '
                    '(stream.wait_stream / synchronize) or use one model per '
                    'stream')
            event = event if previous.device == stream.device \
                else torch.cuda.Event()
        else:
            event = torch.cuda.Event()
        return stream, event

    def _release_workspace(self, claim):
        if claim is not None:
            claim[1].record(claim[0])
            self._busy = claim

    ###########################################################################
    # Forward
    ###########################################################################

    def forward(self, x, g, p=None):
        """x (B, C, T) features, g (B|1, G, 1) globals -> (B, 1, T * hop).

        `p` (previous samples) is accepted and ignored, as in the reference
        (promonet/model/hifigan.py:63)."""
        return self._run(x, g, channels_last=False)

    def forward_channels_last(self, x_cl, g, lengths=None):
        """x_cl (B, T, C_pad) as written by `pm_prepare_features`; `lengths`
        (B,) frames makes the batch ragged: every utterance is synthesised
        as if alone, the audio past its end is zero."""
        return self._run(x_cl, g, channels_last=True, lengths=lengths)

    def _run(self, x, g, channels_last, lengths=None):
        _lib.require_gpu(x)
        engine = self.engine()
        lib = _lib.lib()
        x = x.to(torch.float32).contiguous()
        batch = x.shape[0]
        frames = x.shape[1] if channels_last else x.shape[2]
        g = g.reshape(g.shape[0], -1).to(torch.float32).contiguous()
        if g.shape[1] != self.gin_channels:
            raise ValueError(
                f'expected {self.gin_channels} global channels, '
                f'got {g.shape[1]}')
        if g.shape[0] not in (1, batch):
            raise ValueError('global features must have batch 1 or B')
        if not channels_last and x.shape[1] != self.initial_channel:
            raise ValueError(
                f'expected {self.initial_channel} feature channels, '
                f'got {x.shape[1]}')
        out = torch.empty(
            batch, 1, frames * self.hopsize, dtype=torch.float32,
            device=x.device)
        if lengths is not None:
            lengths = torch.as_tensor(lengths).to(
                device=x.device, dtype=torch.int32).contiguous()
            if lengths.shape != (batch,):
                raise ValueError('lengths must have shape (B,)')
        with torch.cuda.device(x.device):
            claim = self._claim_workspace(x.device)
            workspace = self.workspace(batch, frames, x.device)
            try:
                if lengths is not None:
                    _lib.check(lib.pm_hifigan_forward_ragged(
                        engine, _lib.ptr(x), int(channels_last), _lib.ptr(g),
                        g.shape[0], _lib.ptr(lengths, torch.int32),
                        _lib.ptr(out), batch, frames, workspace.data_ptr(),
                        workspace.numel(), _lib.stream()))
                else:
                    fn = lib.pm_hifigan_forward_cl if channels_last \
                        else lib.pm_hifigan_forward
                    _lib.check(fn(
                        engine, _lib.ptr(x), _lib.ptr(g), g.shape[0],
                        _lib.ptr(out), batch, frames, workspace.data_ptr(),
                        workspace.numel(), _lib.stream()))
            finally:
                self._release_workspace(claim)
        return out

    def remove_weight_norm(self):
        """No-op: the engine folds weight norm once at load
        (reference: hifigan.py:72-77 only matters for TorchScript export)."""
