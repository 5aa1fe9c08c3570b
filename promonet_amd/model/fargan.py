"""FARGAN vocoder running on the MI355X HIP engine.

Drop-in for `promonet.model.FARGAN` (promonet/model/fargan.py:13-131,
selected by `config/fargan.py`): same constructor arguments, same
`forward(features, global_features, previous_samples)`, same `state_dict()`
keys (weight-normed Linear layers keep `weight_g` / `weight_v`). The whole
frame-autoregressive loop - 4 sub-frame steps per frame, the pitch lookback
gather, three GRU cells, GLUs - runs inside one persistent kernel per
utterance (`pm_fargan_forward`); there is no PyTorch compute path.
"""
import ctypes
import math

import torch

import promonet_amd
from promonet_amd import _lib
from .core import attach


class FARGAN(torch.nn.Module):

    # auto mode: calls served by the one-workgroup-per-utterance kernel after
    # the cluster exchange has timed out twice in a row, before the clusters
    # are tried again
    RETRY_AFTER = 16

    def __init__(self, num_features, global_channels):
        super().__init__()
        self.num_features = num_features
        self.global_channels = global_channels
        self.hopsize = promonet_amd.HOPSIZE
        self.weight_dtype = promonet_amd.FARGAN_WEIGHT_DTYPE
        self.check_exchange = True
        self.kernel_mode = 0    # 0 auto, 1 workgroup per utterance, 2 clusters
        self._fallback_calls = 0    # auto mode: calls left on the slow kernel
        for key, tensor in self._initial_state().items():
            attach(self, key, tensor)
        self._engine = None
        self._engine_key = None
        self._workspace = None
        self.register_load_state_dict_post_hook(
            lambda module, keys: module._destroy())

    def _initial_state(self):
        """Reference init: orthogonal Linear weights (fargan.py:418-424),
        GRUCell default U(-1/sqrt(H), 1/sqrt(H)), weight-norm g = ||v||."""
        hop, sub = self.hopsize, self.hopsize // 4
        channels = self.num_features + self.global_channels
        state = {}

        def orthogonal(rows, cols):
            return torch.nn.init.orthogonal_(torch.empty(rows, cols))

        def normed(prefix, rows, cols):
            v = orthogonal(rows, cols)
            state[prefix + '.weight_g'] = torch.linalg.vector_norm(
                v, dim=1, keepdim=True)
            state[prefix + '.weight_v'] = v

        state['conditioning_network.0.weight'] = orthogonal(channels, channels)
        state['conditioning_network.2.weight'] = orthogonal(channels, channels)
        state['conditioning_network.4.weight'] = orthogonal(2 * hop, channels)
        p = 'subframe_network.'
        normed(p + 'framewise_convolution.model.0', hop, 2 * (4 * sub + 4))
        normed(p + 'framewise_convolution.model.2.gate', hop, hop)
        bound = 1. / math.sqrt(hop)
        for n in (1, 2, 3):
            state[p + f'gru{n}.weight_ih'] = torch.empty(
                3 * hop, hop + 2 * sub).uniform_(-bound, bound)
            state[p + f'gru{n}.weight_hh'] = torch.empty(
                3 * hop, hop).uniform_(-bound, bound)
        for name in ('gru1_glu', 'gru2_glu', 'gru3_glu', 'skip_glu'):
            normed(p + name + '.gate', hop, hop)
        state[p + 'skip_dense.weight'] = orthogonal(hop, 4 * hop + 2 * sub)
        state[p + 'output_layer.weight'] = orthogonal(sub, hop)
        return state

    ###########################################################################
    # Engine lifetime
    ###########################################################################

    def _invalidate(self):
        self._destroy()

    def _destroy(self):
        if getattr(self, '_engine', None) is not None:
            _lib.lib().pm_fargan_destroy(self._engine)
        self._engine = None
        self._engine_key = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _apply(self, fn, *args, **kwargs):
        result = super()._apply(fn, *args, **kwargs)
        self._destroy()
        return result

    def engine(self):
        first = next(self.parameters())
        if not first.is_cuda:
            raise RuntimeError(
                'promonet_amd.model.FARGAN runs on an AMD GPU only; move the '
                'model with .to("cuda:N") (no CPU fallback)')
        key = (first.device, self.weight_dtype)
        if self._engine is not None and self._engine_key == key:
            return self._engine
        self._destroy()
        lib = _lib.lib()
        handle = ctypes.c_void_p()
        with torch.cuda.device(first.device):
            _lib.check(lib.pm_fargan_create(
                self.num_features, self.global_channels,
                _lib.DTYPES[self.weight_dtype], ctypes.byref(handle)))
            try:
                for name, tensor in self.state_dict().items():
                    tensor = tensor.detach().to(torch.float32).contiguous()
                    _lib.check(lib.pm_fargan_load_tensor(
                        handle, name.encode(), _lib.ptr(tensor),
                        _lib.shape_array(tensor.shape), tensor.ndim,
                        _lib.stream()))
                _lib.check(lib.pm_fargan_finalize(handle, _lib.stream()))
            except Exception:
                lib.pm_fargan_destroy(handle)
                raise
        self._engine = handle
        self._engine_key = key
        return handle

    ###########################################################################
    # Forward (fargan.py:21-59)
    ###########################################################################

    def forward(self, features, global_features, previous_samples):
        """features (B, 114, T) with the pitch period in samples as the last
        channel, global_features (B|1, 258, 1), previous_samples
        (B|1, 1, 512) -> (B, 1, 256 T). Unlike the reference (whose gather
        needs previous_samples expanded to the batch, fargan.py:238-241),
        batch-1 globals / previous samples broadcast."""
        return self._run(features, global_features, previous_samples, False)

    def forward_channels_last(
        self, features_cl, global_features, previous_samples, lengths=None
    ):
        """`lengths` (B,) frames: ragged batch of zero-padded utterances, each
        synthesised exactly as if alone (the model is causal); tails are 0."""
        return self._run(
            features_cl, global_features, previous_samples, True, lengths)

    def _run(self, x, g, previous, channels_last, lengths=None):
        _lib.require_gpu(x)
        engine = self.engine()
        lib = _lib.lib()
        x = x.to(torch.float32).contiguous()
        batch = x.shape[0]
        frames = x.shape[1] if channels_last else x.shape[2]
        if not channels_last and x.shape[1] != self.num_features + 1:
            raise ValueError(
                f'expected {self.num_features + 1} feature channels (the last '
                f'one is the pitch period), got {x.shape[1]}')
        g = g.reshape(g.shape[0], -1).to(torch.float32).contiguous()
        if g.shape[1] != self.global_channels or g.shape[0] not in (1, batch):
            raise ValueError('global features must be (B|1, 258[, 1])')
        pointer, pbatch = None, 1
        if previous is not None:
            previous = previous.reshape(previous.shape[0], -1).to(
                device=x.device, dtype=torch.float32).contiguous()
            if previous.shape[1] != 2 * self.hopsize or \
                    previous.shape[0] not in (1, batch):
                raise ValueError('previous_samples must be (B|1, 1, 512)')
            pointer, pbatch = _lib.ptr(previous), previous.shape[0]
        out = torch.empty(
            batch, 1, frames * self.hopsize, dtype=torch.float32,
            device=x.device)
        with torch.cuda.device(x.device):
            size = lib.pm_fargan_workspace_bytes(engine, batch, frames)
            if self._workspace is None or self._workspace.numel() < size or \
                    self._workspace.device != x.device:
                self._workspace = torch.empty(
                    size, dtype=torch.uint8, device=x.device)

            if lengths is not None:
                lengths = torch.as_tensor(lengths).to(
                    device=x.device, dtype=torch.int32).contiguous()
                if lengths.shape != (batch,):
                    raise ValueError('lengths must have shape (B,)')

            if self.kernel_mode == 0 and not self.check_exchange:
                # the clusters' bounded waits are the only safety net of auto
                # mode: without the check a timed-out exchange would pass as
                # audio
                raise RuntimeError(
                    'FARGAN: check_exchange=False needs an explicit '
                    'kernel_mode (1 or 2)')

            def launch(mode):
                _lib.check(lib.pm_fargan_set_mode(engine, mode))
                if lengths is None:
                    _lib.check(lib.pm_fargan_forward(
                        engine, _lib.ptr(x), int(channels_last), _lib.ptr(g),
                        g.shape[0], pointer, pbatch, _lib.ptr(out), batch,
                        frames, self._workspace.data_ptr(),
                        self._workspace.numel(), _lib.stream()))
                else:
                    _lib.check(lib.pm_fargan_forward_ragged(
                        engine, _lib.ptr(x), int(channels_last), _lib.ptr(g),
                        g.shape[0], pointer, pbatch,
                        _lib.ptr(lengths, torch.int32), _lib.ptr(out), batch,
                        frames, self._workspace.data_ptr(),
                        self._workspace.numel(), _lib.stream()))
                if self.check_exchange:
                    # the cluster kernel's inter-workgroup waits are bounded;
                    # a tripped bound must not pass as audio (one stream sync)
                    _lib.check(lib.pm_fargan_check(
                        engine, batch, frames, self._workspace.data_ptr(),
                        _lib.stream()))
            if self.kernel_mode == 0 and self._fallback_calls > 0:
                # the mode is chosen BEFORE the counter moves (the call that
                # takes it to 0 still runs the fallback kernel; the next one
                # tries the clusters again inside the guarded block below)
                launch(1)
                self._fallback_calls -= 1
                return out
            try:
                launch(self.kernel_mode)
            except _lib.LibraryError as error:
                # The cluster kernel needs all its workgroups resident at
                # once; a GPU shared with another process (or CU-masked) can
                # break that and the bounded exchange gives up (PM_ETIMEOUT,
                # the audio of that launch is invalid). Auto mode retries the
                # clusters once - co-tenancy is often transient - and only
                # then runs this call on the one-workgroup-per-utterance
                # kernel; the clusters are tried again after RETRY_AFTER calls.
                if self.kernel_mode != 0 or error.code != _lib.PM_ETIMEOUT:
                    raise
                try:
                    launch(0)
                except _lib.LibraryError as again:
                    if again.code != _lib.PM_ETIMEOUT:
                        raise
                    import warnings
                    warnings.warn(
                        'FARGAN cluster exchange timed out twice (GPU shared '
                        'or partially masked?); using the one-workgroup-per-'
                        f'utterance kernel for the next {self.RETRY_AFTER} '
                        'calls')
                    self._fallback_calls = self.RETRY_AFTER
                    launch(1)
        return out

    def remove_weight_norm(self):
        """No-op: weight norm is folded once at load."""
