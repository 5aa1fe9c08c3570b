"""Generator = feature preparation + vocoder, on the HIP engine.

Drop-in for `promonet.model.Generator` (promonet/model/generator.py:84-197):
no-argument constructor from the import-time config, the 8-argument
`forward`, `prepare_features` / `prepare_global_features`, the same
`state_dict()` keys and buffers.
"""
import numpy as np
import torch

import promonet_amd
from promonet_amd import _lib
from .fargan import FARGAN
from .hifigan import HiFiGAN


class Generator(torch.nn.Module):

    def __init__(self):
        super().__init__()
        # Model selection (generator.py:18-31)
        self.fargan = promonet_amd.MODEL == 'fargan'
        if promonet_amd.MODEL == 'fargan':
            self.model = FARGAN(
                promonet_amd.NUM_FEATURES, promonet_amd.GLOBAL_CHANNELS)
        elif promonet_amd.MODEL == 'hifigan':
            self.model = HiFiGAN(
                promonet_amd.NUM_FEATURES, promonet_amd.GLOBAL_CHANNELS)
        else:
            raise ValueError(
                f'Generator model {promonet_amd.MODEL} is not defined')
        # generator.py:35-42: ZERO_SHOT swaps the speaker table for a Linear
        # over WavLM x-vectors; torch default inits (Embedding N(0, 1))
        self.zero_shot = bool(promonet_amd.ZERO_SHOT)
        self.augment_pitch = bool(promonet_amd.AUGMENT_PITCH)
        self.augment_loudness = bool(promonet_amd.AUGMENT_LOUDNESS)
        self.sparse_method = promonet_amd.SPARSE_PPG_METHOD
        if self.sparse_method not in _lib.SPARSE_METHODS:
            raise ValueError(
                f'Sparsify method {self.sparse_method} is not defined')
        if self.zero_shot:
            self.speaker_embedding = torch.nn.Linear(
                promonet_amd.WAVLM_EMBEDDING_CHANNELS,
                promonet_amd.SPEAKER_CHANNELS)
        else:
            self.speaker_embedding = torch.nn.Embedding(
                promonet_amd.NUM_SPEAKERS, promonet_amd.SPEAKER_CHANNELS)
        self.pitch_embedding = torch.nn.Embedding(
            promonet_amd.PITCH_BINS, promonet_amd.PITCH_EMBEDDING_SIZE)
        for parameter in self.parameters():
            parameter.requires_grad_(False)

        # generator.py:45-47, 101-104, 112-114
        self.register_buffer(
            'default_previous_samples',
            torch.zeros(1, 1, promonet_amd.NUM_PREVIOUS_SAMPLES))
        if self.sparse_method is not None:               # generator.py:97-104
            self.register_buffer(
                'ppg_threshold',
                torch.tensor(
                    promonet_amd.SPARSE_PPG_THRESHOLD, dtype=torch.float))
        self.register_buffer(
            'pitch_distribution', promonet_amd.load.pitch_distribution())

        # host copy of the threshold: reading the buffer every forward would
        # be a device sync (and is illegal inside a graph capture)
        self._threshold = None
        self.register_load_state_dict_post_hook(
            lambda module, keys: setattr(module, '_threshold', None))

    ###########################################################################
    # Forward (generator.py:116-135)
    ###########################################################################

    def forward(
        self,
        loudness,
        pitch,
        periodicity,
        ppg,
        speakers,
        spectral_balance_ratios,
        loudness_ratios,
        previous_samples=None,
        lengths=None
    ):
        """`lengths` (B,) frames (not in the reference): ragged batch of
        zero-padded utterances, each synthesised exactly as if alone."""
        features_cl = self._features(
            loudness, pitch, periodicity, ppg, channels_last=True)
        global_features = self.prepare_global_features(
            speakers, spectral_balance_ratios, loudness_ratios)
        if self.fargan:
            if previous_samples is None:
                previous_samples = self.default_previous_samples
            return self.model.forward_channels_last(
                features_cl, global_features, previous_samples, lengths)
        return self.model.forward_channels_last(
            features_cl, global_features, lengths)

    def prepare_features(self, loudness, pitch, periodicity, ppg):
        """(B, 113, T) conditioning tensor (generator.py:137-197)."""
        return self._features(
            loudness, pitch, periodicity, ppg, channels_last=False)

    def _features(self, loudness, pitch, periodicity, ppg, channels_last):
        lib = _lib.lib()
        _lib.require_gpu(pitch)
        if loudness.ndim == 2:
            loudness = loudness[None]
        loudness = loudness.to(torch.float32).contiguous()
        pitch = pitch.to(torch.float32).contiguous()
        periodicity = periodicity.to(torch.float32).contiguous()
        ppg = ppg.to(torch.float32).contiguous()
        batch, rows, frames = loudness.shape
        channels = ppg.shape[1]
        if pitch.shape != (batch, frames) or \
                periodicity.shape != (batch, frames) or \
                ppg.shape != (batch, channels, frames):
            raise ValueError('feature shapes disagree')
        total = (
            channels + promonet_amd.PITCH_EMBEDDING_SIZE +
            promonet_amd.LOUDNESS_BANDS + 1 + int(self.fargan))
        device = pitch.device
        edges = self.pitch_distribution.to(torch.float32).contiguous()
        table = self.pitch_embedding.weight.detach().to(
            torch.float32).contiguous()
        if channels_last:
            cpad = (total + 31) // 32 * 32
            out = torch.empty(batch, frames, cpad, device=device)
            out_ref, out_cl = None, _lib.ptr(out)
        else:
            cpad = 0
            out = torch.empty(batch, total, frames, device=device)
            out_ref, out_cl = _lib.ptr(out), None
        with torch.cuda.device(device):
            _lib.check(lib.pm_prepare_features(
                _lib.ptr(loudness), _lib.ptr(pitch), _lib.ptr(periodicity),
                _lib.ptr(ppg), _lib.ptr(edges), _lib.ptr(table), out_ref,
                out_cl, batch, frames, rows, channels,
                promonet_amd.PITCH_BINS, promonet_amd.PITCH_EMBEDDING_SIZE,
                promonet_amd.LOUDNESS_BANDS, cpad,
                _lib.SPARSE_METHODS[self.sparse_method],
                self._host_threshold(),
                promonet_amd.FMIN, promonet_amd.FMAX, promonet_amd.MIN_DB,
                promonet_amd.REF_DB,
                float(promonet_amd.SAMPLE_RATE) if self.fargan else 0.,
                _lib.stream()))
        return out

    def _host_threshold(self):
        if self.sparse_method is None:
            return 0.
        if self._threshold is None:
            self._threshold = float(np.float32(self.ppg_threshold.item()))
        return self._threshold

    def prepare_global_features(
        self,
        speakers,
        spectral_balance_ratios,
        loudness_ratios
    ):
        """(B, 258, 1) global conditioning (generator.py:49-70). `speakers`
        is (B,) integer ids, or with ZERO_SHOT (B, 512) WavLM x-vectors."""
        lib = _lib.lib()
        _lib.require_gpu(speakers)
        batch = speakers.shape[0]
        device = speakers.device

        def ratio(values, used):
            if not used:
                return None
            values = values.to(device=device, dtype=torch.float32).contiguous()
            if values.shape != (batch,):
                raise ValueError('ratios must have shape (B,)')
            return values
        sbr = ratio(spectral_balance_ratios, self.augment_pitch)
        lr = ratio(loudness_ratios, self.augment_loudness)
        channels = promonet_amd.SPEAKER_CHANNELS
        out = torch.empty(
            batch, channels + (sbr is not None) + (lr is not None), 1,
            device=device)
        with torch.cuda.device(device):
            if self.zero_shot:
                speakers = speakers.to(torch.float32).contiguous()
                weight = self.speaker_embedding.weight.detach().to(
                    torch.float32).contiguous()
                bias = self.speaker_embedding.bias.detach().to(
                    torch.float32).contiguous()
                if speakers.shape != (batch, weight.shape[1]):
                    raise ValueError(
                        f'zero-shot speakers must be (B, {weight.shape[1]})')
                _lib.check(lib.pm_prepare_global_features_linear(
                    _lib.ptr(speakers), _lib.ptr(weight), _lib.ptr(bias),
                    _lib.ptr(sbr), _lib.ptr(lr), _lib.ptr(out), batch,
                    weight.shape[1], channels, _lib.stream()))
            else:
                speakers = speakers.to(torch.long).contiguous()
                table = self.speaker_embedding.weight.detach().to(
                    torch.float32).contiguous()
                _lib.check(lib.pm_prepare_global_features(
                    _lib.ptr(speakers, torch.long), _lib.ptr(sbr),
                    _lib.ptr(lr), _lib.ptr(table), _lib.ptr(out), batch,
                    channels, table.shape[0], _lib.stream()))
        return out

    ###########################################################################
    # Packed (nn~) interface: generator.py:225-422. One (B, 53, frames)
    # tensor carries [loudness 8 | pitch | periodicity | ppg 40 | speaker |
    # spectral balance | loudness ratio]; the last three are read at frame 0.
    ###########################################################################

    def labels(self):
        """Semantic label of every packed input channel (generator.py:225-253)"""
        return (
            [f'loudness-{i}' for i in range(promonet_amd.LOUDNESS_BANDS)] +
            ['pitch', 'periodicity'] +
            [f'ppg-{i}' for i in range(promonet_amd.PPG_CHANNELS)] +
            ['speaker', 'spectral balance', 'loudness ratio'])

    def unpack_features(self, x):
        """Split a packed tensor into forward()'s arguments (:381-422)"""
        bands, channels = promonet_amd.LOUDNESS_BANDS, promonet_amd.PPG_CHANNELS
        first = bands + 2
        tail = first + channels
        return (
            x[:, :bands],
            x[:, bands],
            x[:, bands + 1],
            x[:, first:tail],
            x[:, tail, 0].to(torch.long),
            x[:, tail + 1, 0],
            x[:, tail + 2, 0])

    def pack_features(
        self, loudness, pitch, periodicity, ppg, speakers,
        spectral_balance_ratios, loudness_ratios
    ):
        """Inverse of unpack_features (:255-311): band-averaged loudness,
        sparsified PPG, per-utterance scalars repeated along time. pitch and
        periodicity are (B, 1, T) here, as in the reference."""
        frames = pitch.shape[-1]
        sparse = self.prepare_features(
            loudness, pitch[:, 0], periodicity[:, 0], ppg
        )[:, :promonet_amd.PPG_CHANNELS]
        averaged = promonet_amd.preprocess.loudness.band_average(loudness)

        def along_time(values):
            return values.to(
                device=pitch.device, dtype=torch.float32
            )[:, None, None].expand(-1, 1, frames)

        return torch.cat((
            averaged, pitch, periodicity, sparse, along_time(speakers),
            along_time(spectral_balance_ratios), along_time(loudness_ratios)),
            dim=1)

    def packed_inference(self, x, graph=False):
        """(B, 53, frames) -> (B, 1, 256 frames) float32 (:313-343).

        `graph=True` is the low-latency schedule for the streaming (nn~) use
        of this method - one small chunk per call, where the ~40 kernel
        launches of a forward cost more than the kernels: the whole call
        (unpack, feature preparation, vocoder) is captured ONCE per input
        shape as a hipGraph and replayed on every later call - one launch
        instead of forty, bit-identical output
        (tests/test_gpu_model.py::test_packed_inference_graph). HiFi-GAN only."""
        if graph:
            return self._packed_graph(x)
        unpacked = [t.contiguous() for t in self.unpack_features(x)]
        return self(*unpacked, self.default_previous_samples).to(torch.float)

    # captured graphs kept per Generator, oldest dropped first
    MAX_PACKED_GRAPHS = 8

    def _packed_graph(self, x):
        if not isinstance(self.model, HiFiGAN):
            raise ValueError(
                'packed_inference(graph=True) is implemented for the HiFi-GAN '
                'vocoder (FARGAN checks its inter-workgroup exchanges on the '
                'host after every forward)')
        _lib.require_gpu(x)
        x = x.to(torch.float32)
        self.model.engine()                    # (re)built outside any capture
        key = (tuple(x.shape), x.device, self.model._generation)
        cache = self.__dict__.setdefault('_packed_graphs', {})
        entry = cache.get(key)
        if entry is None:
            for stale in [k for k in cache if k[2] != key[2]]:
                del cache[stale]
            while len(cache) >= self.MAX_PACKED_GRAPHS:
                del cache[next(iter(cache))]
            entry = cache[key] = self._capture_packed(x)
        entry['input'].copy_(x)
        entry['graph'].replay()
        return entry['output'].clone()

    def _capture_packed(self, x):
        """One eager call (engine packed, host-side constants cached), then the
        capture. The graph gets a workspace of its OWN: the module's shared
        one may be re-allocated by a later, larger forward, and a captured
        kernel keeps the pointer it was recorded with."""
        vocoder = self.model
        # (inference mode OFF for the capture: the static input is written in
        # place on every replay, by callers inside AND outside
        # torch.inference_mode() - an inference tensor would refuse the second
        # kind - and the captured output is cloned by both)
        with torch.inference_mode(False), torch.no_grad(), \
                torch.cuda.device(x.device):
            static = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            static.copy_(x)
            self.packed_inference(static)
            shared = vocoder._workspace
            vocoder._workspace = None
            try:
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize(x.device)
                with torch.cuda.graph(graph):
                    output = self.packed_inference(static)
                private = vocoder._workspace
            finally:
                vocoder._workspace = shared
            torch.cuda.synchronize(x.device)
        return {'graph': graph, 'input': static, 'output': output,
                'workspace': private}

    def remove_weight_norm(self):
        self.model.remove_weight_norm()
