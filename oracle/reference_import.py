"""Import the real reference (`/root/reference/promonet`) in THIS container.

TEST INFRASTRUCTURE ONLY. Never imported by the product (`promonet_amd`),
never shipped to / used on the GPU box (`/root/reference` does not exist
there). It exists so that `oracle/make_golden.py` can pin the CPU restatement
(`oracle/restatement.py`) against the reference's own Python and emit the
committed fixtures under `tests/golden/`.

The reference imports ~18 third-party modules that are not installed here
(yapecs, torchutil, ppgs, penn, librosa, torchaudio, ...). None of them is on
the synthesis hot path except `ppgs.sparsify` and a few `librosa` functions
(both absent => "parity unpinned"). We install `sys.modules`
stubs *before* import. The stubs only provide what is touched at import time.
"""
import argparse
import contextlib
import importlib
import importlib.abc
import importlib.machinery
import sys
import time
import types
from pathlib import Path
from unittest import mock

REFERENCE_ROOT = Path('/root/reference')


def available():
    return (REFERENCE_ROOT / 'promonet' / '__init__.py').exists()


def _sparsify(ppg, method='percentile', threshold=0.85):
    """Restatement of the published `ppgs.sparsify` (package absent here)."""
    import torch
    if method == 'percentile':
        q = torch.quantile(ppg, threshold, dim=-2, keepdim=True)
        ppg = torch.where(ppg > q, ppg, torch.zeros_like(ppg))
    elif method == 'constant':
        ppg = torch.where(ppg > threshold, ppg, torch.zeros_like(ppg))
    elif method == 'topk':
        k = int(threshold)
        kth = torch.topk(ppg, k, dim=-2).values[..., -1:, :]
        ppg = torch.where(ppg >= kth, ppg, torch.zeros_like(ppg))
    else:
        raise ValueError(method)
    return torch.softmax(torch.log(ppg + 1e-8), -2)


def _install_stubs(config_files):
    import torch

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # yapecs: apply `--config`-style files onto the defaults module
    def configure(name, defaults):
        for file in config_files:
            scope = {}
            exec(compile(Path(file).read_text(), str(file), 'exec'), scope)
            if scope.get('MODULE') != name:
                continue
            for key, value in scope.items():
                if key.isupper() and key != 'MODULE' and hasattr(defaults, key):
                    setattr(defaults, key, value)

    module(
        'yapecs',
        configure=configure,
        ArgumentParser=argparse.ArgumentParser)
    module('GPUtil', getGPUs=lambda: [])

    # torchutil
    seconds = {}

    @contextlib.contextmanager
    def time_context(name):
        start = time.perf_counter()
        try:
            yield
        finally:
            seconds[name] = seconds.get(name, 0.) + time.perf_counter() - start

    @contextlib.contextmanager
    def inference_context(model):
        was_training = model.training
        model.eval()
        with torch.inference_mode():
            yield
        model.train(was_training)

    class _Metric:
        def __init__(self, *a, **k):
            pass

        def reset(self):
            pass

        def update(self, *a, **k):
            pass

        def __call__(self):
            return {}

    def notify(*a, **k):
        return lambda fn: fn

    torchutil = module('torchutil', notify=notify)
    torchutil.time = module(
        'torchutil.time',
        context=time_context,
        reset=seconds.clear,
        results=lambda: dict(seconds))
    torchutil.inference = module(
        'torchutil.inference', context=inference_context)
    torchutil.metrics = module(
        'torchutil.metrics',
        RMSE=type('RMSE', (_Metric,), {}),
        L1=type('L1', (_Metric,), {}),
        PearsonCorrelation=type('PearsonCorrelation', (_Metric,), {}),
        Average=type('Average', (_Metric,), {}),
        Accuracy=type('Accuracy', (_Metric,), {}))
    for name in ('checkpoint', 'tensorboard', 'download', 'iterator', 'cuda',
                 'gradients', 'paths'):
        setattr(torchutil, name, mock.MagicMock())
        sys.modules[f'torchutil.{name}'] = getattr(torchutil, name)

    import restatement
    ppgs = module(
        'ppgs',
        REPRESENTATION_KIND='ppg',
        SIMILARITY_EXPONENT=1.2,
        # explicit phoneme inventory (the restated third-party constants) so
        # that the reference's selective time-stretch loop (edit/core.py:57-110)
        # can run for the goldens
        PHONEMES=list(restatement.PHONEMES),
        VOICED=list(restatement.VOICED),
        PHONEME_TO_INDEX_MAPPING={
            p: i for i, p in enumerate(restatement.PHONEMES)},
        sparsify=_sparsify,
        representation_file_extension=lambda: 'ppg')
    for name in ('edit', 'load', 'preprocess', 'plot', 'data'):
        setattr(ppgs, name, mock.MagicMock())
        sys.modules[f'ppgs.{name}'] = getattr(ppgs, name)
    # ppgs.edit.grid constructors used by promonet.edit (restated, unpinned)
    def of_length(tensor, length):
        return torch.linspace(0., tensor.shape[-1] - 1., int(length))

    ppgs.edit.grid.of_length = of_length
    ppgs.edit.grid.constant = lambda tensor, ratio: of_length(
        tensor, round(tensor.shape[-1] / ratio + 1e-4))
    ppgs.from_audio = mock.MagicMock()
    ppgs.distance = mock.MagicMock()
    # pypar: only the silence token is read on this path (edit/core.py:66,74)
    pypar = mock.MagicMock(name='pypar')
    pypar.SILENCE = restatement.SILENCE
    sys.modules['pypar'] = pypar

    # Everything else: any (sub)module of these top-level names is a mock
    mocked = {
        'penn', 'librosa', 'torchaudio', 'pyworld', 'resampy',
        'soundfile', 'jiwer', 'umap', 'torbi', 'whisper', 'vocos',
        'transformers', 'matplotlib', 'pysodic', 'pyfoal', 'pyloudnorm',
        'torchcrepe', 'encodec', 'g2p_en', 'huggingface_hub'}

    class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname.split('.')[0] in mocked:
                return importlib.machinery.ModuleSpec(
                    fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = mock.MagicMock(name=spec.name)
            m.__name__ = spec.name
            m.__path__ = []
            m.__spec__ = spec
            m.__loader__ = self
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, _MockFinder())


def load(config_files=()):
    """Import and return the reference `promonet` package (stubbed deps)."""
    if not available():
        raise RuntimeError('reference not present at /root/reference')
    if 'promonet' in sys.modules:
        return sys.modules['promonet']
    _install_stubs([Path(f) for f in config_files])
    sys.path.insert(0, str(REFERENCE_ROOT))
    try:
        import promonet
    finally:
        sys.path.remove(str(REFERENCE_ROOT))
    return promonet
