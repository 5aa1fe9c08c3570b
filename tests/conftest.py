import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))

GOLDEN = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real AMD GPU (run on the MI355X box)')


@pytest.fixture(scope='session')
def golden_default():
    import torch
    return torch.load(GOLDEN / 'generator_default.pt', weights_only=False)


@pytest.fixture(scope='session')
def golden_small():
    import torch
    return torch.load(GOLDEN / 'generator_small.pt', weights_only=False)


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def golden_fargan():
    import torch
    return torch.load(GOLDEN / 'generator_fargan.pt', weights_only=False)
