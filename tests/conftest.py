import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))

GOLDEN = ROOT / 'tests' / 'golden'

# the library's test hooks (pm_debug_force / pm_debug_skew) are off in a
# production process: opt in before it is loaded
os.environ.setdefault('PROMONET_HIP_DEBUG', '1')


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


def pytest_sessionfinish(session, exitstatus):
    """PM_RECORD_ERRORS=1: dump the largest error every parity check measured
    (tests/util.py::check) so that the gates can be set from it."""
    import json
    import os
    if not os.environ.get('PM_RECORD_ERRORS'):
        return
    try:
        import util
    except ImportError:
        return
    if not util.MEASURED:
        return
    out = ROOT / 'gpurun_out'
    out.mkdir(exist_ok=True)
    path = out / 'measured_errors.json'
    previous = {}
    if path.exists():
        previous = json.loads(path.read_text())
    for key, value in util.MEASURED.items():
        previous[key] = max(previous.get(key, 0.), value)
    path.write_text(json.dumps(previous, indent=1, sort_keys=True))
