import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a box without a GPU skips the gpu-marked tests instead of failing them.  With a GPU
    nothing is skipped: the HIP path must run — a missing libgoat_hip.so fails loudly there."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a GPU (marked gpu; run on the MI355X box)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
