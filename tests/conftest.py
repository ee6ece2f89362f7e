import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')
    # The CPU oracle runs beside every GPU parity test.  torch's intra-op pool defaults to one thread per core — 256 on the GPU
    # boxes — and the small-model oracle is then dominated by waking the pool (bench.py's cpu_baseline measured the full-size
    # oracle 70x slower on 256 threads than on 64).  Results do not depend on the thread count beyond float summation order.
    import torch
    n = int(os.environ.get('GOAT_TEST_THREADS', '32'))
    if n > 0:
        torch.set_num_threads(max(1, min(n, os.cpu_count() or n)))


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
