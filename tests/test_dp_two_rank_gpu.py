"""Two ranks on the one GPU of the test box (gloo rendezvous and collectives, HIP kernels for everything else): the N>1 bench
path — two-phase backward with the early all-reduce overlapped, sparse exchange of the word-embedding gradient — must give
the gradients of the plain path (one backward + dense all-reduce).  RCCL refuses two ranks per GPU, so the collectives run
through gloo here; `scripts/rccl_smoke.py` covers the RCCL API itself."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_sparse_embedding_and_two_phase_backward():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29561', os.path.join(ROOT, 'scripts', 'dp_sparse_check.py')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'DP_SPARSE_CHECK_OK' in out, out[-3000:]
