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


def test_two_rank_finetune_iteration_equals_single_process():
    """BASELINE configs[3]: dp.wrap_finetune_models around the HIP navigation model + critic; 2 ranks x 2 samples == 1 process x 4 samples
    (float32 and bf16 wire), scripts/dp_nav_check.py."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29562', os.path.join(ROOT, 'scripts', 'dp_nav_check.py')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'DP_NAV_CHECK_OK' in out, out[-3000:]


def test_gradient_exchange_captured_inside_the_step_graph_equals_the_host_launched_exchange():
    """One-rank RCCL group + dp.FORCE_COLLECTIVES: every phase's exchange forked inside ONE captured step graph (both wire formats) gives
    the gradients of the host-launched exchange between phases, scripts/in_graph_comm_check.py."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'in_graph_comm_check.py')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=900, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'IN_GRAPH_COMM_OK' in out, out[-3000:]
