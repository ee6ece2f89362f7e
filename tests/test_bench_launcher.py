"""`python bench.py --gpus N` must itself start N ranks (one process per GPU, as pretrain_src/utils/distributed.py:53-72 does
under its launcher) and report n_gpus = N.  CPU check of the launch path only (gloo rendezvous + one all-reduce)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, GOAT_BENCH_LAUNCH_ONLY='1', GOAT_DIST_BACKEND='gloo', **extra_env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks():
    out = _run({}, '--gpus', '2', '--steps', '1', '--warmup', '0')
    assert out['n_gpus'] == 2 and out['gpus_flag'] == 2


def test_single_gpu_default_does_not_spawn():
    out = _run({})
    assert out['n_gpus'] == 1
