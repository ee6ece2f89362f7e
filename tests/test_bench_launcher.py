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


import pytest  # noqa: E402


@pytest.mark.gpu
def test_eight_rank_bench_preflight_on_one_gpu():
    """The command the driver runs for the scaling curve — `python bench.py --gpus 8 ...` — end to end before an 8-GPU node ever sees it
    (VERDICT r4 #7): rank spawn under torch.distributed.run, rendezvous on 127.0.0.1, eight ranks (all on the test box's ONE GPU, so the
    collectives go through gloo: RCCL refuses two ranks per device), per-rank batches, the phased backward with its per-phase gradient
    exchange, the sparse exchange of the word-embedding gradient, the CFP all-gather, max-over-ranks timing and the `dp` diagnostics block
    of the JSON line.  A small model and batch: this checks plumbing, not speed."""
    env = dict(os.environ, GOAT_DIST_BACKEND='gloo', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'GOAT_BENCH_LAUNCH_ONLY'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--batch', '4',
                        '--layers', '2,1,1', '--no-cpu-baseline', '--no-roofline', '--no-extra-configs', '--no-autotune'],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['scaling'] == 'weak' and out['unit'] == 'trajectory-steps/s'
    assert out['config']['global_batch'] == 32 and out['config']['parallelism'] == 'dp8'
    assert out['value'] > 0 and out['steps'] == 3
    dp = out['dp']
    assert dp['ranks_seen_by_rccl'] == 8 and dp['backend'] == 'gloo' and len(dp['ranks']) == 8
    assert sorted(x['rank'] for x in dp['ranks']) == list(range(8))
    assert dp['n_phases'] and dp['n_phases'] >= 2                      # the phased backward (gradient exchange overlapped with the next phase)
    assert set(dp['allreduce_alone']) == {'mlm', 'sap', 'cfp'}
    for t, rec in dp['allreduce_alone'].items():
        assert rec['bytes'] and rec['bytes'] > 1e6 and rec['ms'] > 0, (t, rec)
    assert dp['compute_only_ms_per_step'] > 0 and 'exposed_comm_ms_per_step' in dp
