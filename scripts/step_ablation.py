"""Wall-time share of each kernel family in the replayed pre-training step, by ABLATION: the C-ABI entry points of one family are
replaced by no-ops (status 0, nothing launched) before the step graphs are captured, and the headline loop of bench.py is timed.
Results are garbage — only the clock is read.  What a family costs on the wall is (baseline - ablated), which, unlike the sum of
kernel durations in a trace, accounts for what already overlaps on the side streams.

    python scripts/step_ablation.py                # runs every ablation in a child process, prints the table
    python scripts/step_ablation.py --one ln_bwd   # (internal) one ablation in this process

Experiment tooling only (scripts/): the product library never skips a launch.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = {
    'none': [],
    'wgrad_grouped': ['goat_wgrad_grouped'],
    'ln_bwd': ['goat_ln_bwd', 'goat_ln_bwd_do', 'goat_ln_reduce_batched'],
    'ln_fwd': ['goat_ln_fwd', 'goat_ln_fwd_do'],
    'attn_fwd': ['goat_attn_fwd'],
    'attn_bwd': ['goat_attn_bwd'],
    'gemm_bf16': ['goat_gemm_bf16'],
    'act_dropout': ['goat_act_bwd', 'goat_dropout_add_fwd', 'goat_dropout_bwd'],
    'wgrad+ln_bwd': ['goat_wgrad_grouped', 'goat_ln_bwd', 'goat_ln_bwd_do', 'goat_ln_reduce_batched'],
    'ln+attn': ['goat_ln_bwd', 'goat_ln_bwd_do', 'goat_ln_reduce_batched', 'goat_ln_fwd', 'goat_ln_fwd_do', 'goat_attn_fwd', 'goat_attn_bwd'],
}


class _Proxy:
    def __init__(self, handle, skip):
        self._h = handle
        self._skip = set(skip)

    def __getattr__(self, name):
        if name in self._skip:
            return lambda *a: 0
        return getattr(self._h, name)


def one(name, steps, warmup, batch):
    sys.argv = ['bench.py', '--steps', str(steps), '--warmup', str(warmup), '--no-extra-configs', '--no-cpu-baseline', '--no-roofline']
    if batch:
        sys.argv += ['--batch', str(batch)]
    import bench
    from vln_goat_amd import _lib
    args = bench.parse()
    h = _lib.lib()
    _lib._lib = _Proxy(h, FAMILIES[name])
    m = bench.measure_pretrain(args, 1, 0, 'config2', steps, warmup)
    print('ABLATION ' + json.dumps({'name': name, 'ms_per_step': 1e3 * m['dt'] / steps}), flush=True)


def main():
    steps = int(os.environ.get('ABL_STEPS', '96'))
    batch = os.environ.get('ABL_BATCH')
    if '--one' in sys.argv:
        one(sys.argv[sys.argv.index('--one') + 1], steps, 6, batch)
        return
    res = {}
    for name in FAMILIES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--one', name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        for ln in r.stdout.splitlines():
            if ln.startswith('ABLATION '):
                res[name] = json.loads(ln[9:])['ms_per_step']
        if name not in res:
            print('%s failed:\n%s' % (name, r.stdout[-1500:]))
    base = res.get('none')
    print('replayed pre-training step (config 2, per-rank batch %s, mlm/sap/cfp 1:1:1), %d timed steps per row' % (batch or 48, steps))
    print('%-16s %10s %12s %8s' % ('entry points off', 'ms / step', 'saves ms', 'share'))
    for name, v in res.items():
        print('%-16s %10.3f %12.3f %7.1f%%' % (name, v, base - v, 100 * (base - v) / base))


if __name__ == '__main__':
    main()
