"""Run-to-run reproducibility of one task's fwd+bwd (same seeds, same batch, no arena): per-parameter relative difference
of the gradients of two runs, largest first — float atomics give ~1e-7; anything larger is a race or an unseeded draw.
    python scripts/diag_determinism.py [task] [bf16|f32] [case]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
import vln_goat_amd
from vln_goat_amd import hipops
from helpers import build_case

task = sys.argv[1] if len(sys.argv) > 1 else 'cfp'
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == 'bf16') else torch.float32
case = sys.argv[3] if len(sys.argv) > 3 else 'pretrain_small_fixed'
from vln_goat_amd import synth
cfg, model, batch = build_case(case)
model = model.cuda().eval()
gb = synth.batch_to(batch, 'cuda')
vln_goat_amd.set_compute_dtype(dtype)


def run():
    hipops.manual_seed(5)
    for p in model.parameters():
        p.grad = None
    out = model(gb, task, compute_loss=True)
    out.mean().backward()
    torch.cuda.synchronize()
    return out.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}

l0, g0 = run()
for rep in range(3):
    l1, g1 = run()
    rows = []
    for n in g0:
        d = float((g0[n].double() - g1[n].double()).norm())
        s = float(g0[n].double().norm())
        rows.append((d / max(s, 1e-30), d, s, n))
    rows.sort(reverse=True)
    print('rep %d: loss diff %.3e ; params differing > 1e-5 relative: %d of %d' % (rep, float((l0 - l1).abs().max()), sum(r[0] > 1e-5 for r in rows), len(rows)))
    for r in rows[:int(os.environ.get('DIAG_TOP', '12'))]:
        print('   %.3e  (|d| %.3e, |g| %.3e)  %s' % r)
