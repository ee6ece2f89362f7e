"""Diagnostic: per-parameter gradient error of the HIP model vs the CPU oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from helpers import build_case, oracle_run
import vln_goat_amd
from vln_goat_amd import synth

case = sys.argv[1] if len(sys.argv) > 1 else 'pretrain_small_ragged'
for dtype in (torch.float32, torch.bfloat16):
    for task in ('mlm', 'sap', 'cfp'):
        cfg, model, batch = build_case(case)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        ref_loss, ref_grads = oracle_run(cfg, sd, batch, task)
        vln_goat_amd.set_compute_dtype(dtype)
        model = model.cuda().eval()
        gb = synth.batch_to(batch, 'cuda')
        loss = model(gb, task, compute_loss=True)
        loss.mean().backward()
        torch.cuda.synchronize()
        gmax = max(float(g.norm()) for g in ref_grads.values() if g is not None)
        rows = []
        for n, p in model.named_parameters():
            rg = ref_grads.get(n)
            if rg is None or p.grad is None:
                if (rg is None) != (p.grad is None) and not (rg is None and float(p.grad.norm()) == 0):
                    rows.append((9.99, n, 'presence mismatch', 0, 0))
                continue
            d = float((p.grad.double().cpu() - rg.double()).norm())
            rn = float(rg.double().norm())
            rows.append((d / max(rn, 1e-30), n, d, rn, d / gmax))
        rows.sort(reverse=True)
        print('==== %s %s %s  loss err %.2e  (gmax %.3e)' % (case, task, dtype, float((loss.detach().float().cpu() - ref_loss).abs().max()), gmax))
        for r in rows[:12]:
            print('   rel %.3e  %-70s abs %s ref %s abs/gmax %s' % r)
