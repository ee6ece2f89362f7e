"""Per-tensor bf16 gradient error of the HIP path next to stock torch.autocast(bfloat16) on the CPU oracle, both against the
fp32 oracle (calibration of tests/test_model_parity_gpu.py::_check_grads_bf16).   python scripts/diag_bf16_grads.py [case ...]"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import vln_goat_amd
from vln_goat_amd import synth
from helpers import build_case, case_tasks, oracle_run

cases = sys.argv[1:] or ['pretrain_small_fixed', 'pretrain_small_ragged', 'pretrain_reverie_small', 'pretrain_bacl_type2_door']
for case in cases:
    for task in case_tasks(case):
        cfg, model, batch = build_case(case)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        _, ref = oracle_run(cfg, sd, batch, task)
        _, ac = oracle_run(cfg, sd, batch, task, autocast_bf16=True)
        vln_goat_amd.set_compute_dtype(torch.bfloat16)
        model = model.cuda().eval()
        model(synth.batch_to(batch, 'cuda'), task, compute_loss=True).mean().backward()
        torch.cuda.synchronize()
        vln_goat_amd.set_compute_dtype(torch.float32)
        gmax = max(float(g.norm()) for g in ref.values() if g is not None)
        rows = []
        for n, p in model.named_parameters():
            rg = ref.get(n)
            if rg is None or float(rg.norm()) <= 2e-3 * gmax or p.grad is None:
                continue
            rn = float(rg.double().norm())
            g, ga = p.grad.double().cpu(), ac[n].double()
            rows.append((float((g - rg.double()).norm()) / rn, float((ga - rg.double()).norm()) / rn, abs(float(g.norm()) / rn - 1),
                         abs(float(ga.norm()) / rn - 1), rn / gmax, n))
        agg_h = sum(r[0] * r[4] for r in rows) / sum(r[4] for r in rows)
        agg_a = sum(r[1] * r[4] for r in rows) / sum(r[4] for r in rows)
        print('== %s %s: %d tensors, aggregate e_hip %.4f e_ac %.4f | median e_hip/e_ac %.2f' % (
            case, task, len(rows), agg_h, agg_a, sorted(r[0] / max(r[1], 1e-9) for r in rows)[len(rows) // 2]))
        for r in sorted(rows, key=lambda r: -r[0] / max(r[1], 0.01))[:6]:
            print('   worst e ratio: e_hip %.4f e_ac %.4f  r_hip %.4f r_ac %.4f  |g|/gmax %.3f  %s' % r)
        for r in sorted(rows, key=lambda r: -r[2])[:4]:
            print('   worst norm dev: e_hip %.4f e_ac %.4f  r_hip %.4f r_ac %.4f  |g|/gmax %.3f  %s' % r)
