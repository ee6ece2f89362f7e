"""bf16 gradient noise of the door-gate parameters: HIP door_gate vs the torch formula, several batches, against the f32 run."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
import vln_goat_amd
from vln_goat_amd import hipops, synth
from helpers import build_case, CASES

def run(dtype, seed, patch=False):
    cfg, model, _ = build_case('pretrain_bacl_type2_door')
    kw = dict(CASES['pretrain_bacl_type2_door'][1]); kw['seed'] = seed
    batch = synth.make_pretrain_batch(**kw)
    vln_goat_amd.set_compute_dtype(dtype)
    old = hipops.door_gate
    if patch:
        def dg(aug_lin, ori_lin, aug, ori):
            w = torch.sigmoid(aug_lin(aug).float() + ori_lin(ori).float()).to(aug.dtype)
            return w * aug + (1 - w) * ori
        hipops.door_gate = dg
    try:
        model = model.cuda().eval()
        gb = synth.batch_to(batch, 'cuda')
        model(gb, 'sap', compute_loss=True).mean().backward()
        torch.cuda.synchronize()
        return {n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        hipops.door_gate = old
        vln_goat_amd.set_compute_dtype(torch.float32)

n = 'bert.lang_encoder.instr_aug_linear.weight'
for seed in (8, 101, 102, 103, 104, 105):
    ref = run(torch.float32, seed)
    a, b = run(torch.bfloat16, seed), run(torch.bfloat16, seed, True)
    agg = lambda g: sum(float((g[k] - ref[k]).norm()) for k in ref) / sum(float(ref[k].norm()) for k in ref)
    print('seed %3d  gate-weight rel err: hip %.3f  torch %.3f   | all-parameter aggregate: hip %.4f torch %.4f' % (
        seed, float((a[n] - ref[n]).norm() / ref[n].norm()), float((b[n] - ref[n]).norm() / ref[n].norm()), agg(a), agg(b)))
