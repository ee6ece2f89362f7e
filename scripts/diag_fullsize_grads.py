"""bf16 vs fp32 parameter gradients of the HIP model at full size (which tensors are off, and by how much)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import vln_goat_amd
from vln_goat_amd import synth, hipops
from helpers import build_case
case, task = (sys.argv[1:3] + ['pretrain_config2_full', 'mlm'])[:2]
if 'notuned' in sys.argv:
    hipops._TUNED.clear()
cfg, model, batch = build_case(case)
model = model.cuda().eval()
gb = synth.batch_to(batch, 'cuda')
res = {}
for dt in (torch.float32, torch.bfloat16):
    vln_goat_amd.set_compute_dtype(dt)
    for p in model.parameters():
        p.grad = None
    model(gb, task, compute_loss=True).mean().backward()
    torch.cuda.synchronize()
    res[dt] = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
vln_goat_amd.set_compute_dtype(torch.float32)
rows = []
for n, g32 in res[torch.float32].items():
    g16 = res[torch.bfloat16].get(n)
    if g16 is None:
        print('missing in bf16:', n); continue
    rn = float(g32.norm())
    rows.append((abs(float(g16.norm()) / max(rn, 1e-30) - 1), float((g16 - g32).norm()) / max(rn, 1e-30), rn, float(g16.norm()), n))
rows.sort(reverse=True)
for r in rows[:25]:
    print('norm-dev %.3e  rel-err %.3e  |g32| %.4e  |g16| %.4e  %s' % r)
