"""First module whose output contains NaN/Inf (full-size bf16 model, eval)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import vln_goat_amd
from vln_goat_amd import synth
from helpers import build_case
case, task = (sys.argv[1:3] + ['pretrain_config2_full', 'mlm'])[:2]
cfg, model, batch = build_case(case)
vln_goat_amd.set_compute_dtype(torch.bfloat16)
model = model.cuda().eval()
gb = synth.batch_to(batch, 'cuda')
bad = []
def hook(name):
    def h(m, i, o):
        outs = o if isinstance(o, (tuple, list)) else [o]
        for k, t in enumerate(outs):
            if torch.is_tensor(t) and t.is_floating_point() and not torch.isfinite(t.float()).all():
                nz = (~torch.isfinite(t.float())).nonzero()
                bad.append((name, k, tuple(t.shape), nz[0].tolist(), int(nz.shape[0])))
    return h
for n, m in model.named_modules():
    m.register_forward_hook(hook(n))
with torch.no_grad():
    loss = model(gb, task, compute_loss=True)
torch.cuda.synchronize()
print('loss nan count', int(torch.isnan(loss).sum()), 'of', loss.numel())
for b in bad[:12]:
    print(b)
