"""Which parameters still receive their gradient through autograd's AccumulateGrad (a `grad += g` launch each) instead of a
kernel writing into the arena slice: hooks fire only when autograd delivers a gradient tensor."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import hipops, dp

class A: pass
args = A(); args.batch = 48; args.dtype = 'bf16'; args.layers = '6,3,2'
torch.cuda.set_device(0)
cfg, model, batch, gb, _static = bench.build(args, 0)
hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')
wrapper = dp.GoatDataParallel(model)
for task in bench.TASKS:
    for p in model.parameters():
        p.grad = None
    model(gb, task, compute_loss=True).mean().backward()
    wrapper.record_usage(task)
for p in model.parameters():
    p.grad = None
arena = wrapper.build_arena()
seen = collections.Counter()
TASK = ['']
for n, p in model.named_parameters():
    p.register_hook(lambda g, n=n: seen.update([(TASK[0], n, tuple(g.shape))]) or None)
for task in bench.TASKS:
    TASK[0] = task
    arena.zero(task)
    model(gb, task, compute_loss=True).mean().backward()
torch.cuda.synchronize()
for k, c in sorted(seen.items()):
    print('%-4s %3d  %-70s %s' % (k[0], c, k[1], k[2]))
