"""Run eager fwd+bwd of every task with the caching allocator disabled (every tensor its own hipMalloc) and
blocking launches: an out-of-bounds access of any kernel then faults at the offending launch."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import hipops
class A: pass
args = A(); args.batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8; args.dtype = 'bf16'; args.layers = sys.argv[2] if len(sys.argv) > 2 else '2,2,2'
torch.cuda.set_device(0)
cfg, model, batch, gb, _static = bench.build(args, 0)
hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')
for rep in range(2):
    for task in bench.TASKS:
        for p in model.parameters():
            p.grad = None
        loss = model(gb, task, compute_loss=True)
        loss.mean().backward()
        torch.cuda.synchronize()
        print('ok', rep, task, float(loss.float().mean()))
