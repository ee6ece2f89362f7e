"""Source lines behind the ATen glue kernels of a step: leaf aten ops with device time, grouped by (op, autograd node or the innermost
vln-goat_amd python frame of the forward pass) — torch.profiler with_stack, one eager step per task of the mlm+sap+cfp cycle."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import hipops, dp
from torch.profiler import profile, ProfilerActivity

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
for rep in range(2):
    for task in bench.TASKS:
        arena.zero(task)
        model(gb, task, compute_loss=True).mean().backward()
torch.cuda.synchronize()
agg, tim = collections.Counter(), collections.Counter()
for task in bench.TASKS:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        arena.zero(task)
        model(gb, task, compute_loss=True).mean().backward()
        torch.cuda.synchronize()
    for ev in prof.events():
        if not ev.name.startswith('aten::'):
            continue
        dt = getattr(ev, 'self_device_time_total', 0) or 0
        if dt <= 0:
            continue
        node, par = None, ev.cpu_parent
        while par is not None:
            if 'evaluate_function' in par.name:
                node = 'bwd ' + par.name.split(':')[-1].strip()
                break
            par = par.cpu_parent
        where = ''
        for fr in (ev.stack or []):
            if 'vln-goat_amd' in fr or 'vln_goat_amd' in fr:
                where = fr.split('vln-goat_amd/')[-1].split('vln_goat_amd/')[-1][:70]
                break
        key = (ev.name, node or 'fwd', where)
        agg[key] += 1
        tim[key] += dt
print('per mlm+sap+cfp cycle: leaf aten ops with device time by autograd node / source line   (total %.1f us, %d launches)' % (sum(tim.values()), sum(agg.values())))
for k, c in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:90]:
    print('  %4d  %8.1f us  %-22s %-34s %s' % (c, tim[k], k[0], k[1], k[2]))
