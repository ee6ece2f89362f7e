"""Which lines of the package launch the ATen glue kernels of a step (fills, adds, copies, casts ...)?  One eager step per task
under torch.profiler with Python stacks; ops are grouped by (aten op, innermost frame inside vln-goat_amd/).
    python scripts/glue_sources.py [batch]"""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import hipops, dp

class A: pass
args = A(); args.batch = int(sys.argv[1]) if len(sys.argv) > 1 else 48; args.dtype = 'bf16'; args.layers = '6,3,2'
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
from torch.profiler import profile, ProfilerActivity
for task in bench.TASKS:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        arena.zero(task)
        model(gb, task, compute_loss=True).mean().backward()
        torch.cuda.synchronize()
    agg = collections.Counter()
    tim = collections.Counter()
    for ka in prof.key_averages(group_by_stack_n=12):
        if not ka.key.startswith('aten::'):
            continue
        dt = getattr(ka, 'self_device_time_total', 0) or getattr(ka, 'self_cuda_time_total', 0)
        if dt <= 0:
            continue
        frame = 'outside the package'
        for fr in ka.stack:
            if ('vln-goat_amd' in fr or 'vln_goat_amd' in fr or 'bench.py' in fr or 'synth.py' in fr) and 'torch/' not in fr:
                frame = fr.strip().split('/')[-1]
                break
        agg[(ka.key, frame)] += ka.count
        tim[(ka.key, frame)] += dt
    print('=== task %s: aten ops with device time (count, total us) by innermost package frame' % task)
    for (name, frame), c in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:60]:
        print('  %4d  %8.1f us  %-28s %s' % (c, tim[(name, frame)], name, frame))
