"""Python call sites (inside the package) of the tensor-creating / casting / filling calls of one eager step per task:
torch.zeros / zeros_like / full / empty?no, Tensor.zero_ / fill_ / to / float / contiguous / clone, torch.cat / stack / arange.
Counts per (function, file:line) — the sources of the ATen fill / copy / cat kernels the kernel trace shows."""
import sys, os, collections, traceback
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
for rep in range(2):
    for task in bench.TASKS:
        arena.zero(task)
        model(gb, task, compute_loss=True).mean().backward()
torch.cuda.synchronize()

counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ('vln-goat_amd' in fr.filename or 'vln_goat_amd' in fr.filename) :
            return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    return 'outside'
def wrap_fn(mod, name, tag, cond=None):
    orig = getattr(mod, name)
    def f(*a, **k):
        r = orig(*a, **k)
        try:
            t = r if torch.is_tensor(r) else (a[0] if a and torch.is_tensor(a[0]) else None)
            if t is not None and t.is_cuda and (cond is None or cond(a, k, r)):
                counts[(TASK[0], tag, site())] += 1
        except Exception:
            pass
        return r
    setattr(mod, name, f)
TASK = ['']
for n in ('zeros', 'zeros_like', 'full', 'ones', 'cat', 'stack', 'arange', 'where'):
    wrap_fn(torch, n, 'torch.' + n)
for n in ('zero_', 'fill_', 'clone', 'masked_fill', 'masked_fill_', 'new_zeros', 'index_select', 'gather', 'sum', 'mean', 'add', 'add_', 'mul', '__add__', '__mul__', '__sub__', '__rsub__', '__truediv__', 'logical_not', '__lt__', '__ne__', '__invert__', 'squeeze', 'unsqueeze'):
    if n in ('squeeze', 'unsqueeze'):
        continue
    wrap_fn(torch.Tensor, n, 'Tensor.' + n)
wrap_fn(torch.Tensor, 'to', 'Tensor.to (copy)', lambda a, k, r: r.data_ptr() != a[0].data_ptr())
wrap_fn(torch.Tensor, 'float', 'Tensor.float (copy)', lambda a, k, r: r.data_ptr() != a[0].data_ptr())
wrap_fn(torch.Tensor, 'contiguous', 'Tensor.contiguous (copy)', lambda a, k, r: r.data_ptr() != a[0].data_ptr())
for task in bench.TASKS:
    TASK[0] = task
    arena.zero(task)
    model(gb, task, compute_loss=True).mean().backward()
torch.cuda.synchronize()
tot = collections.Counter()
for (task, tag, s), c in counts.items():
    tot[(tag, s)] += c
print('calls per mlm+sap+cfp cycle, by (call, innermost package frame):')
for (tag, s), c in sorted(tot.items(), key=lambda kv: -kv[1])[:90]:
    print('  %4d  %-28s %s   [%s]' % (c, tag, s, ' '.join('%s:%d' % (t, counts[(t, tag, s)]) for t in bench.TASKS if counts[(t, tag, s)])))
