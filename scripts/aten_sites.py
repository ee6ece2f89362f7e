"""Which source lines launch the ATen (non-HIP-extension) kernels of a step?  Runs one eager fwd + bwd per task under a
TorchDispatchMode, counts every aten op that is not a pure view, and groups by the innermost vln-goat_amd frame of the Python stack
(ops issued by autograd's own backward nodes have no such frame: listed as `autograd engine`, by op).
    python scripts/aten_sites.py [task ...]"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from vln_goat_amd import hipops, dp

VIEWS = {'view', '_unsafe_view', 'reshape', 'select', 'slice', 'transpose', 't', 'expand', 'detach', 'unsqueeze', 'squeeze', 'as_strided',
         'split', 'split_with_sizes', 'unbind', 'permute', 'alias', 'view_as', 'narrow', 'chunk', '_reshape_alias', 'empty', 'empty_like',
         'empty_strided', 'new_empty', 'new_empty_strided', 'is_same_size', 'sym_size', 'sym_stride', 'sym_numel', 'stride', 'size',
         'is_pinned', '_local_scalar_dense', 'lift_fresh', 'unsafe_split', 'unsafe_chunk', 'is_contiguous', 'numel', 'dim', 'unflatten',
         'flatten', 'result_type', 'can_cast', 'is_nonzero', 'sym_storage_offset', 'storage_offset', '_to_copy_view'}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            big = any(torch.is_tensor(a) and a.is_cuda for a in args) or name in ('zeros', 'ones', 'full', 'arange', 'zeros_like', 'ones_like')
            if big:
                where = 'autograd engine'
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if 'vln-goat_amd' in fr.filename or 'vln_goat_amd' in fr.filename:
                        where = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, (fr.line or '').strip()[:90])
                        break
                self.count[(name, where)] += 1
        return func(*args, **(kwargs or {}))


class A:
    pass


args = A(); args.batch = 48; args.dtype = 'bf16'; args.layers = '6,3,2'
torch.cuda.set_device(0)
cfg, model, batch, gb, _static = bench.build(args, 0)
hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')
wrapper = dp.GoatDataParallel(model)
for task in bench.TASKS:
    for p in model.parameters():
        p.grad = None
    hipops.backward_mean(model(gb, task, compute_loss=True))
    wrapper.record_usage(task)
for p in model.parameters():
    p.grad = None
arena = wrapper.build_arena()
for rep in range(2):
    for task in bench.TASKS:
        arena.zero(task)
        hipops.backward_mean(model(gb, task, compute_loss=True))          # (the step of bench.py)
torch.cuda.synchronize()
for task in (sys.argv[1:] or bench.TASKS):
    with Sites() as s:
        arena.zero(task)
        hipops.backward_mean(model(gb, task, compute_loss=True))          # (the step of bench.py)
    torch.cuda.synchronize()
    print('== %s: %d non-view aten calls on device tensors' % (task, sum(s.count.values())))
    for (name, where), c in sorted(s.count.items(), key=lambda kv: -kv[1]):
        print('  %3d  %-22s %s' % (c, name, where))
