"""aten_sites.py for the fine-tuning rollout (bench.py's config-4 episode: B = 12, L = 200, 3 steps, BACL + FACL on): non-extension
launches of one eager episode (forward + loss + backward) by source line."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import collections
import traceback
from types import SimpleNamespace
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import vln_goat_amd
from vln_goat_amd import nav_model, synth, hipops

VIEWS = {'view', '_unsafe_view', 'reshape', 'select', 'slice', 'transpose', 't', 'expand', 'detach', 'unsqueeze', 'squeeze', 'as_strided',
         'split', 'split_with_sizes', 'unbind', 'permute', 'alias', 'view_as', 'narrow', 'chunk', '_reshape_alias', 'empty', 'empty_like',
         'empty_strided', 'new_empty', 'new_empty_strided', 'is_same_size', 'sym_size', 'sym_stride', 'sym_numel', 'stride', 'size',
         'is_pinned', '_local_scalar_dense', 'lift_fresh', 'unsafe_split', 'unsafe_chunk', 'is_contiguous', 'numel', 'dim', 'unflatten',
         'flatten', 'result_type', 'can_cast', 'is_nonzero', 'sym_storage_offset', 'storage_offset'}


def on_device(a):
    if torch.is_tensor(a):
        return a.is_cuda
    if isinstance(a, (list, tuple)):
        return any(on_device(x) for x in a)
    return False


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS and (on_device(args) or name in ('zeros', 'ones', 'full', 'arange', 'zeros_like', 'ones_like')):
            where = 'autograd engine'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'vln-goat_amd' in fr.filename or 'vln_goat_amd' in fr.filename:
                    where = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, (fr.line or '').strip()[:90])
                    break
            if where == 'autograd engine' and name in ('add', 'add_'):      # gradient accumulation: which tensors fan out?
                where += '  shape %s %s' % (tuple(args[0].shape), str(args[0].dtype).replace('torch.', ''))
            self.count[(name, where)] += 1
        return func(*args, **(kwargs or {}))


a = SimpleNamespace(num_l_layers=6, num_x_layers=3, num_pano_layers=2, dropout=0.1, feat_dropout=0.5, vocab_size=50265,
                    do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                    do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', mode='train')
torch.manual_seed(0)
torch.cuda.set_device(0)
model = nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(a)).cuda().train()
vln_goat_amd.set_compute_dtype(torch.bfloat16)
ep = synth.make_nav_episode(B=12, L=200, n_steps=3, seed=21, vocab_size=50265, extra_nodes=51)
mv = lambda x: x.cuda() if torch.is_tensor(x) else x
for st in ep['steps']:
    st['nav_fusion'] = nav_model.nav_fusion_matrix(st['vp_cand_vpids'], st['gmap_vpids'], st['gmap_visited_masks'],
                                                   st['gmap_step_ids'].shape[1], st['vp_masks'].shape[1])
ep = {k: ([{kk: mv(vv) for kk, vv in st.items()} for st in v] if k == 'steps' else mv(v)) for k, v in ep.items()}
hipops.manual_seed(1)
hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')


arena = [None]


def episode():
    if arena[0] is not None:
        arena[0].zero('nav')
    else:
        for p in model.parameters():
            p.grad = None
    loss, _ = synth.run_nav_episode(lambda m, b: model(m, b), ep, device='cuda', hoist_text_kv=True, hoist_pano=True)
    loss.backward()


from vln_goat_amd import dp
for _ in range(2):
    episode()
wrapper = dp.GoatDataParallel(model)            # as bench.py's config-4 leg: gradient arena, grouped weight gradients
wrapper.record_usage('nav')
for p in model.parameters():
    p.grad = None
arena[0] = wrapper.build_arena()
for _ in range(2):
    episode()
torch.cuda.synchronize()
with Sites() as s:
    episode()
torch.cuda.synchronize()
print('== config-4 episode: %d non-view aten calls on device tensors' % sum(s.count.values()))
for (name, where), c in sorted(s.count.items(), key=lambda kv: -kv[1]):
    print('  %3d  %-22s %s' % (c, name, where))
