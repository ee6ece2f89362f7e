"""Module-tree contract of the reference models and an optimizer trace of the reference's AdamW, written as fixtures
(this container only; nothing here runs on the GPU box).

    python tests/golden/make_contract.py pretrain    -> tests/golden/contract_pretrain.json, adamw_trace.npz
    python tests/golden/make_contract.py nav         -> tests/golden/contract_nav.json        (separate process: clashing packages)

contract_*.json: for the reference model built from its shipped configuration — every `torch.nn.Dropout` module name and p
(what P/utils/misc.py:19-25 `set_dropout` iterates), every parameter name and shape, and the weight-decay grouping of
P/optim/misc.py:13-23.  adamw_trace.npz: three steps of P/optim/adamw.py:53-110 (+ clip_grad_norm_ 5.0 as in
P/train_r2r_goat.py:349-366) on seeded tensors — the pin of the fused HIP optimizer.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402


def contract(model):
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']          # P/optim/misc.py:13
    return {
        'dropout': {n: float(m.p) for n, m in model.named_modules() if isinstance(m, torch.nn.Dropout)},
        'params': {n: list(p.shape) for n, p in model.named_parameters()},
        'no_decay': sorted(n for n, _ in model.named_parameters() if any(nd in n for nd in no_decay)),
        'state_dict_keys': sorted(model.state_dict().keys()),
    }


def pretrain():
    pg = ref_shim.import_pretrain()
    out = {}
    for tag, js, over in (('r2r', 'r2r_GOAT_model_config.json', dict(pretrain_tasks={'mlm', 'sap', 'cfp'}, name='R2R')),
                          ('reverie', 'reverie_GOAT_model_config.json', dict(pretrain_tasks={'mlm', 'mrc', 'sap', 'og', 'cfp'}, name='REVERIE'))):
        cfg = ref_shim.make_config('/root/reference/pretrain_src/config/' + js, **over)
        torch.manual_seed(0)
        m = pg.GlocalTextPathCMTPreTraining(cfg)
        out[tag] = contract(m)
        if tag == 'r2r':         # init statistics of a-17 (HF _init_weights N(0, 0.02), LN (1, 0), tim_*_attn U(-0.1, 0.1))
            sd = m.state_dict()
            out['init_stats'] = {k: [float(sd[k].float().mean()), float(sd[k].float().std()), float(sd[k].min()), float(sd[k].max())]
                                 for k in ('bert.lang_encoder.layer.0.attention.self.query.weight', 'bert.embeddings.word_embeddings.weight',
                                           'bert.img_embeddings.img_linear.weight', 'tim_txt_attn', 'tim_global_attn',
                                           'bert.lang_encoder.layer.0.attention.output.LayerNorm.weight',
                                           'bert.lang_encoder.layer.0.attention.output.LayerNorm.bias',
                                           'bert.lang_encoder.layer.0.attention.self.query.bias')}
    with open(os.path.join(HERE, 'contract_pretrain.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('wrote contract_pretrain.json', {k: len(v.get('params', v)) for k, v in out.items()})

    # ---- reference AdamW trace -------------------------------------------------------------------------------------------------
    spec = importlib.util.spec_from_file_location('ref_adamw', '/root/reference/pretrain_src/optim/adamw.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rs = np.random.RandomState(3)
    shapes = {'enc.dense.weight': (96, 64), 'enc.dense.bias': (96,), 'enc.LayerNorm.weight': (64,), 'enc.LayerNorm.bias': (64,),
              'emb.word.weight': (300, 64), 'head.weight': (1, 64), 'unused.weight': (8, 8)}
    params = {n: torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32) * 0.05)) for n, s in shapes.items()}
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    groups = [{'params': [p for n, p in params.items() if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01},
              {'params': [p for n, p in params.items() if any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
    opt = mod.AdamW(groups, lr=5e-5, betas=(0.9, 0.98))      # r2r_GOAT_pretrain.json: learning_rate 5e-5, betas [0.9, 0.98], weight_decay 0.01
    store = {'names': np.array(list(shapes)), 'lr': np.array([5e-5, 1e-4, 2.5e-5], dtype=np.float64)}
    for n, p in params.items():
        store['p0_' + n] = p.detach().numpy().copy()
    for step in range(3):
        for n, p in params.items():
            if n == 'unused.weight' or (n == 'head.weight' and step == 1):      # no gradient: the optimizer skips the tensor
                p.grad = None
                continue
            g = rs.standard_normal(shapes[n]).astype(np.float32) * (3.0 if step == 0 else 0.02)     # step 0 is clipped, 1-2 are not
            store['g%d_%s' % (step, n)] = g
            p.grad = torch.from_numpy(g.copy())
        for grp in opt.param_groups:
            grp['lr'] = float(store['lr'][step])
        gn = torch.nn.utils.clip_grad_norm_(list(params.values()), 5.0)
        store['gnorm%d' % step] = np.array([float(gn)], dtype=np.float64)
        opt.step()
        for n, p in params.items():
            store['p%d_%s' % (step + 1, n)] = p.detach().numpy().copy()
    for n, p in params.items():
        st = opt.state.get(p, {})
        if st:
            store['m_' + n], store['v_' + n] = st['exp_avg'].numpy().copy(), st['exp_avg_sq'].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'adamw_trace.npz'), **store)
    print('wrote adamw_trace.npz', [float(store['gnorm%d' % i][0]) for i in range(3)])


def nav():
    vg = ref_shim.import_nav()
    from types import SimpleNamespace
    from vln_goat_amd import nav_model
    out = {}
    for tag, over in (('r2r', {}), ('reverie', dict(dataset='reverie', obj_feat_size=768))):
        args = SimpleNamespace(num_l_layers=6, num_x_layers=3, num_pano_layers=2, dropout=0.1, feat_dropout=0.5, do_back_img=True,
                               do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True, do_back_txt_type='type_2',
                               do_back_img_type='type_1', do_add_method='door', mode='train', **over)
        cfg = nav_model.nav_config_from_args(args)
        torch.manual_seed(0)
        out[tag] = contract(vg.GlocalTextPathNavCMT(cfg))
    with open(os.path.join(HERE, 'contract_nav.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('wrote contract_nav.json', {k: len(v['params']) for k, v in out.items()})


if __name__ == '__main__':
    nav() if sys.argv[1:] == ['nav'] else pretrain()
