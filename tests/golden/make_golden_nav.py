"""Golden vectors for the fine-tuning (navigation) path FROM THE IMPORTED REFERENCE (this container only):
`GlocalTextPathNavCMT` (/root/reference/map_nav_src/models/vilmodel_GOAT.py:556) with BACL (type_2 + door and
type_1) and FACL on, driven through language -> (panorama -> navigation) x 3 steps with BPTT through [MEM]
by vln_goat_amd.synth.run_nav_episode.  Run in its own process (the pre-training tree
defines clashing top-level packages):   python tests/golden/make_golden_nav.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import projections  # noqa: E402

CASES = {
    'nav_type2_door': dict(do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door'),
    'nav_type1_add': dict(do_back_txt_type='type_1', do_back_img_type='type_2', do_add_method='add'),
    # REVERIE: object tokens in every panorama + object-grounding head
    'nav_reverie_objects': dict(do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', dataset='reverie',
                                obj_feat_size=768),
    # BASELINE.json configs[3] at the size of the fine-tuning script (M/scripts/run_r2r_goat.sh: 6/3/2 layers, batch 12,
    # max_instr_len 200, BACL + FACL on, dictionaries 35/39/50/24) with a G ~ 60 global map, full vocabulary
    'nav_config4_full': dict(do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door'),
}
FULL = {'nav_config4_full': dict(num_l_layers=6, num_x_layers=3, num_pano_layers=2, vocab_size=50265, dropout=0.1, feat_dropout=0.5)}
EPISODE = {'nav_reverie_objects': dict(objects=5, seed=9),
           'nav_config4_full': dict(B=12, L=200, n_steps=3, seed=21, vocab_size=50265, extra_nodes=51)}
WEIGHT_SEED = 11
VOCAB = 1200


def fingerprint(g):
    if g is None:
        return np.zeros(9, dtype=np.float32)
    flat = g.detach().float().reshape(-1)
    first = torch.zeros(8)
    first[:min(8, flat.numel())] = flat[:8]
    return np.concatenate([[float(flat.double().norm())], first.numpy()]).astype(np.float32)


def main(only=None):
    vg = ref_shim.import_nav()
    from vln_goat_amd import nav_model, synth
    for name, over in CASES.items():
        if only and name not in only:
            continue
        args = SimpleNamespace(**{**dict(num_l_layers=2, num_x_layers=2, num_pano_layers=2, dropout=0.5, feat_dropout=0.4,
                                         do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                                         vocab_size=VOCAB, mode='train'), **FULL.get(name, {}), **over})
        cfg = nav_model.nav_config_from_args(args)
        torch.manual_seed(0)
        ref = vg.GlocalTextPathNavCMT(cfg)
        ours = nav_model.GlocalTextPathNavCMT(cfg)
        sd = synth.seeded_state_dict(ours, seed=WEIGHT_SEED)
        rk, ok = set(ref.state_dict().keys()), set(sd.keys())
        assert rk == ok, (sorted(rk - ok)[:8], sorted(ok - rk)[:8])
        for k, v in ref.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
        ref.load_state_dict(sd)
        ref.eval()
        ep = synth.make_nav_episode(**{**dict(B=2, L=44, n_steps=3, seed=5, vocab_size=VOCAB), **EPISODE.get(name, {})})
        for k in ('front_txt_feats', 'front_vp_feats', 'front_gmap_feats', 'instr_z_direction_features', 'instr_z_landmark_features',
                  'z_img_features'):
            ep[k].requires_grad_(True)
        loss, rec = synth.run_nav_episode(lambda m, b: ref(m, b), ep)
        loss.backward()
        store = {'loss': np.array([float(loss)], dtype=np.float32),
                 'param_names': np.array([n for n, _ in ref.named_parameters()]),
                 'grad_fp': np.stack([fingerprint(p.grad) for _, p in ref.named_parameters()]),
                 'txt_embeds': rec['txt_embeds'][:, :, :16].detach().numpy()}
        for k in ('front_txt_feats', 'front_gmap_feats', 'z_img_features', 'instr_z_direction_features'):
            store['dinput_' + k] = fingerprint(ep[k].grad)
            store['dproj_' + k] = projections(ep[k].grad)
        # seeded random projections of every parameter gradient (helpers.projections: all elements covered)
        store['grad_proj'] = np.stack([projections(p.grad) for _, p in ref.named_parameters()])
        # What stock bf16 autocast does to THIS model's gradients (the yardstick of the bf16 GPU test, VERDICT r4 #5b): the same episode
        # on the reference under torch.autocast('cpu', bfloat16); per parameter the relative L2 error and the norm ratio against the
        # float32 gradients above, and the norm-weighted aggregate.
        g32 = {n: (p.grad.detach().double().clone() if p.grad is not None else None) for n, p in ref.named_parameters()}
        ref.zero_grad(set_to_none=True)
        for k in ('front_txt_feats', 'front_vp_feats', 'front_gmap_feats', 'instr_z_direction_features', 'instr_z_landmark_features',
                  'z_img_features'):
            ep[k].grad = None
        with torch.autocast('cpu', dtype=torch.bfloat16):
            loss_ac, _ = synth.run_nav_episode(lambda m, b: ref(m, b), ep)
        loss_ac.float().backward()
        errs, ratios = [], []
        for n, p in ref.named_parameters():
            r = g32[n]
            if r is None or p.grad is None or float(r.norm()) == 0.0:
                errs.append(0.0)
                ratios.append(0.0)
                continue
            g = p.grad.detach().double()
            errs.append(float((g - r).norm() / r.norm()))
            ratios.append(abs(float(g.norm() / r.norm()) - 1.0))
        store['grad_err_autocast'] = np.array(errs, dtype=np.float32)
        store['grad_norm_ratio_autocast'] = np.array(ratios, dtype=np.float32)
        store['loss_autocast'] = np.array([float(loss_ac)], dtype=np.float32)
        for t, s in enumerate(rec['steps']):
            for k in ('global_logits', 'local_logits', 'fused_logits', 'cls_embeds'):
                store['s%d_%s' % (t, k)] = s[k].detach().numpy()
            store['s%d_gmap_embeds' % t] = s['gmap_embeds'][:, :, :16].detach().numpy()
            store['s%d_vp_embeds' % t] = s['vp_embeds'][:, :, :16].detach().numpy()
            store['s%d_pano_fused' % t] = s['pano_fused'][:, :32].detach().numpy()
            if s.get('obj_logits') is not None:
                store['s%d_obj_logits' % t] = s['obj_logits'].detach().numpy()
        path = os.path.join(HERE, name + '.npz')
        if os.path.exists(path):       # a regeneration must reproduce what is committed, bit for bit (new arrays may be added)
            old = dict(np.load(path, allow_pickle=False))
            for k, v in old.items():
                assert k in store and np.array_equal(np.asarray(store[k]), v), 'regenerated %s differs from the committed %s' % (k, path)
        np.savez_compressed(path, **store)
        print('wrote', path, os.path.getsize(path) // 1024, 'KiB', 'loss', float(loss), 'autocast loss', float(loss_ac),
              'autocast grad error: median %.4f max %.4f' % (float(np.median(store['grad_err_autocast'])), float(store['grad_err_autocast'].max())))


def extract_case():
    """mode='extract_cfp_features' (M/models/vilmodel_GOAT.py:884-927: the pass that builds the FACL dictionaries) and
    'instr_zdict_update' on an R2R batch; no interventions (the mode is run on a pre-trained model before fine-tuning)."""
    vg = ref_shim.import_nav()
    from vln_goat_amd import nav_model, synth
    args = SimpleNamespace(num_l_layers=2, num_x_layers=2, num_pano_layers=2, dropout=0.5, feat_dropout=0.4,
                           do_back_img=False, do_back_txt=False, do_front_img=False, do_front_his=False, do_front_txt=False,
                           vocab_size=VOCAB, mode='extract_cfp_features')
    cfg = nav_model.nav_config_from_args(args)
    torch.manual_seed(0)
    ref = vg.GlocalTextPathNavCMT(cfg)
    ours = nav_model.GlocalTextPathNavCMT(cfg)
    sd = synth.seeded_state_dict(ours, seed=WEIGHT_SEED)
    rk, ok = set(ref.state_dict().keys()), set(sd.keys())
    assert rk == ok, (sorted(rk - ok)[:8], sorted(ok - rk)[:8])
    ref.load_state_dict(sd)
    ref.eval()
    batch = synth.make_pretrain_batch(B=3, T=[2, 4, 1], L=[30, 21, 12], seed=13, vocab_size=VOCAB, style='rich', ragged_views=True)
    batch['txt_masks'] = torch.arange(batch['txt_ids'].shape[1])[None, :] < batch['txt_lens'][:, None]
    from collections import defaultdict
    with torch.no_grad():
        out = ref('extract_cfp_features', defaultdict(lambda: None, batch))
        z = ref('instr_zdict_update', defaultdict(lambda: None, {'z_txt': batch['txt_ids'], 'z_txt_mask': batch['txt_masks']}))
    store = {k: v.numpy() for k, v in out.items()}
    store['zdict_txt'] = z[:, :, :32].numpy()
    path = os.path.join(HERE, 'nav_extract_cfp.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    if sys.argv[1:] == ['nav_extract_cfp']:
        extract_case()
    else:
        main(sys.argv[1:] or None)
        if not sys.argv[1:]:
            extract_case()
