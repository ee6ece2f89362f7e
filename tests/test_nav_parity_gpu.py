"""Fine-tuning path (VLNBert / GlocalTextPathNavCMT with BACL + FACL) on the GPU against golden vectors of the
imported reference (tests/golden/make_golden_nav.py): language -> (panorama -> navigation) x 3 with BPTT
through the [MEM] token.  fp32 path 1e-3, bf16 path 2e-2 on outputs."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import fingerprint, load_golden

pytestmark = pytest.mark.gpu

CASES = {
    'nav_type2_door': dict(do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door'),
    'nav_type1_add': dict(do_back_txt_type='type_1', do_back_img_type='type_2', do_add_method='add'),
    # REVERIE: object tokens in every panorama + object-grounding head (must match tests/golden/make_golden_nav.py)
    'nav_reverie_objects': dict(do_back_txt_type='type_2', do_back_img_type='type_1', do_add_method='door', dataset='reverie',
                                obj_feat_size=768),
}
EPISODE = {'nav_reverie_objects': dict(objects=5, seed=9)}


def _build(over, epkw=None):
    from vln_goat_amd import nav_model, synth
    args = SimpleNamespace(num_l_layers=2, num_x_layers=2, num_pano_layers=2, dropout=0.5, feat_dropout=0.4,
                           do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                           vocab_size=1200, mode='train', **over)
    cfg = nav_model.nav_config_from_args(args)
    model = nav_model.GlocalTextPathNavCMT(cfg)
    model.load_state_dict(synth.seeded_state_dict(model, seed=11))
    ep = synth.make_nav_episode(**{**dict(B=2, L=44, n_steps=3, seed=5, vocab_size=1200), **(epkw or {})})
    return model, ep


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', list(CASES))
def test_nav_episode_matches_reference_golden(case, dtype):
    import vln_goat_amd
    from vln_goat_amd import synth
    gold = load_golden(case)
    model, ep = _build(CASES[case], EPISODE.get(case))
    vln_goat_amd.set_compute_dtype(dtype)
    try:
        model = model.cuda().eval()
        for k in ('front_txt_feats', 'front_gmap_feats', 'z_img_features', 'instr_z_direction_features'):
            ep[k] = ep[k].cuda().requires_grad_(True)
        loss, rec = synth.run_nav_episode(lambda m, b: model(m, b), ep, device='cuda')
        loss.backward()
        torch.cuda.synchronize()
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    assert abs(float(loss) - float(gold['loss'][0])) / max(1.0, abs(float(gold['loss'][0]))) < tol
    for t, s in enumerate(rec['steps']):
        for k in ('global_logits', 'local_logits', 'fused_logits'):
            ref = gold['s%d_%s' % (t, k)]
            got = s[k].detach().float().cpu().numpy()
            assert np.array_equal(np.isinf(got), np.isinf(ref)), (t, k)
            m = ~np.isinf(ref)
            assert np.abs(got[m] - ref[m]).max() / max(1.0, np.abs(ref[m]).max()) < tol, (t, k)
        if ('s%d_obj_logits' % t) in gold:
            ref = gold['s%d_obj_logits' % t]
            got = s['obj_logits'].detach().float().cpu().numpy()
            assert np.array_equal(np.isinf(got), np.isinf(ref)), (t, 'obj_logits')
            m = ~np.isinf(ref)
            assert np.abs(got[m] - ref[m]).max() / max(1.0, np.abs(ref[m]).max()) < tol, (t, 'obj_logits')
        for k, sl in (('cls_embeds', None), ('gmap_embeds', 16), ('vp_embeds', 16), ('pano_fused', 32)):
            ref = gold['s%d_%s' % (t, k)]
            got = s[k].detach().float().cpu().numpy()
            if sl is not None:
                got = got[..., :sl]
            assert np.abs(got - ref).max() / np.abs(ref).max() < 2 * tol, (t, k)
    if dtype == torch.float32:
        names = [str(n) for n in gold['param_names']]
        params = dict(model.named_parameters())
        gmax = float(gold['grad_fp'][:, 0].max())
        for i, n in enumerate(names):
            ref = gold['grad_fp'][i]
            got = fingerprint(params[n].grad)
            if ref[0] <= 1e-6 * gmax:
                assert got[0] <= 1e-4 * gmax, n
                continue
            assert np.abs(got - ref).max() / ref[0] < 2e-3, (n, got, ref)
        for k in ('front_txt_feats', 'front_gmap_feats', 'z_img_features', 'instr_z_direction_features'):
            ref = gold['dinput_' + k]
            got = fingerprint(ep[k].grad)
            assert np.abs(got - ref).max() / max(ref[0], 1e-12) < 2e-3, k


def test_vlnbert_wrapper_and_critic_run():
    import vln_goat_amd
    from vln_goat_amd import nav_model, synth
    args = SimpleNamespace(num_l_layers=1, num_x_layers=1, num_pano_layers=1, dropout=0.5, feat_dropout=0.4,
                           do_back_img=True, do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True,
                           vocab_size=500, mode='train', do_back_txt_type='type_2', do_back_img_type='type_1',
                           do_add_method='door', bert_ckpt_file=None)
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        net = nav_model.VLNBert(args).cuda().train()
        critic = nav_model.Critic(args).cuda().train()
        ep = synth.make_nav_episode(B=2, L=30, n_steps=2, seed=1, vocab_size=500)
        loss, rec = synth.run_nav_episode(lambda m, b: net(m, dict(b)), ep, device='cuda')
        v = critic(rec['steps'][-1]['cls_embeds'])
        (loss + v.sum()).backward()
        assert torch.isfinite(loss)
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_extract_cfp_features_and_zdict_update_match_reference_golden(dtype):
    """mode='extract_cfp_features' (the pass that builds the FACL dictionaries, M/models/vilmodel_GOAT.py:884-927) and
    mode='instr_zdict_update' against the imported reference (tests/golden/make_golden_nav.py: extract_case)."""
    from collections import defaultdict
    import vln_goat_amd
    from vln_goat_amd import nav_model, synth
    gold = load_golden('nav_extract_cfp')
    args = SimpleNamespace(num_l_layers=2, num_x_layers=2, num_pano_layers=2, dropout=0.5, feat_dropout=0.4,
                           do_back_img=False, do_back_txt=False, do_front_img=False, do_front_his=False, do_front_txt=False,
                           vocab_size=1200, mode='extract_cfp_features')
    model = nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(args))
    model.load_state_dict(synth.seeded_state_dict(model, seed=11))
    batch = synth.make_pretrain_batch(B=3, T=[2, 4, 1], L=[30, 21, 12], seed=13, vocab_size=1200, style='rich', ragged_views=True)
    batch['txt_masks'] = torch.arange(batch['txt_ids'].shape[1])[None, :] < batch['txt_lens'][:, None]
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    vln_goat_amd.set_compute_dtype(dtype)
    try:
        model = model.cuda().eval()
        gb = synth.batch_to(batch, 'cuda')
        with torch.no_grad():
            out = model('extract_cfp_features', defaultdict(lambda: None, gb))
            z = model('instr_zdict_update', defaultdict(lambda: None, {'z_txt': gb['txt_ids'], 'z_txt_mask': gb['txt_masks']}))
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)
    for k in ('txt_outputs', 'vp_outputs', 'gmap_outputs'):
        got, ref = out[k].float().cpu().numpy(), gold[k]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() / max(1.0, np.abs(ref).max()) < tol, k
    got, ref = z.float().cpu().numpy()[:, :, :32], gold['zdict_txt']
    assert np.abs(got - ref).max() / max(1.0, np.abs(ref).max()) < tol
