"""SURVEY §8 a-18 on the device: the pre-training step harness (vln_goat_amd.train_step.PretrainStep) drives the HIP model
with the gradient arena attached — forward, backward, clip_grad_norm_(5.0), AdamW — and the loss of a fixed batch falls."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('use_arena', [False, True])
def test_pretrain_step_trains_the_hip_model(use_arena):
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, dp, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=2, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        wrapper = dp.GoatDataParallel(model)
        if use_arena:
            for t in ('mlm', 'sap', 'cfp'):
                for p in model.parameters():
                    p.grad = None
                model(gb, t, True).mean().backward()
                wrapper.record_usage(t)
            for p in model.parameters():
                p.grad = None
            wrapper.build_arena()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
        step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
        first, last = {}, {}
        for it in range(8):
            for name in ('mlm', 'sap_r2r', 'cfp'):
                info = step(name, gb)
                assert info['updated'] and info['task'] == name.split('_')[0]
                assert info['grad_norm'] is not None and info['grad_norm'] == info['grad_norm']      # finite, not NaN
                assert info['n_loss_units'] > 0
                first.setdefault(name, info['loss'])
                last[name] = info['loss']
        torch.cuda.synchronize()
        assert step.global_step == 24
        for name in first:
            assert last[name] < first[name], (name, first[name], last[name])
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)
