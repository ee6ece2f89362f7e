"""Drop-in boundary of the nn.Module API against fixtures extracted from the reference models
(tests/golden/make_contract.py): parameter names / shapes / state_dict keys, the torch.nn.Dropout modules `set_dropout`
(P/utils/misc.py:19-25) iterates, the weight-decay grouping of P/optim/misc.py:13-23, the initialisation rule (a-17),
`from_pretrained` and the checkpoint key maps (M/models/vlnbert_init.py:24-69).  CPU only (module construction)."""
import json
import os
import warnings
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import ROOT

GOLD = os.path.join(ROOT, 'tests', 'golden')


def _contract(model):
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    return {'dropout': {n: float(m.p) for n, m in model.named_modules() if isinstance(m, torch.nn.Dropout)},
            'params': {n: list(p.shape) for n, p in model.named_parameters()},
            'no_decay': sorted(n for n, _ in model.named_parameters() if any(nd in n for nd in no_decay)),
            'state_dict_keys': sorted(model.state_dict().keys())}


def _pretrain_model(tag):
    from vln_goat_amd import config as gcfg, pretrain_model
    over = dict(name='REVERIE', obj_feat_size=768, obj_prob_size=1000, image_prob_size=1000, obj_name_vocab_size=45, use_obj_name=True,
                pretrain_tasks=['mlm', 'mrc', 'sap', 'og', 'cfp']) if tag == 'reverie' else {}
    torch.manual_seed(0)
    return pretrain_model.GlocalTextPathCMTPreTraining(gcfg.make_config(**over))


def _nav_model(tag):
    from vln_goat_amd import nav_model
    over = dict(dataset='reverie', obj_feat_size=768) if tag == 'reverie' else {}
    args = SimpleNamespace(num_l_layers=6, num_x_layers=3, num_pano_layers=2, dropout=0.1, feat_dropout=0.5, do_back_img=True,
                           do_back_txt=True, do_front_img=True, do_front_his=True, do_front_txt=True, do_back_txt_type='type_2',
                           do_back_img_type='type_1', do_add_method='door', mode='train', **over)
    torch.manual_seed(0)
    return nav_model.GlocalTextPathNavCMT(nav_model.nav_config_from_args(args))


@pytest.mark.parametrize('tree,tag', [('pretrain', 'r2r'), ('pretrain', 'reverie'), ('nav', 'r2r'), ('nav', 'reverie')])
def test_module_tree_contract_matches_reference(tree, tag):
    with open(os.path.join(GOLD, 'contract_%s.json' % tree)) as f:
        ref = json.load(f)[tag]
    got = _contract(_pretrain_model(tag) if tree == 'pretrain' else _nav_model(tag))
    assert got['params'] == ref['params']                          # names, order-independent, and shapes
    assert got['state_dict_keys'] == ref['state_dict_keys']
    assert got['no_decay'] == ref['no_decay']                       # weight-decay grouping by name (P/optim/misc.py:13-23)
    assert got['dropout'] == ref['dropout']                         # what set_dropout() finds, with the configured probabilities


def test_set_dropout_reaches_every_dropout_site():
    """The reference tunes dropout by mutating `.p` of the nn.Dropout modules; the HIP ops read the probability from those
    modules at call time (layers._p), so no site may keep a private copy."""
    import vln_goat_amd.layers as layers
    model = _pretrain_model('r2r')
    for _, m in model.named_modules():                              # P/utils/misc.py:19-25
        if isinstance(m, torch.nn.Dropout) and m.p != 0.3:
            m.p = 0.3
    model.train()
    drops = [m for m in model.modules() if isinstance(m, torch.nn.Dropout)]
    assert drops and all(layers._p(m) == 0.3 for m in drops)
    model.eval()
    assert all(layers._p(m) == 0.0 for m in drops)                  # eval(): dropout off, as F.dropout(training=False)


def test_initialisation_rule():
    """a-17: Linear / Embedding ~ N(0, initializer_range), biases 0, LayerNorm (1, 0), tim_*_attn ~ U(-0.1, 0.1), tied decoder."""
    with open(os.path.join(GOLD, 'contract_pretrain.json')) as f:
        ref = json.load(f)['init_stats']
    model = _pretrain_model('r2r')
    sd = model.state_dict()
    for k, (mean, std, lo, hi) in ref.items():
        v = sd[k].float()
        if 'LayerNorm' in k or k.endswith('.bias'):
            assert float(v.min()) == lo and float(v.max()) == hi, k      # exactly 1 / 0
        elif k.startswith('tim_'):
            assert -0.1 <= float(v.min()) and float(v.max()) <= 0.1 and abs(float(v.std()) - 0.2 / 12 ** 0.5) < 0.006, k
        else:
            assert abs(float(v.mean())) < 2e-3 and abs(float(v.std()) - std) < 1e-3, (k, float(v.std()), std)
    assert model.mlm_head.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight


def test_from_pretrained_reports_and_rejects():
    from vln_goat_amd import config as gcfg, pretrain_model
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300)
    src = pretrain_model.GlocalTextPathCMTPreTraining(cfg)
    sd = {k: v.clone() + 0.5 for k, v in src.state_dict().items()}
    sd['module_that_does_not_exist.weight'] = torch.zeros(3)
    sd['bert.embeddings.position_embeddings.weight'] = torch.zeros(7, 768)            # wrong shape: dropped, reported
    del sd['bert.lang_encoder.layer.0.attention.self.query.bias']
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m = pretrain_model.GlocalTextPathCMTPreTraining.from_pretrained(None, config=cfg, state_dict=sd)
    assert any('from_pretrained' in str(x.message) for x in w)
    rep = m.load_report
    assert rep['unexpected'] == ['module_that_does_not_exist.weight']
    assert rep['mismatched'] == ['bert.embeddings.position_embeddings.weight']
    assert 'bert.lang_encoder.layer.0.attention.self.query.bias' in rep['missing']
    got = m.state_dict()
    assert torch.equal(got['bert.lang_encoder.layer.0.attention.self.query.weight'], sd['bert.lang_encoder.layer.0.attention.self.query.weight'])
    assert got['bert.embeddings.position_embeddings.weight'].shape == (514, 768)       # kept at its initialisation
    assert m.mlm_head.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight    # re-tied after loading
    with pytest.raises(RuntimeError):                                                    # nothing matches: a wrong key map must not pass silently
        pretrain_model.GlocalTextPathCMTPreTraining.from_pretrained(None, config=cfg, state_dict={'vln_bert.x': torch.zeros(1)})


def test_checkpoint_key_maps():
    """M/models/vlnbert_init.py:52-69 (pre-train checkpoint), :24-33 (bert), :34-49 (METER), each followed by the `bert.`
    prefix strip of HF from_pretrained on a model whose base_model_prefix is 'bert'."""
    from vln_goat_amd import nav_model
    t = torch.zeros(1)
    table = {
        'module.bert.embeddings.word_embeddings.weight': 'embeddings.word_embeddings.weight',
        'vln_bert.lang_encoder.layer.0.output.dense.weight': 'lang_encoder.layer.0.output.dense.weight',
        'module.vln_bert.img_embeddings.img_linear.weight': 'img_embeddings.img_linear.weight',
        'mlm_head.predictions.bias': 'mlm_head.predictions.bias',                       # '_head' -> bert.<key> -> stripped again
        'global_sap_head.net.0.weight': 'global_sap_head.net.0.weight',
        'sap_fuse_linear.net.0.weight': 'sap_fuse_linear.net.0.weight',
        'tim_txt_attn': 'tim_txt_attn',
        'temperature': 'temperature',
        'bert.global_encoder.tim_self_encoder.self.query.weight': 'global_encoder.tim_self_encoder.self.query.weight',
        'bert.local_encoder.encoder.x_layers.0.visn_output.dense.bias': 'local_encoder.encoder.x_layers.0.visn_output.dense.bias',
    }
    out = nav_model.remap_pretrain_checkpoint({k: t for k in table})
    assert set(out) == set(table.values())
    for k, v in table.items():
        assert v in out, k
    bert = nav_model.remap_bert_checkpoint({'bert.encoder.layer.3.attention.self.key.weight': t, 'bert.embeddings.LayerNorm.weight': t,
                                            'bert.pooler.dense.weight': t})
    assert set(bert) == {'lang_encoder.layer.3.attention.self.key.weight', 'embeddings.LayerNorm.weight', 'pooler.dense.weight'}
    meter = nav_model.remap_meter_checkpoint({'text_transformer.embeddings.word_embeddings.weight': t,
                                              'text_transformer.encoder.layer.2.output.dense.weight': t,
                                              'cross_modal_image_layers.1.attention.self.query.weight': t, 'vit_model.x': t})
    assert set(meter) == {'embeddings.word_embeddings.weight', 'lang_encoder.layer.2.output.dense.weight',
                          'local_encoder.encoder.crossattention.1.attention.self.query.weight',
                          'global_encoder.encoder.crossattention.1.attention.self.query.weight', 'vit_model.x'}


def test_fast_gelu_polynomials_are_bf16_grade():
    """The epilogue polynomials of csrc/common.hpp (gelu_fast / dgelu_fast) against erf-GELU in float64: the bound quoted there."""
    import re
    from scipy.special import erf
    src = open(os.path.join(ROOT, 'vln-goat_amd', 'csrc', 'common.hpp')).read()

    def coefs(name):
        return [np.float32(float(re.search(r'#define GOAT_%s_C%d (\S+?)f\n' % (name, k), src).group(1))) for k in range(8)]

    def poly(c, x):
        xc = np.clip(x, -4.0, 4.0).astype(np.float32)
        t = xc * xc
        p = c[7]
        for k in range(6, -1, -1):
            p = (p * t + c[k]).astype(np.float32)
        return (xc * p).astype(np.float32) + np.float32(0.5)
    x = np.linspace(-10, 10, 400001)
    Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
    phi = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
    assert np.abs(x * poly(coefs('GELU'), x).astype(np.float64) - x * Phi).max() < 4.4e-4
    assert np.abs(poly(coefs('DGELU'), x).astype(np.float64) - (Phi + x * phi)).max() < 3e-4
