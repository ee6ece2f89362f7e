"""Model configuration: the attributes the reference reads from its HF `PretrainedConfig`
(P/config/r2r_GOAT_model_config.json; attribute list in SURVEY.md §8b).  Any object with these
attributes works (a real transformers.PretrainedConfig drops in unchanged)."""
import json
from types import SimpleNamespace

R2R_GOAT_MODEL_CONFIG = dict(
    pred_head_dropout_prob=0.1, attention_probs_dropout_prob=0.1, hidden_act='gelu', hidden_dropout_prob=0.1,
    hidden_size=768, image_feat_size=768, image_prob_size=1000, angle_feat_size=4, obj_feat_size=0, obj_prob_size=0,
    initializer_range=0.02, intermediate_size=3072, num_l_layers=6, num_x_layers=3, num_top_layer=3,
    num_pano_layers=2, layer_norm_eps=1e-12, max_position_embeddings=514, max_action_steps=100,
    num_attention_heads=12, type_vocab_size=1, update_lang_bert=True, vocab_size=50265, use_lang2visn_attn=True,
    graph_sprels=True, glocal_fuse=True, adaptive_pano_fusion=True, cfp_extra_head=True, cfp_temperature=1.0,
    do_back_txt=False, do_back_img=False, do_back_txt_type='type_1', do_back_imgobj_type='type_1',
    do_add_method='add', do_front_img=False, do_front_his=False, do_front_txt=False, front_n_clusters=24,
    z_cross_attn=False, pad_token_id=None, is_decoder=False, add_cross_attention=False, chunk_size_feed_forward=0,
    name='R2R', empty_cache=False,
)


def make_config(json_path=None, **overrides):
    """Defaults = the shipped R2R pre-training model config; JSON file then keyword overrides on top."""
    d = dict(R2R_GOAT_MODEL_CONFIG)
    if json_path:
        with open(json_path) as f:
            d.update(json.load(f))
    d.update(overrides)
    if 'pretrain_tasks' not in d:
        d['pretrain_tasks'] = {'mlm', 'sap', 'cfp'}
    d['pretrain_tasks'] = set(d['pretrain_tasks'])
    return SimpleNamespace(**d)
