"""GOAT fine-tuning (navigation) model on the HIP kernels — drop-in for the reference's
`models.model.VLNBert` / `Critic` (M/models/model.py:12,40) and `GlocalTextPathNavCMT`
(M/models/vilmodel_GOAT.py:556) with BACL (back-door: dictionary-weighted / cross-attended intervention) and
FACL (front-door: attention over K-means CFP features) — SURVEY.md §8a rows a-3, a-4, a-14, a-15, a-16, a-19.

`VLNBert.forward(mode, batch)` with mode in {language, panorama, navigation, instr_zdict_update,
extract_cfp_features}; return types and parameter names/shapes as the reference (M/models/vilmodel_GOAT.py:847-927).
R2R / RxR branch (REVERIE/SOON object tokens: not built yet).
"""
import collections

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import graphmap, hipops
from .layers import (BertAttention, BertLayerNorm, BertPooler, BertPredictionHeadTransform, ClsPrediction,
                     CrossmodalEncoder, Linear, RobertaAttention, RobertaEmbeddings, RobertaLayer, _p, compute_dtype, project_kv_bank,
                     create_transformer_encoder, gen_seq_masks, neg_mask)
from .pretrain_model import GoatPreTrainedModel, attn_pool


def _door(aug_lin, ori_lin, aug, ori):
    """door gate: w = sigmoid(Linear_a(aug) + Linear_o(ori)); out = w*aug + (1-w)*ori
    (M/models/vilmodel_GOAT.py:147-153, 548-552)."""
    return hipops.door_gate(aug_lin, ori_lin, aug, ori)


class LanguageEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_l_layers = config.num_l_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for _, p in self.layer.named_parameters():
                p.requires_grad = False

    def forward(self, txt_embeds, txt_masks, *unused):
        km = neg_mask(txt_masks)
        for i, layer in enumerate(self.layer):       # between layers the state travels as a layers._pair (fork=True)
            txt_embeds = layer(txt_embeds, km, fork=i + 1 < len(self.layer))
        return txt_embeds if self.update_lang_bert else txt_embeds.detach()


class LanguageEncoderDo(nn.Module):
    """M/models/vilmodel_GOAT.py:55-162 (BACL-txt type_1 / type_2, FACL-txt, door / add / concat)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_l_layers = config.num_l_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for _, p in self.layer.named_parameters():
                p.requires_grad = False
        H = config.hidden_size
        if config.do_back_txt or config.do_front_txt:
            self.z_txt_linear = Linear(H, H)
            self.z_direct_linear = Linear(H, H)
            self.z_landm_linear = Linear(H, H)
            self.z_concat_layernorm = BertLayerNorm(H, eps=config.layer_norm_eps)
            self.z_direct_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
            self.z_landm_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
            if config.do_back_txt_type == 'type_2':
                self.z_direc_cross_attn = RobertaAttention(config)
                self.z_landm_cross_attn = RobertaAttention(config)
                self.instr_aug_linear = Linear(H, 1)
                self.instr_ori_linear = Linear(H, 1)
                self.instr_sigmoid = nn.Sigmoid()
                self.concat_linear = Linear(H * 3, H)
        if config.do_front_txt:
            self.z_front_cross_attn = RobertaAttention(config)
            self.z_front_linear = Linear(H, H)
            self.z_front_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, txt_embeds, txt_masks, z_direc=None, z_direc_pzs=None, z_landm=None, z_landm_pzs=None, front_txt=None):
        cfg = self.config
        km = neg_mask(txt_masks)
        for i, layer in enumerate(self.layer):       # between layers the state travels as a layers._pair (fork=True)
            txt_embeds = layer(txt_embeds, km, fork=i + 1 < len(self.layer))
        if not self.update_lang_bert:
            txt_embeds = txt_embeds.detach()
        if not (cfg.do_back_txt or cfg.do_front_txt):
            return txt_embeds
        dt = txt_embeds.dtype
        z_front = None
        if cfg.do_back_txt_type == 'type_1':
            if cfg.do_back_txt:
                sd = hipops.dict_weighted_sum(z_direc, z_direc_pzs, dt)
                sl = hipops.dict_weighted_sum(z_landm, z_landm_pzs, dt)
                txt_embeds = self.z_txt_linear(txt_embeds) + self.z_direct_linear(sd) + self.z_landm_linear(sl)
            if cfg.do_front_txt and front_txt is not None:
                zf = self.z_front_cross_attn(txt_embeds, None, front_txt.to(dt), None)
                txt_embeds = txt_embeds + self.z_front_ln(self.z_front_linear(zf))
            return self.z_concat_layernorm(txt_embeds)
        # type_2: cross-attention of the text over each confounder dictionary (no key mask)
        zd = zl = None
        if cfg.do_back_txt:
            zd = self.z_direct_ln(self.z_direct_linear(self.z_direc_cross_attn(txt_embeds, None, z_direc.to(dt), None)))
            if z_landm is not None:
                zl = self.z_landm_ln(self.z_landm_linear(self.z_landm_cross_attn(txt_embeds, None, z_landm.to(dt), None)))
        if cfg.do_front_txt and front_txt is not None:
            z_front = self.z_front_ln(self.z_front_linear(self.z_front_cross_attn(txt_embeds, None, front_txt.to(dt), None)))
        if cfg.do_add_method == 'door':
            aug = None
            if cfg.do_back_txt:
                aug = zd
                if zl is not None:
                    aug = aug + zl
                if front_txt is not None:
                    aug = aug + z_front
            elif cfg.do_front_txt and front_txt is not None:
                aug = z_front
            txt_embeds = _door(self.instr_aug_linear, self.instr_ori_linear, aug, txt_embeds)
        elif cfg.do_add_method == 'add':
            if cfg.do_back_txt:
                txt_embeds = txt_embeds + zd + zl
            if cfg.do_front_txt and front_txt is not None:
                txt_embeds = txt_embeds + z_front
        elif cfg.do_add_method == 'concat':
            txt_embeds = self.concat_linear(torch.cat((txt_embeds, zd, zl), -1))
        return self.z_concat_layernorm(txt_embeds)


class CausalImageEmbeddings(nn.Module):
    """M/models/vilmodel_GOAT.py:164-316: R2R/RxR branch and the REVERIE/SOON branch (object tokens appended to the
    panorama, M:693-720)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.reverie = config.name in ('REVERIE', 'SOON')
        H = config.hidden_size
        self.img_linear = Linear(config.image_feat_size, H)
        self.img_layer_norm = BertLayerNorm(H, eps=1e-12)
        self.loc_linear = Linear(config.angle_feat_size + 3, H)
        self.loc_layer_norm = BertLayerNorm(H, eps=1e-12)
        if not self.reverie:
            self.img_self_encoder = create_transformer_encoder(config, config.num_pano_layers, norm=True)
        self.do_back_img = config.do_back_img
        if self.do_back_img:
            self.do_img_before_linear = Linear(config.image_feat_size, H)
            self.do_img_layer_norm = BertLayerNorm(H, eps=1e-12)
            self.do_img_attn = BertAttention(config)
            self.do_img_after_linear = Linear(H, H)
            self.img_after_linear = Linear(H, H)
            self.do_img_concat_layernorm = BertLayerNorm(H, eps=1e-12)
            if config.do_back_img_type == 'type_2':
                if config.do_add_method == 'door':
                    self.sigmoid = nn.Sigmoid()
                elif config.do_add_method == 'concat':
                    self.do_concat_img_linear = Linear(H * 2, H)
        if self.reverie:
            if config.use_obj_name:
                self.obj_name_linear = nn.Embedding(config.obj_name_vocab_size, H)
            self.obj_reverie_linear = Linear(config.obj_feat_size, H)
            self.obj_reverie_layer_norm = BertLayerNorm(H, eps=1e-12)
            self.nav_type_embedding = nn.Embedding(3, H)
            self.pano_encoder = create_transformer_encoder(config, config.num_pano_layers, norm=True)
        else:
            self.nav_type_embedding = nn.Embedding(2, H)
        if config.adaptive_pano_fusion:
            self.adaptive_pano_attn = Linear(H, 1)
        self.layer_norm = BertLayerNorm(H, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def intervene(self, x, z_img_features, z_img_pzs):
        """BACL-img (M/models/vilmodel_GOAT.py:659-681)."""
        cfg = self.config
        dt = x.dtype
        z = self.do_img_layer_norm(self.do_img_before_linear(z_img_features.to(dt)))
        if cfg.do_back_img_type == 'type_1':
            s = hipops.dict_weighted_sum(z, z_img_pzs, dt)
            x = self.img_after_linear(x) + self.do_img_after_linear(s)
        else:
            z = self.do_img_attn(x, None, z, None)
            if cfg.do_add_method == 'door':
                w = torch.sigmoid(self.img_after_linear(x).float() + self.do_img_after_linear(z).float()).to(dt)
                x = w * x + (1 - w) * z
            elif cfg.do_add_method == 'add':
                x = x + z
            elif cfg.do_add_method == 'concat':
                x = self.do_concat_img_linear(torch.cat((x, z), -1))
        return self.do_img_concat_layernorm(x)

    def encode(self, view_img_fts, loc_fts, view_lens, z_img_features=None, z_img_pzs=None, loc_before=False,
               nav_types=None, obj_fts=None, obj_lens=None, obj_names=None, obj_concat=None):
        """-> (embeds [N,W,H], masks [N,W] bool, fused [N,H] | None).  `loc_before` = pre-training order
        (location added before the intervention, M:225-252); per-step navigation adds it after (M:688-691).
        REVERIE/SOON (M:693-720): object tokens follow the views of every row; loc_fts / nav_types are [N,W,...]."""
        dt = compute_dtype()
        x = self.img_layer_norm(self.img_linear(view_img_fts.to(dt)))
        if self.reverie:
            if z_img_features is not None:
                x = self.intervene(x, z_img_features, z_img_pzs)
            o = self.obj_reverie_linear(obj_fts.to(dt))
            if self.config.use_obj_name:
                o = o + hipops.embedding(obj_names, self.obj_name_linear.weight, out_dtype=dt)
            o = self.obj_reverie_layer_norm(o)
            N, V, H = x.shape
            W = nav_types.shape[1]
            src = torch.cat([x.reshape(N * V, H), o.reshape(-1, H)], 0)
            if obj_concat is not None:      # (idx, start, inv_idx, inv_start) already on the device: shape-stable callers (captured episodes:
                x = hipops.gather_segmean(src, obj_concat[0], obj_concat[1], None, N * W, tuple(obj_concat[2:4])).view(N, W, H)      # no host read of the lengths)
            else:
                ci = graphmap.build_obj_concat_index(view_lens, obj_lens, V, o.shape[1], W)
                x = hipops.gather_segmean(src, ci[0].to(x.device), ci[1].to(x.device), None, N * W).view(N, W, H)
            x = x + self.loc_layer_norm(self.loc_linear(loc_fts.to(dt))) \
                + hipops.embedding(nav_types, self.nav_type_embedding.weight, out_dtype=dt)
            x = self.layer_norm(x, p_out=_p(self.dropout))
            masks = gen_seq_masks(view_lens + obj_lens, W)
            x = self.pano_encoder(x, masks)
        else:
            loc_in = self.loc_linear(loc_fts.to(dt))
            if loc_before:
                x = x + self.loc_layer_norm(loc_in)
            if z_img_features is not None:
                x = self.intervene(x, z_img_features, z_img_pzs)
            if not loc_before:       # dropout(x + loc_LN(...)): the sum and the dropout inside the LayerNorm's launch
                x = self.loc_layer_norm(loc_in, post_add=x, p_out=_p(self.dropout))
            else:
                x = hipops.dropout(x, _p(self.dropout))
            masks = gen_seq_masks(view_lens, view_img_fts.shape[1])
            x = self.img_self_encoder(x, masks)
        fused = None
        if self.config.adaptive_pano_fusion:
            fused = hipops.pano_fusion(x, self.adaptive_pano_attn.weight, self.adaptive_pano_attn.bias)
        return x, masks, fused


class LocalVPEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.vp_pos_embeddings = nn.Sequential(Linear(config.angle_feat_size * 2 + 6, config.hidden_size),
                                               BertLayerNorm(config.hidden_size, eps=1e-12))
        self.encoder = CrossmodalEncoder(config, with_lang_branch=False)
        if config.mode == 'extract_cfp_features':
            self.tim_self_encoder = BertAttention(config)


class GlobalMapEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.gmap_pos_embeddings = nn.Sequential(Linear(config.angle_feat_size + 3, config.hidden_size),
                                                 BertLayerNorm(config.hidden_size, eps=1e-12))
        self.gmap_step_embeddings = nn.Embedding(config.max_action_steps, config.hidden_size)
        self.encoder = CrossmodalEncoder(config, with_lang_branch=False)
        self.sprel_linear = Linear(1, 1) if config.graph_sprels else None
        if config.mode == 'extract_cfp_features':
            self.tim_self_encoder = BertAttention(config)


class FrontDoorEncoder(nn.Module):
    """FACL for view / history tokens (M/models/vilmodel_GOAT.py:526-554)."""

    def __init__(self, config):
        super().__init__()
        self.ll_self_attn = BertAttention(config)
        self.lg_cross_attn = BertAttention(config)
        self.ln = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.aug_linear = Linear(config.hidden_size, 1)
        self.ori_linear = Linear(config.hidden_size, 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, local_feats, global_feats, local_masks=None):
        km = neg_mask(local_masks) if local_masks is not None else None
        # the tokens are read five times (projection + residual of either attention block, the gate's pass-through input): one autograd
        # handle each, their gradients meet in one launch (hipops.fanout)
        h = hipops.fanout(local_feats, 5)
        ll = self.ll_self_attn((h[0], h[1]), km)
        lg = self.lg_cross_attn((h[2], h[3]), None, global_feats.to(local_feats.dtype), None)
        out = self.ln(ll, residual=lg)
        return _door(self.aug_linear, self.ori_linear, out, h[4])


class GlocalTextPathNavCMT(GoatPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        H = config.hidden_size
        self.embeddings = RobertaEmbeddings(config)
        self.lang_encoder = LanguageEncoderDo(config) if (config.do_back_txt or config.do_front_txt) else LanguageEncoder(config)
        self.img_embeddings = CausalImageEmbeddings(config)
        self.local_encoder = LocalVPEncoder(config)
        self.global_encoder = GlobalMapEncoder(config)
        self.global_sap_head = ClsPrediction(H)
        self.local_sap_head = ClsPrediction(H)
        self.sap_fuse_linear = ClsPrediction(H, input_size=H * 2) if config.glocal_fuse else None
        if getattr(config, 'obj_feat_size', 0) > 0:
            self.og_head = ClsPrediction(H)
        self.object_encoder = None
        self.extra_drop = nn.Dropout(0.2)
        self.gmap_pooler = BertPooler(config)
        self.vp_pooler = BertPooler(config)
        self.txt_pooler = BertPooler(config)
        self.local_his_map = Linear(H * 3, H)
        self.local_his_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
        self.drop_env = nn.Dropout(p=config.feat_dropout)
        cfp = config.mode == 'extract_cfp_features'
        if cfp:
            self.tim_local_head = BertPredictionHeadTransform(config)
            self.tim_local_attn = nn.Parameter(torch.empty(H, 1).uniform_(-0.1, 0.1))
            self.temperature = config.cfp_temperature
        if config.do_front_img:
            self.front_local_encoder = FrontDoorEncoder(config)
        if cfp:
            self.tim_global_head = BertPredictionHeadTransform(config)
            self.tim_global_attn = nn.Parameter(torch.empty(H, 1).uniform_(-0.1, 0.1))
        if config.do_front_his:
            self.front_global_encoder = FrontDoorEncoder(config)
        if cfp:
            self.tim_txt_head = BertPredictionHeadTransform(config)
            self.tim_txt_attn = nn.Parameter(torch.empty(H, 1).uniform_(-0.1, 0.1))
        if config.do_front_txt:
            self.front_txt_encoder = FrontDoorEncoder(config)
        self.init_weights()
        if config.fix_lang_embedding or config.fix_local_branch:
            for m in (self.embeddings, self.lang_encoder):
                for _, v in m.named_parameters():
                    v.requires_grad = False
        if config.fix_pano_embedding or config.fix_local_branch:
            for _, v in self.img_embeddings.named_parameters():
                v.requires_grad = False
        if config.fix_local_branch:
            for m in (self.local_encoder, self.local_sap_head):
                for _, v in m.named_parameters():
                    v.requires_grad = False

    # ---- language ---------------------------------------------------------------------------------
    def forward_text(self, txt_ids, txt_masks, z_direc=None, z_direc_pzs=None, z_landm=None, z_landm_pzs=None, front_txt=None):
        e = self.embeddings(txt_ids)
        return self.lang_encoder(e, txt_masks, z_direc, z_direc_pzs, z_landm, z_landm_pzs, front_txt)

    # ---- panorama -------------------------------------------------------------------------------------
    def forward_panorama_do_per_step(self, view_img_fts, loc_fts, nav_types, view_lens, z_img_features=None, z_img_pzs=None,
                                     reverie_obj_fts=None, reverie_obj_lens=None, reverie_obj_names=None, reverie_obj_concat=None):
        return self.img_embeddings.encode(view_img_fts, loc_fts, view_lens, z_img_features, z_img_pzs, loc_before=False,
                                          nav_types=nav_types, obj_fts=reverie_obj_fts, obj_lens=reverie_obj_lens,
                                          obj_names=reverie_obj_names, obj_concat=reverie_obj_concat)

    # ---- navigation -------------------------------------------------------------------------------------
    def forward_navigation_per_step(self, txt_embeds, txt_masks, gmap_img_embeds, gmap_step_ids, gmap_pos_fts, gmap_masks,
                                    gmap_pair_dists, gmap_visited_masks, gmap_vpids, vp_img_embeds, vp_pos_fts, vp_masks,
                                    vp_nav_masks, vp_obj_masks, vp_cand_vpids, front_vp_feats=None, front_gmap_feats=None,
                                    flops_count=False, nav_fusion=None, txt_kv=None):
        """txt_kv: optional `self.text_kv(txt_embeds)` — the instruction is constant over an episode, so the K|V projections the
        six cross-modal layers apply to it (M/models/vilmodel_GOAT.py:739-839 recomputes them in every step) can be computed once;
        identical outputs, the gradients of the steps are summed by autograd.
        nav_fusion: optional precomputed `nav_fusion_matrix(...)` on the device ([B, G, W] float32) — the agent knows the
        id strings and the visited flags on the host when it collates the step, so the call itself needs no device -> host
        read (and can be captured into a hipGraph)."""
        dt = compute_dtype()
        txt_embeds = txt_embeds.to(dt)
        txt_masks = txt_masks.bool() if txt_masks.dtype != torch.bool else txt_masks
        txt_km = neg_mask(txt_masks)
        B, G = gmap_step_ids.shape
        ge = self.global_encoder
        # (image embedding + step embedding) + LayerNorm(Linear(position features)): the second sum rides in the LayerNorm launch (post_add)
        ie = gmap_img_embeds.to(dt) + hipops.embedding(gmap_step_ids, ge.gmap_step_embeddings.weight, out_dtype=dt)
        gmap = ge.gmap_pos_embeddings[1](ge.gmap_pos_embeddings[0](gmap_pos_fts.to(dt)), post_add=ie)
        bias = None
        if ge.sprel_linear is not None:
            bias = gmap_pair_dists.float() * ge.sprel_linear.weight.view(()) + ge.sprel_linear.bias.view(())
        # the global-map branch (FACL front-door block, cross-modal encoder, its action head and pooler) is independent of the local
        # one until the logit fusion: a parallel branch of the captured step / episode graph, as in the pre-training model
        # (hipops.Branch; both chains are launch-latency-bound at 12 x 60 / 12 x 38 rows)
        with hipops.Branch('global', 'nav_step') as bg:
            if front_gmap_feats is not None:
                gmap = self.front_global_encoder(gmap, front_gmap_feats, gmap_masks)
            gmap = ge.encoder(gmap, neg_mask(gmap_masks), txt_embeds, txt_km, bias, kv_cache=None if txt_kv is None else txt_kv['global'])
            # the encoder output feeds the action head and — through its [CLS] row — the pooler and the fusion logit: two autograd handles
            # (their gradients meet in one launch), the [CLS] row selected once for both of its readers
            gmap_h, gmap_c = hipops.fanout(gmap, 2)
            g_cls = gmap_c[:, 0]
            g_scores = self.global_sap_head(gmap_h).squeeze(2)
            g_pool = torch.tanh(self.gmap_pooler.dense(g_cls))

        le = self.local_encoder
        vp = le.vp_pos_embeddings[1](le.vp_pos_embeddings[0](vp_pos_fts.to(dt)), post_add=vp_img_embeds.to(dt))
        if front_vp_feats is not None:
            vp = self.front_local_encoder(vp, front_vp_feats, vp_masks)
        vp = le.encoder(vp, neg_mask(vp_masks), txt_embeds, txt_km, kv_cache=None if txt_kv is None else txt_kv['local'])

        # scores of the two heads -> masked global / local / fused logits in one launch per direction (hipops.sap_fuse; the
        # reference's chain: M/models/vilmodel_GOAT.py:803-839).  The stop column of the fused logits takes the local stop logit.
        n_vp = 3 if (vp_obj_masks is not None and getattr(self.config, 'dataset', 'r2r') in ('reverie', 'soon')) else 2
        vp_hs = hipops.fanout(vp, n_vp)
        v_cls = vp_hs[1][:, 0]
        l_scores = self.local_sap_head(vp_hs[0]).squeeze(2)
        bg.join(gmap, g_scores, g_pool, g_cls)
        fwl = None if self.sap_fuse_linear is None else self.sap_fuse_linear(torch.cat([g_cls, v_cls], 1))
        M = None
        if not flops_count:
            M = nav_fusion if nav_fusion is not None else \
                nav_fusion_matrix(vp_cand_vpids, gmap_vpids, gmap_visited_masks, G, vp.shape[1]).to(gmap.device)
        gl, ll, fused, _ = hipops.sap_fuse(g_scores, l_scores, fwl,
                                           gvis=gmap_visited_masks, gvalid=gmap_masks, lmask=vp_nav_masks, lmask_is_valid=True, M=M,
                                           add_stop=True)
        obj_logits = None
        if vp_obj_masks is not None and getattr(self.config, 'dataset', 'r2r') in ('reverie', 'soon'):
            obj_logits = torch.where(vp_obj_masks, self.og_head(vp_hs[2]).squeeze(2).float(), -float('inf'))      # (no clone: see pretrain_model.forward_og)
        cls = torch.cat((g_pool, torch.tanh(self.vp_pooler.dense(v_cls)), self.txt_pooler(txt_embeds)), dim=-1)
        cls_embeds = self.local_his_ln(self.local_his_map(cls))
        return {'gmap_embeds': gmap, 'vp_embeds': vp, 'global_logits': gl, 'local_logits': ll, 'fused_logits': fused,
                'obj_logits': obj_logits, 'txt_embeds': txt_embeds, 'cls_embeds': cls_embeds}

    def text_kv(self, txt_embeds):
        """per-episode K|V projections of the instruction for the cross-modal layers of the global and the local branch."""
        t = txt_embeds.to(compute_dtype())
        kg, kl = project_kv_bank(t, [self.global_encoder.encoder, self.local_encoder.encoder])      # one GEMM for the twelve projections
        return {'global': kg, 'local': kl}

    # ---- CFP feature extraction (builds the FACL dictionaries) -------------------------------------------------
    def extract_cfp_features(self, batch):
        if self.img_embeddings.reverie:
            # upstream binds traj_reverie_obj_locs to the z_img_features argument in this mode (M:889-893) — unusable as shipped
            raise NotImplementedError('extract_cfp_features on REVERIE/SOON is broken in the reference (argument mismatch)')
        txt = self.forward_text(batch['txt_ids'], batch['txt_masks'])
        x, _, fused = self.img_embeddings.encode(batch['traj_view_img_fts'], batch['traj_loc_fts'], batch['traj_vp_view_lens'],
                                                 loc_before=True)
        N, V, H = x.shape
        rows = x.view(N * V, H)
        src = torch.cat([rows, fused], 0) if fused is not None else rows
        dev = x.device
        lens_cpu = batch['traj_vp_view_lens'].cpu()
        G = batch['gmap_step_ids'].shape[1]
        gi = graphmap.build_gmap_index(batch['traj_step_lens'], lens_cpu, batch['traj_vpids'], batch['traj_cand_vpids'],
                                       batch['gmap_vpids'], G, V, fused is not None)
        vi = graphmap.build_vp_index(batch['traj_step_lens'], lens_cpu, V)
        B = batch['gmap_step_ids'].shape[0]
        ge, le = self.global_encoder, self.local_encoder
        gimg = hipops.gather_segmean(src, gi[0].to(dev), gi[1].to(dev), gi[2].to(dev), B * G).view(B, G, H)
        gmap = gimg + hipops.embedding(batch['gmap_step_ids'], ge.gmap_step_embeddings.weight, out_dtype=x.dtype) \
            + ge.gmap_pos_embeddings[1](ge.gmap_pos_embeddings[0](batch['gmap_pos_fts'].to(x.dtype)))
        gmap = ge.tim_self_encoder(gmap, neg_mask(gen_seq_masks(batch['gmap_lens'], G)))
        W = vi[3]
        vimg = hipops.gather_segmean(x, vi[0].to(dev), vi[1].to(dev), None, B * W).view(B, W, H)
        vp = vimg + le.vp_pos_embeddings[1](le.vp_pos_embeddings[0](batch['vp_pos_fts'][:, :W].to(x.dtype)))
        vp = le.tim_self_encoder(vp, neg_mask(gen_seq_masks(vi[2].to(dev), W)))
        gmap, vp, txt = self.tim_global_head(gmap), self.tim_local_head(vp), self.tim_txt_head(txt)
        return {'txt_outputs': attn_pool(txt, self.tim_txt_attn), 'vp_outputs': attn_pool(vp, self.tim_local_attn),
                'gmap_outputs': attn_pool(gmap, self.tim_global_attn)}

    def forward(self, mode, batch, **kwargs):
        if mode == 'language':
            return self.forward_text(batch['txt_ids'], batch['txt_masks'], batch['instr_z_direction_features'],
                                     batch['instr_z_direction_pzs'], batch['instr_z_landmark_features'],
                                     batch['instr_z_landmark_pzs'], batch['front_txt_feats'])
        if mode == 'panorama':
            return self.forward_panorama_do_per_step(batch['view_img_fts'], batch['loc_fts'], batch['nav_types'],
                                                     batch['view_lens'], batch['z_img_features'], batch['z_img_pzs'],
                                                     batch['reverie_obj_img_fts'], batch['reverie_obj_lens'],
                                                     batch['reverie_obj_names'], batch['reverie_obj_concat'])
        if mode == 'navigation':
            return self.forward_navigation_per_step(
                batch['txt_embeds'], batch['txt_masks'], batch['gmap_img_embeds'], batch['gmap_step_ids'], batch['gmap_pos_fts'],
                batch['gmap_masks'], batch['gmap_pair_dists'], batch['gmap_visited_masks'], batch['gmap_vpids'],
                batch['vp_img_embeds'], batch['vp_pos_fts'], batch['vp_masks'], batch['vp_nav_masks'], batch['vp_obj_masks'],
                batch['vp_cand_vpids'], batch['front_vp_feats'], batch['front_gmap_feats'], flops_count=batch['flops_count'],
                nav_fusion=batch.get('nav_fusion'), txt_kv=batch.get('txt_kv'))
        if mode == 'text_kv':               # (build-side extension: see forward_navigation_per_step)
            return self.text_kv(batch['txt_embeds'])
        if mode == 'instr_zdict_update':
            return self.forward_text(batch['z_txt'], batch['z_txt_mask'], batch['instr_z_direction_features'],
                                     batch['instr_z_direction_pzs'], batch['instr_z_landmark_features'],
                                     batch['instr_z_landmark_pzs'], batch['front_txt_feats'])
        if mode == 'extract_cfp_features':
            return self.extract_cfp_features(batch)
        raise ValueError('invalid mode %r' % mode)


def nav_fusion_matrix(vp_cand_vpids, gmap_vpids, gmap_visited_masks, G, W):
    """fused[b,g] += sum_j M[b,g,j]*local[b,j] — the host loop of M/models/vilmodel_GOAT.py:790-806
    ([stop] and [MEM] slots, j <= 1, are skipped on both sides)."""
    vis = gmap_visited_masks.detach().cpu().tolist()
    B = len(gmap_vpids)
    M = np.zeros((B, G, W), dtype=np.float32)
    for b in range(B):
        visited = set(vp for vp, m in zip(gmap_vpids[b], vis[b]) if m)
        tmp, bw = {}, []
        for j, c in enumerate(vp_cand_vpids[b]):
            if j > 1:
                if c in visited:
                    bw.append(j)
                else:
                    tmp[c] = j
        for g, vp in enumerate(gmap_vpids[b]):
            if g > 1 and vp not in visited:
                if vp in tmp:
                    M[b, g, tmp[vp]] += 1.0
                else:
                    for j in bw:
                        M[b, g, j] += 1.0
    return torch.from_numpy(M)


_nav_fusion_matrix = nav_fusion_matrix


class VLNBert(nn.Module):
    """M/models/model.py:12-38: environment dropout on raw view features + mode dispatch."""

    def __init__(self, args, config=None):
        super().__init__()
        self.args = args
        self.vln_bert = get_vlnbert_models(args, config=config)
        self.drop_env = nn.Dropout(p=args.feat_dropout)

    def forward(self, mode, batch):
        batch = collections.defaultdict(lambda: None, batch)
        if mode == 'panorama':
            if not batch['already_dropout']:
                batch['view_img_fts'] = hipops.dropout(batch['view_img_fts'].to(compute_dtype()).contiguous(), _p(self.drop_env))
            if 'reverie_obj_img_fts' in batch and batch['reverie_obj_img_fts'] is not None:
                # object features are dropped regardless of `already_dropout` (M/models/model.py:31-32)
                batch['reverie_obj_img_fts'] = hipops.dropout(batch['reverie_obj_img_fts'].to(compute_dtype()).contiguous(),
                                                              _p(self.drop_env))
        return self.vln_bert(mode, batch)


class Critic(nn.Module):
    # M/models/model.py:40-50
    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(Linear(768, 512), nn.ReLU(), nn.Dropout(args.dropout), Linear(512, 1))

    def forward(self, state):
        h = self.state2value[0](state.to(compute_dtype()), act='relu')
        h = hipops.dropout(h, _p(self.state2value[2]))
        return self.state2value[3](h).float().squeeze()


def nav_config_from_args(args):
    """The config object `get_vlnbert_models` assembles (M/models/vlnbert_init.py:79-154) on top of the
    roberta config (layer_norm_eps 1e-5, pad_token_id 1)."""
    from types import SimpleNamespace
    g = lambda k, d: getattr(args, k, d)
    return SimpleNamespace(
        dataset=g('dataset', 'r2r'), mode=g('mode', 'train'), max_action_steps=100, image_feat_size=g('image_feat_size', 768),
        angle_feat_size=g('angle_feat_size', 4), obj_feat_size=g('obj_feat_size', 0), obj_loc_size=3, obj_name_vocab_size=45,
        num_l_layers=g('num_l_layers', 6), num_pano_layers=g('num_pano_layers', 2), num_x_layers=g('num_x_layers', 3),
        graph_sprels=g('graph_sprels', True), glocal_fuse=g('fusion', 'dynamic') == 'dynamic',
        fix_lang_embedding=g('fix_lang_embedding', False), fix_pano_embedding=g('fix_pano_embedding', False),
        fix_local_branch=g('fix_local_branch', False), update_lang_bert=not g('fix_lang_embedding', False),
        pred_head_dropout_prob=0.1, max_instr_len=g('max_instr_len', 200), feat_dropout=g('feat_dropout', 0.4),
        adaptive_pano_fusion=g('adaptive_pano_fusion', True), do_back_img=g('do_back_img', False),
        do_back_txt=g('do_back_txt', False), do_front_img=g('do_front_img', False), do_front_his=g('do_front_his', False),
        do_front_txt=g('do_front_txt', False), cfp_temperature=g('cfp_temperature', 1.0),
        do_back_txt_type=g('do_back_txt_type', 'type_2'), do_back_img_type=g('do_back_img_type', 'type_1'),
        do_add_method=g('do_add_method', 'door'), type_vocab_size=1, max_position_embeddings=514, vocab_size=g('vocab_size', 50265),
        num_top_layer=g('num_x_layers', 3), hidden_size=768, num_attention_heads=12, num_hidden_layers=g('num_l_layers', 6),
        hidden_dropout_prob=g('dropout', 0.5), attention_probs_dropout_prob=0.1, intermediate_size=3072, hidden_act='gelu',
        layer_norm_eps=1e-5, pad_token_id=1, initializer_range=0.02, is_decoder=False, add_cross_attention=False,
        chunk_size_feed_forward=0, use_lang2visn_attn=False,
        name={'reverie': 'REVERIE', 'soon': 'SOON'}.get(g('dataset', 'r2r'), 'R2R'), use_obj_name=g('dataset', 'r2r') == 'reverie')


def remap_pretrain_checkpoint(ckpt_weights):
    """Pre-train -> fine-tune key map (M/models/vlnbert_init.py:52-69): strip `module.`, `vln_bert` -> `bert`, heads /
    `sap_fuse` / `tim*` / `temperature` keys move under `bert.` (except the `*self_encoder*` ones).  The reference then loads
    the dictionary through HF `from_pretrained` of a model whose base_model_prefix is `bert`, which strips that prefix again:
    the keys returned here are the fine-tune model's own state_dict keys."""
    new = {}
    for k, v in ckpt_weights.items():
        if k.startswith('module'):
            k = k[7:]
        if k.startswith('vln_bert'):
            k = 'bert' + k[8:]
        if '_head' in k or 'sap_fuse' in k:
            new['bert.' + k] = v
        elif 'tim' in k or 'temperature' in k:
            new[k if 'self_encoder' in k else 'bert.' + k] = v
        else:
            new[k] = v
    return {(k[5:] if k.startswith('bert.') else k): v for k, v in new.items()}


def remap_bert_checkpoint(named_params):
    """`bert-base-uncased` initialisation (M/models/vlnbert_init.py:24-33): `bert.encoder.layer` -> `bert.lang_encoder.layer`."""
    new = {}
    for k, v in named_params.items():
        k = k.replace('bert.encoder.layer', 'bert.lang_encoder.layer')
        new[k[5:] if k.startswith('bert.') else k] = v
    return new


def remap_meter_checkpoint(state_dict):
    """METER initialisation (M/models/vlnbert_init.py:34-49; P/train_r2r_goat.py:153-171): text transformer -> embeddings +
    lang_encoder, `cross_modal_image_layers` -> the cross-attention stacks of BOTH the local and the global encoder."""
    new = {}
    for k, v in state_dict.items():
        if 'text_transformer.embeddings' in k:
            new[k.replace('text_transformer.', 'bert.')] = v
        elif 'text_transformer.encoder' in k:
            new[k.replace('text_transformer.encoder', 'bert.lang_encoder')] = v
        elif 'cross_modal_image_layers' in k:
            new[k.replace('cross_modal_image_layers', 'bert.local_encoder.encoder.crossattention')] = v
            new[k.replace('cross_modal_image_layers', 'bert.global_encoder.encoder.crossattention')] = v
        else:
            new[k] = v
    return {(k[5:] if k.startswith('bert.') else k): v for k, v in new.items()}


def get_vlnbert_models(args, config=None):
    cfg = config if config is not None else nav_config_from_args(args)
    sd = None
    path = getattr(args, 'bert_ckpt_file', None)
    if path == 'bert':
        raise RuntimeError("bert_ckpt_file='bert' needs the hub model (no network here): pass remap_bert_checkpoint(named_parameters) as state_dict")
    if path == 'meter':
        raw = torch.load('datasets/pretrained/METER/meter_clip16_224_roberta_pretrain.ckpt', map_location='cpu')['state_dict']
        sd = remap_meter_checkpoint(raw)
    elif path:
        raw = torch.load(path, map_location='cpu')
        sd = remap_pretrain_checkpoint(raw)
    return GlocalTextPathNavCMT.from_pretrained(None, config=cfg, state_dict=sd)
