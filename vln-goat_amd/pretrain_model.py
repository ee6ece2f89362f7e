"""GOAT pre-training model on the HIP kernels — drop-in for the reference's
`model.pretrain_goat.GlocalTextPathCMTPreTraining` (P/model/pretrain_goat.py:40) and
`model.vilmodel_goat.GlocalTextPathCMT` (P/model/vilmodel_goat.py:529).

Same constructor (`config`), same `from_pretrained(None, config=, state_dict=)`, same
`forward(batch, task, compute_loss)` return values, same parameter names/shapes (state_dict contract,
including the parameters the reference creates but never trains).  Host-side Python loops over viewpoint
strings are replaced by per-batch index tensors (graphmap.py) + one gather kernel.
"""
from collections import defaultdict

import torch
import torch.nn.functional as F
from torch import nn

from . import graphmap, hipops
from .layers import (BertAttention, BertLayerNorm, BertOnlyMLMHead, BertPredictionHeadTransform, ClsPrediction,
                     CrossmodalEncoder, LayerNorm, Linear, RegionClassification, RobertaEmbeddings, RobertaLayer, _p, compute_dtype, project_kv_bank,
                     create_transformer_encoder, gen_seq_masks, neg_mask)


class GoatPreTrainedModel(nn.Module):
    """The four behaviours of transformers' BertPreTrainedModel the reference relies on
    (P/train_r2r_goat.py:192-197, P/model/pretrain_goat.py:83-89)."""

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        std = getattr(self.config, 'initializer_range', 0.02)
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def init_weights(self):
        self.apply(self._init_weights)
        self.tie_weights()

    def tie_weights(self):
        pass

    def _tie_or_clone_weights(self, output_embeddings, input_embeddings):
        output_embeddings.weight = input_embeddings.weight

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config=None, state_dict=None, **kwargs):
        """HF `from_pretrained(None, config=, state_dict=)` as the reference calls it (P/train_r2r_goat.py:192-197,
        M/models/vlnbert_init.py:152): non-strict load; keys the model does not have and tensors of another shape are dropped.
        Unlike HF's logger.warning this is reported on the model (`model.load_report`) and through `warnings.warn`, and a
        state_dict from which NOTHING could be loaded raises — a wrong key map must not silently leave random weights."""
        import warnings
        model = cls(config)
        if state_dict:
            own = model.state_dict()
            keep = {k: v for k, v in state_dict.items() if k in own and own[k].shape == v.shape}
            rep = {'loaded': len(keep), 'unexpected': sorted(k for k in state_dict if k not in own),
                   'mismatched': sorted(k for k, v in state_dict.items() if k in own and own[k].shape != v.shape),
                   'missing': sorted(k for k in own if k not in keep)}
            model.load_report = rep
            if not keep:
                raise RuntimeError('from_pretrained: none of the %d checkpoint tensors matches the model (first keys: %s) — wrong key map?'
                                   % (len(state_dict), list(state_dict)[:3]))
            if rep['unexpected'] or rep['mismatched'] or rep['missing']:
                warnings.warn('from_pretrained: loaded %d tensors; %d unexpected, %d shape-mismatched (dropped), %d missing (kept at init)'
                              % (rep['loaded'], len(rep['unexpected']), len(rep['mismatched']), len(rep['missing'])))
            model.load_state_dict(keep, strict=False)
            model.tie_weights()
        return model


def _cfg(config, name, default):
    v = getattr(config, name, default)
    return default if v is None else v


class LanguageEncoder(nn.Module):
    # P/model/vilmodel_goat.py:24-44
    def __init__(self, config):
        super().__init__()
        self.num_l_layers = config.num_l_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for _, param in self.layer.named_parameters():
                param.requires_grad = False

    def forward(self, txt_embeds, txt_kmask):
        for i, layer in enumerate(self.layer):       # between layers the state travels as a layers._pair (fork=True)
            txt_embeds = layer(txt_embeds, txt_kmask, fork=i + 1 < len(self.layer))
        if not self.update_lang_bert:
            txt_embeds = txt_embeds.detach()
        return txt_embeds


def _door(aug_lin, ori_lin, aug, ori):
    """door gate: w = sigmoid(Linear_a(aug) + Linear_o(ori)); out = w*aug + (1-w)*ori (P/model/vilmodel_goat.py:137-143)."""
    return hipops.door_gate(aug_lin, ori_lin, aug, ori)


class LanguageEncoderDo(nn.Module):
    """BACL-txt in pre-training (P/model/vilmodel_goat.py:46-159): after the RoBERTa layers the text is intervened
    with the direction / landmark confounder dictionaries — type_1: probability-weighted dictionary sums through
    three Linears (optionally dictionary->text cross-attention first, z_cross_attn); type_2: text->dictionary
    cross-attention, then door / add / concat — followed by LayerNorm.  Module set mirrors the reference's
    constructor (state_dict keys), including modules it creates and never calls (txt_self_attn, z_front_*)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_l_layers = config.num_l_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([RobertaLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for _, param in self.layer.named_parameters():
                param.requires_grad = False
        H = config.hidden_size
        if config.do_back_txt:
            if config.z_cross_attn:
                self.z_direc_cross_attn = BertAttention(config)
                self.z_landm_cross_attn = BertAttention(config)
            self.z_txt_linear = Linear(H, H)
            self.z_direct_linear = Linear(H, H)
            self.z_landm_linear = Linear(H, H)
            self.z_concat_layernorm = BertLayerNorm(H, eps=config.layer_norm_eps)
            self.z_direct_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
            self.z_landm_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
            if config.do_back_txt_type == 'type_2':
                self.z_direc_cross_attn = BertAttention(config)
                self.z_landm_cross_attn = BertAttention(config)
                self.txt_self_attn = BertAttention(config)
                self.instr_aug_linear = Linear(H, 1)
                self.instr_ori_linear = Linear(H, 1)
                self.instr_sigmoid = nn.Sigmoid()
                self.concat_linear = Linear(H * 3, H)
        if getattr(config, 'do_front_txt', False):
            self.z_front_cross_attn = BertAttention(config)
            self.z_front_linear = Linear(H, H)
            self.z_front_ln = BertLayerNorm(H, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, txt_embeds, txt_kmask, z_direc=None, z_direc_pzs=None, z_landm=None, z_landm_pzs=None):
        cfg = self.config
        for i, layer in enumerate(self.layer):       # between layers the state travels as a layers._pair (fork=True)
            txt_embeds = layer(txt_embeds, txt_kmask, fork=i + 1 < len(self.layer))
        if not self.update_lang_bert:
            txt_embeds = txt_embeds.detach()
        if z_direc is None:
            return txt_embeds
        dt = txt_embeds.dtype
        z_direc, z_landm = z_direc.to(dt), (z_landm.to(dt) if z_landm is not None else None)
        if cfg.do_back_txt_type == 'type_1':
            if cfg.z_cross_attn:       # dictionary entries attend to the (key-masked) text
                z_direc = self.z_direc_cross_attn(z_direc, None, txt_embeds, txt_kmask)
                z_landm = self.z_landm_cross_attn(z_landm, None, txt_embeds, txt_kmask)
            sd = hipops.dict_weighted_sum(z_direc, z_direc_pzs, dt)
            sl = hipops.dict_weighted_sum(z_landm, z_landm_pzs, dt)
            txt_embeds = self.z_txt_linear(txt_embeds) + self.z_direct_linear(sd) + self.z_landm_linear(sl)
            return self.z_concat_layernorm(txt_embeds)
        # type_2: the text attends to each dictionary (no key mask on dictionary entries)
        zd = self.z_direct_ln(self.z_direct_linear(self.z_direc_cross_attn(txt_embeds, None, z_direc, None)))
        zl = None
        if z_landm is not None:
            zl = self.z_landm_ln(self.z_landm_linear(self.z_landm_cross_attn(txt_embeds, None, z_landm, None)))
        if cfg.do_add_method == 'door':
            aug = zd if zl is None else zd + zl
            txt_embeds = _door(self.instr_aug_linear, self.instr_ori_linear, aug, txt_embeds)
        elif cfg.do_add_method == 'add':
            txt_embeds = txt_embeds + zd + zl
        elif cfg.do_add_method == 'concat':
            txt_embeds = self.concat_linear(torch.cat((txt_embeds, zd, zl), -1))
        return self.z_concat_layernorm(txt_embeds)


class CausalImageEmbeddings(nn.Module):
    """P/model/vilmodel_goat.py:234-364: the R2R branch (view + location embeddings through the panorama encoder)
    and the REVERIE/SOON branch (object tokens appended to every panorama, :322-349).  Shipped pre-train configs
    keep do_back_img off — the upstream BACL-img pretrain branch is broken, SURVEY §8a-Q viii."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.reverie = getattr(config, 'name', 'R2R') in ('REVERIE', 'SOON')
        if getattr(config, 'do_back_img', False):
            raise NotImplementedError('pretrain do_back_img is broken upstream (undefined do_back_img_after_linear)')
        H = config.hidden_size
        self.img_linear = Linear(config.image_feat_size, H)
        self.img_layer_norm = BertLayerNorm(H, eps=1e-12)
        self.loc_linear = Linear(config.angle_feat_size + 3, H)
        self.loc_layer_norm = BertLayerNorm(H, eps=1e-12)
        if not self.reverie:
            self.img_self_attn = BertAttention(config)          # created, never used (checkpoint compat)
            self.img_self_encoder = create_transformer_encoder(config, config.num_pano_layers, norm=True)
        if self.reverie:
            self.obj_name_linear = nn.Embedding(config.obj_name_vocab_size, H)
            self.obj_reverie_linear = Linear(config.obj_feat_size, H)
            self.obj_reverie_layer_norm = BertLayerNorm(H, eps=1e-12)
            self.nav_type_embedding = nn.Embedding(3, H)
            self.pano_encoder = create_transformer_encoder(config, config.num_pano_layers, norm=True)
        else:
            self.nav_type_embedding = nn.Embedding(2, H)    # unused on R2R
        if config.adaptive_pano_fusion:
            self.adaptive_pano_attn = Linear(H, 1)
        self.layer_norm = BertLayerNorm(H, eps=1e-12)       # unused on R2R
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, traj_view_img_fts, traj_loc_fts, traj_vp_view_lens, traj_nav_types=None, obj_fts=None, obj_lens=None,
                obj_names=None, cat_index=None, cat_inverse=None):
        """-> (tokens [N,W,H], fused [N,H] | None).  cat_index = graphmap.build_obj_concat_index(...) on device (cat_inverse: its
        graphmap.inverse_index, optional)."""
        dt = compute_dtype()
        x = self.img_layer_norm(self.img_linear(traj_view_img_fts.to(dt)))
        if not self.reverie:
            # dropout(img_LN(...) + loc_LN(...)): the sum and the dropout inside the second LayerNorm's launch
            x = self.loc_layer_norm(self.loc_linear(traj_loc_fts.to(dt)), post_add=x, p_out=_p(self.dropout))
            img_masks = gen_seq_masks(traj_vp_view_lens, traj_view_img_fts.shape[1])
            x = self.img_self_encoder(x, img_masks)
        if obj_fts is not None:
            o = self.obj_reverie_linear(obj_fts.to(dt))
            if self.config.use_obj_name:
                o = o + hipops.embedding(obj_names, self.obj_name_linear.weight, out_dtype=dt)
            o = self.obj_reverie_layer_norm(o)
            N, V, H = x.shape
            W = traj_nav_types.shape[1]
            src = torch.cat([x.reshape(N * V, H), o.reshape(-1, H)], 0)
            x = hipops.gather_segmean(src, cat_index[0], cat_index[1], None, N * W, cat_inverse).view(N, W, H)
            x = x + hipops.embedding(traj_nav_types, self.nav_type_embedding.weight, out_dtype=dt) \
                + self.loc_layer_norm(self.loc_linear(traj_loc_fts.to(dt)))
            x = self.layer_norm(x, p_out=_p(self.dropout))
            x = self.pano_encoder(x, gen_seq_masks(traj_vp_view_lens + obj_lens, W))
        fused = None
        if self.config.adaptive_pano_fusion:
            fused = hipops.pano_fusion(x, self.adaptive_pano_attn.weight, self.adaptive_pano_attn.bias)
        return x, fused


class LocalVPEncoder(nn.Module):
    # P/model/vilmodel_goat.py:366-410
    def __init__(self, config):
        super().__init__()
        self.vp_pos_embeddings = nn.Sequential(Linear(config.angle_feat_size * 2 + 6, config.hidden_size),
                                               BertLayerNorm(config.hidden_size, eps=1e-12))
        self.encoder = CrossmodalEncoder(config)
        if 'cfp' in config.pretrain_tasks:
            self.tim_self_encoder = BertAttention(config)

    def vp_input_embedding(self, pano_embeds, idx, vp_pos_fts, inverse=None):
        """pano_embeds [N,V,H]; idx = graphmap.build_vp_index(...) on device (inverse: its graphmap.inverse_index, optional)."""
        vidx, vstart, vp_lens, width = idx
        B = vp_pos_fts.shape[0]
        vp_img = hipops.gather_segmean(pano_embeds, vidx, vstart, None, B * width, inverse).view(B, width, -1)
        # (kept as a separate add: folding it into the LayerNorm launch (post_add) removes one bf16 rounding, which moved the most
        #  noise-sensitive gradient of the model — sap_fuse_linear, a difference of two softmax-weighted sums — past its calibrated
        #  bf16 bound in 2 of 16 parity cases; 3 us per step are not worth re-calibrating the bound)
        pos = self.vp_pos_embeddings[1](self.vp_pos_embeddings[0](vp_pos_fts[:, :width].to(vp_img.dtype)))
        return vp_img + pos, gen_seq_masks(vp_lens, width)


class GlobalMapEncoder(nn.Module):
    # P/model/vilmodel_goat.py:412-527
    def __init__(self, config):
        super().__init__()
        self.gmap_pos_embeddings = nn.Sequential(Linear(config.angle_feat_size + 3, config.hidden_size),
                                                 BertLayerNorm(config.hidden_size, eps=1e-12))
        self.gmap_step_embeddings = nn.Embedding(config.max_action_steps, config.hidden_size)
        self.encoder = CrossmodalEncoder(config)
        if 'cfp' in config.pretrain_tasks:
            self.tim_self_encoder = BertAttention(config)
        self.sprel_linear = Linear(1, 1) if config.graph_sprels else None

    def gmap_input_embedding(self, src_rows, idx, gmap_step_ids, gmap_pos_fts, gmap_lens, inverse=None):
        gidx, gstart, gscale = idx
        B, G = gmap_step_ids.shape
        img = hipops.gather_segmean(src_rows, gidx, gstart, gscale, B * G, inverse).view(B, G, -1)
        pos = self.gmap_pos_embeddings[1](self.gmap_pos_embeddings[0](gmap_pos_fts.to(img.dtype)))
        e = img + hipops.embedding(gmap_step_ids, self.gmap_step_embeddings.weight, out_dtype=img.dtype) + pos
        return e, gen_seq_masks(gmap_lens, G)

    def sprels(self, gmap_pair_dists):
        # Linear(1,1) on the distance matrix (:496-497); float32, gradient flows back through the attention bias
        w, b = self.sprel_linear.weight.view(()), self.sprel_linear.bias.view(())
        return gmap_pair_dists.float() * w + b


class GlocalTextPathCMT(GoatPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.embeddings = RobertaEmbeddings(config)
        self.lang_encoder = LanguageEncoderDo(config) if getattr(config, 'do_back_txt', False) else LanguageEncoder(config)
        self.img_embeddings = CausalImageEmbeddings(config)
        self.local_encoder = LocalVPEncoder(config)
        self.global_encoder = GlobalMapEncoder(config)
        self.init_weights()

    # -- shared stems -----------------------------------------------------------------------
    def _indices(self, batch):
        """Per-batch host index tensors, cached on the caller's batch dict."""
        cache = batch.get('_goat_cache')
        if cache is None:
            dev = batch['traj_view_img_fts'].device
            V = batch['traj_view_img_fts'].shape[1]
            G = batch['gmap_step_ids'].shape[1]
            fused = bool(self.config.adaptive_pano_fusion)
            lens_cpu = batch['traj_vp_view_lens'].cpu()
            cache = {}
            if batch.get('traj_obj_img_fts') is not None:
                # REVERIE/SOON: every panorama row becomes [views | objects], W slots wide (P/model/vilmodel_goat.py:331-341)
                obj_cpu = batch['traj_vp_obj_lens'].cpu()
                W = batch['traj_nav_types'].shape[1]
                ci = graphmap.build_obj_concat_index(lens_cpu, obj_cpu, V, batch['traj_obj_img_fts'].shape[1], W)
                cache['objcat'] = (ci[0].to(dev), ci[1].to(dev))
                n_rows = int(lens_cpu.shape[0])
                cache['objcat_inv'] = tuple(t.to(dev) for t in graphmap.inverse_index(
                    ci[0], ci[1], None, n_rows * V + n_rows * batch['traj_obj_img_fts'].shape[1]) if t is not None)
                cache['view_lens_cpu'], cache['obj_lens_cpu'] = lens_cpu, obj_cpu
                lens_cpu, V = lens_cpu + obj_cpu, W
            g = graphmap.build_gmap_index(batch['traj_step_lens'], lens_cpu, batch['traj_vpids'],
                                          batch['traj_cand_vpids'], batch['gmap_vpids'], G, V, fused)
            v = graphmap.build_vp_index(batch['traj_step_lens'], lens_cpu, V)
            cache['gmap'] = tuple(t.to(dev) for t in g)
            cache['vp'] = (v[0].to(dev), v[1].to(dev), v[2].to(dev), v[3])
            # inverse indices: the backward passes of the two gathers are gathers over the output gradients (no atomics / fills)
            n_rows = int(lens_cpu.shape[0])
            cache['gmap_inv'] = tuple(t.to(dev) for t in graphmap.inverse_index(g[0], g[1], g[2], n_rows * V + (n_rows if fused else 0)))
            cache['vp_inv'] = tuple(t.to(dev) for t in graphmap.inverse_index(v[0], v[1], None, n_rows * V) if t is not None)
            batch['_goat_cache'] = cache
        return cache

    def _text(self, batch):
        txt_masks = gen_seq_masks(batch['txt_lens'], batch['txt_ids'].shape[1])
        txt_kmask = neg_mask(txt_masks)
        e = self.embeddings(batch['txt_ids'])
        if getattr(self.config, 'do_back_txt', False):
            # the embeddings pass the dictionaries through as float32 (P/model/Bert_backbone.py:117-120)
            txt = self.lang_encoder(e, txt_kmask, batch['instr_z_direction_features'], batch['instr_z_direction_pzs'],
                                    batch['instr_z_landmark_features'], batch['instr_z_landmark_pzs'])
        else:
            txt = self.lang_encoder(e, txt_kmask)
        return txt, txt_kmask

    def _pano(self, batch):
        cache = self._indices(batch)
        x, fused = self.img_embeddings(batch['traj_view_img_fts'], batch['traj_loc_fts'], batch['traj_vp_view_lens'],
                                       batch.get('traj_nav_types'), batch.get('traj_obj_img_fts'),
                                       batch.get('traj_vp_obj_lens'), batch.get('traj_reverie_obj_names'),
                                       cache.get('objcat'), cache.get('objcat_inv'))
        N, V, H = x.shape
        rows = x.view(N * V, H)
        src = torch.cat([rows, fused], 0) if fused is not None else rows
        return x, src

    def _gmap_in(self, batch, src, cache):
        return self.global_encoder.gmap_input_embedding(src, cache['gmap'], batch['gmap_step_ids'],
                                                        batch['gmap_pos_fts'], batch['gmap_lens'], cache.get('gmap_inv'))

    def _vp_in(self, batch, x, cache):
        return self.local_encoder.vp_input_embedding(x, cache['vp'], batch['vp_pos_fts'], cache.get('vp_inv'))

    # -- the three reference entry points ---------------------------------------------------------
    def forward(self, batch, return_gmap_embeds=True):
        """GlocalTextPathCMT.forward (P/model/vilmodel_goat.py:546-594) -> (gmap_embeds, vp_embeds, txt_embeds).
        The panorama stem runs as a parallel branch of the text encoder, the global-map encoder as a parallel branch of
        the local one (hipops.Branch): independent sub-graphs whose kernels are too small to fill the chip alone."""
        cache = self._indices(batch)
        with hipops.Branch('pano') as br:
            x, src = self._pano(batch)
        txt, txt_kmask = self._text(batch)
        br.join(x, src)
        gmap = None
        kv_g = kv_v = None
        if return_gmap_embeds and hipops.LINEAR_BANK:
            # the text states are the attended sequence of all six cross-attention layers (three global, three local): their key | value
            # projections are one GEMM in front of the fork (layers.project_kv_bank); one handle for that bank, one for the caller
            t_kv, t_out = hipops.fanout(txt, 2)
            kv_g, kv_v = project_kv_bank(t_kv, [self.global_encoder.encoder, self.local_encoder.encoder])
            t_g = t_v = t_kv
        else:
            t_g, t_v, t_out = hipops.fanout(txt, 3)           # the text states feed both cross-modal encoders (and the caller)
        if return_gmap_embeds:
            with hipops.Branch('global') as bg:
                g, gm = self._gmap_in(batch, src, cache)
                bias = self.global_encoder.sprels(batch['gmap_pair_dists']) if self.global_encoder.sprel_linear is not None else None
                gmap = self.global_encoder.encoder(g, neg_mask(gm), t_g, txt_kmask, bias, kv_cache=kv_g)
        v, vm = self._vp_in(batch, x, cache)
        vp = self.local_encoder.encoder(v, neg_mask(vm), t_v, txt_kmask, kv_cache=kv_v)
        if return_gmap_embeds:
            bg.join(gmap)
        return gmap, vp, t_out

    def forward_mlm(self, batch):
        # P/model/vilmodel_goat.py:597-648: text queries attend to map / local tokens, outputs summed
        cache = self._indices(batch)
        with hipops.Branch('pano') as br:
            x, src = self._pano(batch)
        txt, txt_kmask = self._text(batch)
        br.join(x, src)
        t = hipops.fanout(txt, 4)                         # text queries of both encoders, each read twice by its first layer (a _pair)
        with hipops.Branch('global') as bg:
            g, gm = self._gmap_in(batch, src, cache)
            gt = self.global_encoder.encoder((t[0], t[1]), txt_kmask, g, neg_mask(gm))
        v, vm = self._vp_in(batch, x, cache)
        vt = self.local_encoder.encoder((t[2], t[3]), txt_kmask, v, neg_mask(vm))
        bg.join(gt)
        return gt + vt

    def forward_cfp(self, batch):
        # P/model/vilmodel_goat.py:650-696: single self-attention blocks, no cross-modal encoders
        cache = self._indices(batch)
        with hipops.Branch('pano') as br:
            x, src = self._pano(batch)
        txt, _ = self._text(batch)
        br.join(x, src)
        with hipops.Branch('global', 'cfp_enc') as bg:
            g, gm = self._gmap_in(batch, src, cache)
            gmap = self.global_encoder.tim_self_encoder(g, neg_mask(gm))
        v, vm = self._vp_in(batch, x, cache)
        vp = self.local_encoder.tim_self_encoder(v, neg_mask(vm))
        bg.join(gmap)
        return gmap, vp, txt


def attn_pool(x, w, smask=None):
    """tanh-attention pooling over ALL slots, no padding mask (P/model/pretrain_goat.py:502-515); float32 [B,H].
    smask: see hipops.attn_pool (bucket padding only)."""
    return hipops.attn_pool(x, w, smask)


def cfp_losses(gmap_o, vp_o, fused_o, txt_o, temperature, gather=None):
    """3 x symmetric InfoNCE (P/model/pretrain_goat.py:519-534).  `gather` (dp.CfpGather) extends the
    negatives across data-parallel ranks; with world_size 1 / None it is exactly the reference."""
    B = gmap_o.shape[0]
    if gather is not None:
        gmap_a, vp_a, fused_a, txt_a, off = gather(gmap_o, vp_o, fused_o, txt_o)
    else:
        gmap_a, vp_a, fused_a, txt_a, off = gmap_o, vp_o, fused_o, txt_o, 0
    if gmap_o.is_cuda:      # one forward + one backward HIP launch instead of ~70 ATen kernels (18 mm, 19 div, 6 log_softmax ...)
        return hipops.infonce(gmap_o, vp_o, fused_o, txt_o, gmap_a, vp_a, fused_a, txt_a, off, float(temperature))
    # (CPU tensors: only the data-parallel engine's gloo tests come here)
    target = torch.arange(B, device=gmap_o.device) + off

    def sym(x_loc, x_all):
        row = F.cross_entropy((x_loc @ txt_a.T) / temperature, target, reduction='none')       # image -> all texts
        col = F.cross_entropy((txt_o @ x_all.T) / temperature, target, reduction='none')        # text -> all images
        return (row + col) / 2.0
    return sym(gmap_o, gmap_a) + sym(vp_o, vp_a) + sym(fused_o, fused_a)


class GlocalTextPathCMTPreTraining(GoatPreTrainedModel):
    def __init__(self, config):
        super().__init__(config)
        self.bert = GlocalTextPathCMT(config)
        tasks = config.pretrain_tasks
        if 'mlm' in tasks:
            self.mlm_head = BertOnlyMLMHead(config)
        if 'mrc' in tasks:
            self.image_classifier = RegionClassification(config.hidden_size, config.image_prob_size)
            if config.obj_prob_size > 0 and config.obj_prob_size != config.image_prob_size:
                self.obj_classifier = RegionClassification(config.hidden_size, config.obj_prob_size)
            else:
                self.obj_classifier = None
        if 'sap' in tasks:
            self.global_sap_head = ClsPrediction(config.hidden_size)
            self.local_sap_head = ClsPrediction(config.hidden_size)
            self.sap_fuse_linear = ClsPrediction(config.hidden_size, input_size=config.hidden_size * 2) \
                if config.glocal_fuse else None
        if 'og' in tasks:
            self.og_head = ClsPrediction(config.hidden_size)
        if 'cfp' in tasks:
            self.tim_txt_head = BertPredictionHeadTransform(config)
            self.tim_global_head = BertPredictionHeadTransform(config)
            self.tim_local_head = BertPredictionHeadTransform(config)
            self.tim_fused_head = BertPredictionHeadTransform(config)      # created, never used (checkpoint compat)
            for n in ('tim_txt_attn', 'tim_global_attn', 'tim_local_attn', 'tim_fused_attn'):
                prm = nn.Parameter(torch.empty(config.hidden_size, 1))
                nn.init.uniform_(prm, -0.1, 0.1)
                setattr(self, n, prm)
            self.temperature = config.cfp_temperature
        self.cfp_gather = None   # set by dp.GoatDataParallel to share CFP negatives across ranks
        self.init_weights()
        self.tie_weights()

    def tie_weights(self):
        if 'mlm' in self.config.pretrain_tasks:
            self._tie_or_clone_weights(self.mlm_head.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, batch, task, compute_loss=True):
        cache_host = batch  # index cache is stored on the caller's dict
        if not isinstance(batch, defaultdict):
            wrapped = defaultdict(lambda: None, batch)
            wrapped['_goat_cache'] = batch.get('_goat_cache')
            batch = wrapped
        try:
            if task.startswith('mlm'):
                return self.forward_mlm(batch, compute_loss)
            elif task.startswith('sap'):
                return self.forward_sap(batch, compute_loss)
            elif task.startswith('cfp'):
                return self.forward_cfp(batch, compute_loss)
            elif task.startswith('mrc'):
                return self.forward_mrc(batch, compute_loss)
            elif task.startswith('og'):
                return self.forward_og(batch, compute_loss)
            elif task.startswith('valid_sap_og'):
                # upstream passes more arguments than forward_sap_og accepts (SURVEY §8a-Q ix): dead in the reference
                raise NotImplementedError('valid_sap_og is dead code in the reference (signature mismatch at pretrain_goat.py:155-169)')
            else:
                raise ValueError('invalid task')
        finally:
            if batch is not cache_host and batch.get('_goat_cache') is not None:
                try:
                    cache_host['_goat_cache'] = batch['_goat_cache']
                except TypeError:
                    pass

    # -- MLM ---------------------------------------------------------------------------------------
    def forward_mlm(self, batch, compute_loss):
        txt = self.bert.forward_mlm(batch)
        labels = batch['txt_labels']
        cache = batch['_goat_cache']
        if 'mlm_idx' not in cache:
            cache['mlm_idx'] = (labels.reshape(-1) != -1).nonzero().squeeze(1)
            cache['mlm_tgt'] = labels.reshape(-1)[cache['mlm_idx']]
        if cache['mlm_idx'].numel() == 0:            # no masked token in the batch: empty loss / score tensor, as the reference
            empty = txt.reshape(-1, txt.shape[-1])[:0].float()
            return empty.sum(1) if compute_loss else empty.new_zeros((0, self.config.vocab_size))
        masked = txt.reshape(-1, txt.shape[-1]).index_select(0, cache['mlm_idx'])
        if compute_loss:
            loss = self.mlm_head.predictions.loss(masked, cache['mlm_tgt'])
            if cache.get('mlm_scale') is not None:
                # shape-bucketed static batch (train_step.StaticBatch): the selection is padded to the bucket's capacity with ignored
                # rows (loss 0); the vector is rescaled by capacity / real rows (a device scalar) so that its MEAN — what the trainer
                # takes, P/train_r2r_goat.py:333 — is the mean over the real masked tokens
                loss = loss * cache['mlm_scale']
            return loss
        return self.mlm_head(masked)            # float32 logits

    # -- SAP ---------------------------------------------------------------------------------------
    def _fuse_weights(self, gmap_embeds, vp_embeds):
        if self.sap_fuse_linear is None:
            return 0.5
        return torch.sigmoid(self.sap_fuse_linear(torch.cat([gmap_embeds[:, 0], vp_embeds[:, 0]], 1)).float())

    def forward_sap(self, batch, compute_loss):
        gmap, vp, _ = self.bert(batch)
        cache = batch['_goat_cache']
        B, G = gmap.shape[:2]
        W = vp.shape[1]
        if 'sap' not in cache:
            step_lens = batch['traj_step_lens']
            last = torch.as_tensor(step_lens).cumsum(0) - 1
            nav = batch['traj_nav_types'][last.to(batch['traj_nav_types'].device)] != 1     # [B, V]
            nav = torch.cat([torch.zeros(B, 1, dtype=torch.bool, device=nav.device), nav], 1)[:, :W]
            M = graphmap.build_sap_fusion(batch['traj_cand_vpids'], batch['gmap_vpids'], batch['gmap_visited_masks'], G, W)
            cache['sap'] = (nav, M.to(gmap.device))
        nav, M = cache['sap']
        # scores of the two heads -> masked global / local / fused logits (and the three cross-entropies) in ONE launch per direction
        # (hipops.sap_fuse: the reference's * fw, masked_fill x4, bmm, log_softmax x3 ... are ~25 launches on [48, 22..37] tensors)
        (gmap, gmap0), (vp, vp0) = hipops.fanout(gmap, 2), hipops.fanout(vp, 2)      # head + [CLS] row of the fusion weight
        fwl = None if self.sap_fuse_linear is None else self.sap_fuse_linear(torch.cat([gmap0[:, 0], vp0[:, 0]], 1))
        ga, la = batch['global_act_labels'], batch['local_act_labels']
        gl, ll, fused, loss = hipops.sap_fuse(self.global_sap_head(gmap).squeeze(2), self.local_sap_head(vp).squeeze(2), fwl,
                                              gvis=batch['gmap_visited_masks'], glens=batch['gmap_lens'], lmask=nav, M=M,
                                              labels=(ga, la) if compute_loss else None)
        if compute_loss:
            return loss
        return gl, ll, fused, ga, la

    # -- OG / MRC (local stream only) ---------------------------------------------------------------
    def _last_lens(self, batch):
        """(view_len, obj_len) of each sample's LAST panorama, as python lists (host index bookkeeping)."""
        cache = batch['_goat_cache']
        if 'last_lens' not in cache:
            last = torch.as_tensor(batch['traj_step_lens']).cumsum(0) - 1
            vl = batch['traj_vp_view_lens'].cpu()[last].tolist()
            ol = batch['traj_vp_obj_lens'].cpu()[last].tolist() if batch['traj_vp_obj_lens'] is not None else [0] * len(vl)
            cache['last_lens'] = (vl, ol)
        return cache['last_lens']

    def forward_og(self, batch, compute_loss):
        """P/model/pretrain_goat.py:356-391: ClsPrediction on the object tokens of the local stream, -inf outside.
        The head is per token, so it runs on all local tokens and the object logits are gathered from the result
        (same values as pad_tensors_wgrad of the object slices followed by the head)."""
        _, vp, _ = self.bert(batch, return_gmap_embeds=False)
        cache = batch['_goat_cache']
        if 'og_idx' not in cache:
            vl, ol = self._last_lens(batch)
            O = max(1, max(ol))
            idx = torch.zeros(len(vl), O, dtype=torch.int64)
            msk = torch.zeros(len(vl), O, dtype=torch.bool)
            for b, (v, o) in enumerate(zip(vl, ol)):
                idx[b, :o] = torch.arange(1 + v, 1 + v + o)
                msk[b, :o] = True
            cache['og_idx'] = (idx.to(vp.device), msk.to(vp.device))
        idx, msk = cache['og_idx']
        # (torch.where, not masked_fill: the latter clones first, and a clone is a memcpy NODE of the captured step graph — §4c of DESIGN.md)
        logits = torch.where(msk, self.og_head(vp).squeeze(2).float().gather(1, idx), -float('inf'))
        if compute_loss:
            return F.cross_entropy(logits, batch['obj_labels'], reduction='none')
        return logits

    def _mrc_rows(self, batch, which, width):
        """flat row indices (into vp.view(B*W1, H)) of the masked view / object tokens, in the reference's order."""
        cache = batch['_goat_cache']
        key = 'mrc_' + which
        if key not in cache:
            vl, ol = self._last_lens(batch)
            mask = batch['vp_view_mrc_masks' if which == 'view' else 'vp_obj_mrc_masks'].cpu()
            rows = []
            for b in range(mask.shape[0]):
                n = vl[b] if which == 'view' else ol[b]
                off = 1 if which == 'view' else 1 + vl[b]
                for j in range(mask.shape[1]):
                    if mask[b, j]:
                        if j >= n:
                            raise ValueError('MRC mask selects a padded %s slot (sample %d, slot %d)' % (which, b, j))
                        rows.append(b * width + off + j)
            dev = batch['traj_view_img_fts'].device
            # (the soft-label rows as flat indices too: boolean-mask indexing would synchronise, and could not be captured)
            cache[key] = (torch.tensor(rows, dtype=torch.int64, device=dev), mask.reshape(-1).nonzero().squeeze(1).to(dev))
        return cache[key]

    def forward_mrc(self, batch, compute_loss):
        """P/model/pretrain_goat.py:226-284: region classification (KL to soft labels) on the masked view tokens —
        and object tokens, REVERIE — of the local stream; only the masked rows go through the classifier."""
        _, vp, _ = self.bert(batch, return_gmap_embeds=False)
        B, W1, H = vp.shape
        flat = vp.reshape(B * W1, H)
        rows, vsel = self._mrc_rows(batch, 'view', W1)
        v_pred = self.image_classifier(flat.index_select(0, rows)).float()
        probs = batch['vp_view_probs']
        v_tgt = probs.reshape(-1, probs.shape[-1]).index_select(0, vsel)
        o_pred = o_tgt = None
        if batch['traj_obj_img_fts'] is not None:
            orow, osel = self._mrc_rows(batch, 'obj', W1)
            head = self.obj_classifier if self.obj_classifier is not None else self.image_classifier
            o_pred = head(flat.index_select(0, orow)).float()
            probs = batch['vp_obj_probs']
            o_tgt = probs.reshape(-1, probs.shape[-1]).index_select(0, osel)
        if not compute_loss:
            return v_pred, v_tgt, o_pred, o_tgt
        cache = batch['_goat_cache']
        loss = F.kl_div(F.log_softmax(v_pred, dim=-1), v_tgt.float(), reduction='none').sum(dim=1)
        wv, wo = cache.get('mrc_view_w'), cache.get('mrc_obj_w')
        if wv is not None:           # shape-bucketed static batch: selection padded to a capacity, padding rows weigh 0
            loss = loss * wv
        if o_pred is not None:
            lo = F.kl_div(F.log_softmax(o_pred, dim=-1), o_tgt.float(), reduction='none').sum(dim=1)
            loss = torch.cat([loss, lo * wo if wo is not None else lo], 0)
        if wv is not None:           # ... and the vector is rescaled so that its mean is the mean over the real rows
            n_real = wv.sum() + (wo.sum() if (o_pred is not None and wo is not None) else 0.0)
            loss = loss * (loss.shape[0] / n_real)
        return loss

    # -- CFP ---------------------------------------------------------------------------------------
    def forward_cfp(self, batch, compute_loss):
        gmap, vp, txt = self.bert.forward_cfp(batch)
        cache = batch.get('_goat_cache') or {}
        with hipops.Branch('global', 'cfp_heads') as bg:          # the three heads are independent: map / text on side streams
            if batch['extra_heads']:
                gmap = self.tim_global_head(gmap)
            go = attn_pool(gmap, self.tim_global_attn, cache.get('cfp_gmap_mask'))
        with hipops.Branch('pano', 'cfp_heads') as bt:
            if batch['extra_heads']:
                txt = self.tim_txt_head(txt)
            to = attn_pool(txt, self.tim_txt_attn, cache.get('cfp_txt_mask'))
        if batch['extra_heads']:
            vp = self.tim_local_head(vp)
        vo = attn_pool(vp, self.tim_local_attn, cache.get('cfp_vp_mask'))
        bg.join(gmap, go)
        bt.join(txt, to)
        if self.sap_fuse_linear is not None and go.is_cuda and hipops.FANOUT:
            # fusion logit -> sigmoid -> fused vector in one launch per direction (hipops.cfp_mix); on a single rank the three InfoNCE losses
            # behind it join the same autograd node (hipops.cfp_tail: no engine adds for the gradients of the pooled vectors)
            fwl = self.sap_fuse_linear(torch.cat([gmap[:, 0], vp[:, 0]], 1))
            single = self.cfp_gather is None or not (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                     and torch.distributed.get_world_size() > 1)
            if compute_loss and single:
                return hipops.cfp_tail(go, vo, fwl, to, self.temperature)
            fo = hipops.cfp_mix(go, vo, fwl)
        else:
            fw = self._fuse_weights(gmap, vp)
            fo = go * fw + vo * (1 - fw)
        if compute_loss:
            return cfp_losses(go, vo, fo, to, self.temperature, self.cfp_gather)
        return go, vo, fo, to
