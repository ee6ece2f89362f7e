"""CPU oracle: a plain-PyTorch fp32 restatement of GOAT's pre-training forward path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product package
(`vln-goat_amd/`) never imports it and fails loudly when its HIP library is missing.

It is a functional restatement (state-dict in, tensors out; autograd supplies the gradients) of the
reference algorithm, each function citing the reference file:line it follows
(P/ = /root/reference/pretrain_src/).  Parity pinning: `tests/golden/*.npz` hold outputs of the
*imported reference itself* (generated in the build container by tests/golden/make_golden_pretrain.py);
`tests/test_oracle_golden.py` checks this oracle against them to <=1e-5.

Numerics: fp32, bidirectional attention, additive -10000 masks in BERT blocks, -inf key-padding in
the panorama encoder, erf-GELU, LayerNorm eps 1e-12 except pano inner norms (1e-5).
"""
import math
from collections import defaultdict

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- small helpers
def _lin(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


def gelu(x):
    # P/model/Bert_backbone.py:41-47
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gen_seq_masks(seq_lens, max_len=None):
    # P/model/ops.py:36-44
    if max_len is None:
        max_len = int(max(seq_lens))
    return torch.arange(max_len, device=seq_lens.device).unsqueeze(0) < seq_lens.unsqueeze(1)


def extend_neg_masks(masks):
    # P/model/ops.py:25-34
    return (1.0 - masks.unsqueeze(1).unsqueeze(2).float()) * -10000.0


def pad_tensors_wgrad(tensors):
    # P/model/ops.py:46-68
    max_len = max(t.size(0) for t in tensors)
    out = []
    for t in tensors:
        if t.size(0) < max_len:
            t = torch.cat([t, t.new_zeros((max_len - t.size(0),) + tuple(t.shape[1:]))], 0)
        out.append(t)
    return torch.stack(out, 0)


class Ctx:
    """cfg + dropout switch (p=0 reproduces eval-mode; training=True draws torch RNG masks)."""

    def __init__(self, cfg, training=False):
        self.cfg = cfg
        self.training = training

    def drop(self, x, p):
        return F.dropout(x, p, self.training) if (self.training and p > 0) else x


# ----------------------------------------------------------------------------- BERT blocks
def self_attention(ctx, sd, pre, hidden, attn_mask, enc_hidden=None, enc_mask=None):
    """BertSelfAttention / RobertaSelfAttention.forward (P/model/Bert_backbone.py:199-296, 419-512).
    For cross-attention the query-side `attn_mask` is ignored (:221-224)."""
    cfg = ctx.cfg
    nh = cfg.num_attention_heads
    q = _lin(sd, pre + '.query', hidden)
    kv_src = hidden if enc_hidden is None else enc_hidden
    mask = attn_mask if enc_hidden is None else enc_mask
    k = _lin(sd, pre + '.key', kv_src)
    v = _lin(sd, pre + '.value', kv_src)

    def split(x):
        return x.view(x.shape[0], x.shape[1], nh, -1).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if mask is not None:
        scores = scores + mask
    probs = ctx.drop(torch.softmax(scores, dim=-1), cfg.attention_probs_dropout_prob)
    ctxl = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
    return ctxl.view(ctxl.shape[0], ctxl.shape[1], -1)


def self_output(ctx, sd, pre, hidden, input_tensor):
    # BertSelfOutput/BertOutput (P/model/Bert_backbone.py:299-310, 359-370)
    h = ctx.drop(_lin(sd, pre + '.dense', hidden), ctx.cfg.hidden_dropout_prob)
    return _ln(sd, pre + '.LayerNorm', h + input_tensor, ctx.cfg.layer_norm_eps)


def bert_attention(ctx, sd, pre, hidden, attn_mask, enc_hidden=None, enc_mask=None):
    # BertAttention / RobertaAttention (P/model/Bert_backbone.py:313-342, 515-543)
    so = self_attention(ctx, sd, pre + '.self', hidden, attn_mask, enc_hidden, enc_mask)
    return self_output(ctx, sd, pre + '.output', so, hidden)


def ffn(ctx, sd, pre_inter, pre_out, x):
    # BertIntermediate + BertOutput (P/model/Bert_backbone.py:345-370)
    inter = gelu(_lin(sd, pre_inter + '.dense', x))
    return self_output(ctx, sd, pre_out, inter, x)


def roberta_layer(ctx, sd, pre, hidden, ext_mask):
    # RobertaLayer.forward (P/model/Bert_backbone.py:595-659)
    a = bert_attention(ctx, sd, pre + '.attention', hidden, ext_mask)
    return ffn(ctx, sd, pre + '.intermediate', pre + '.output', a)


def cross_layer(ctx, sd, pre, hidden, enc_hidden, attn_mask, enc_mask, graph_sprels=None):
    # BertCrossLayer.forward (P/model/Bert_backbone.py:678-727)
    if graph_sprels is not None:
        attn_mask = attn_mask + graph_sprels
    a = bert_attention(ctx, sd, pre + '.attention', hidden, attn_mask)
    a = bert_attention(ctx, sd, pre + '.crossattention', a, attn_mask, enc_hidden, enc_mask)
    return ffn(ctx, sd, pre + '.intermediate', pre + '.output', a)


def crossmodal_encoder(ctx, sd, pre, q, q_masks, kv, kv_masks, graph_sprels=None):
    # CrossmodalEncoder.forward (P/model/Bert_backbone.py:765-781)
    if q_masks.dim() != 4:
        q_masks = extend_neg_masks(q_masks)
    if kv_masks.dim() != 4:
        kv_masks = extend_neg_masks(kv_masks)
    for i in range(ctx.cfg.num_top_layer):
        q = cross_layer(ctx, sd, f'{pre}.crossattention.{i}', q, kv, q_masks, kv_masks, graph_sprels)
    return q


def embeddings(ctx, sd, pre, txt_ids):
    # RobertaEmbeddings.forward (P/model/Bert_backbone.py:85-121): position ids = arange(L)
    L = txt_ids.shape[1]
    e = sd[pre + '.word_embeddings.weight'][txt_ids] \
        + sd[pre + '.token_type_embeddings.weight'][torch.zeros_like(txt_ids)] \
        + sd[pre + '.position_embeddings.weight'][torch.arange(L, device=txt_ids.device)].unsqueeze(0)
    e = _ln(sd, pre + '.LayerNorm', e, ctx.cfg.layer_norm_eps)
    return ctx.drop(e, ctx.cfg.hidden_dropout_prob)


def lang_encoder(ctx, sd, pre, txt_embeds, txt_masks):
    # LanguageEncoder.forward (P/model/vilmodel_goat.py:37-44)
    ext = extend_neg_masks(txt_masks)
    for i in range(ctx.cfg.num_l_layers):
        txt_embeds = roberta_layer(ctx, sd, f'{pre}.layer.{i}', txt_embeds, ext)
    return txt_embeds


def lang_encoder_do(ctx, sd, pre, txt_embeds, txt_masks, batch):
    """LanguageEncoderDo.forward — BACL-txt in pre-training (P/model/vilmodel_goat.py:104-159); no front-door
    dictionary is passed on this path (:549)."""
    cfg = ctx.cfg
    ext = extend_neg_masks(txt_masks)
    for i in range(cfg.num_l_layers):
        txt_embeds = roberta_layer(ctx, sd, f'{pre}.layer.{i}', txt_embeds, ext)
    zd = batch['instr_z_direction_features']
    if zd is None:
        return txt_embeds
    zd = zd.to(torch.float32)                               # RobertaEmbeddings passes them through (Bert_backbone.py:117-120)
    zl = batch['instr_z_landmark_features'].to(torch.float32)
    eps = cfg.layer_norm_eps
    if cfg.do_back_txt_type == 'type_1':
        if cfg.z_cross_attn:
            zd = bert_attention(ctx, sd, pre + '.z_direc_cross_attn', zd, None, txt_embeds, ext)
            zl = bert_attention(ctx, sd, pre + '.z_landm_cross_attn', zl, None, txt_embeds, ext)
        sum_d = torch.sum(zd * batch['instr_z_direction_pzs'].to(torch.float32), 1).unsqueeze(1)
        sum_l = torch.sum(zl * batch['instr_z_landmark_pzs'].to(torch.float32), 1).unsqueeze(1)
        txt_embeds = _lin(sd, pre + '.z_txt_linear', txt_embeds) + _lin(sd, pre + '.z_direct_linear', sum_d) \
            + _lin(sd, pre + '.z_landm_linear', sum_l)
        return _ln(sd, pre + '.z_concat_layernorm', txt_embeds, eps)
    zd = bert_attention(ctx, sd, pre + '.z_direc_cross_attn', txt_embeds, None, zd, None)
    zd = _ln(sd, pre + '.z_direct_ln', _lin(sd, pre + '.z_direct_linear', zd), eps)
    zl = bert_attention(ctx, sd, pre + '.z_landm_cross_attn', txt_embeds, None, zl, None)
    zl = _ln(sd, pre + '.z_landm_ln', _lin(sd, pre + '.z_landm_linear', zl), eps)
    if cfg.do_add_method == 'door':
        aug = zd + zl
        w = torch.sigmoid(_lin(sd, pre + '.instr_aug_linear', aug) + _lin(sd, pre + '.instr_ori_linear', txt_embeds))
        txt_embeds = w * aug + (1 - w) * txt_embeds
    elif cfg.do_add_method == 'add':
        txt_embeds = txt_embeds + zd + zl
    elif cfg.do_add_method == 'concat':
        txt_embeds = _lin(sd, pre + '.concat_linear', torch.cat((txt_embeds, zd, zl), -1))
    return _ln(sd, pre + '.z_concat_layernorm', txt_embeds, eps)


# ----------------------------------------------------------------------------- panorama encoder
def pano_layer(ctx, sd, pre, src, key_pad):
    """TransformerEncoderLayer.forward_pre (P/model/transformer.py:170-182) with
    nn.MultiheadAttention packed in_proj; `src` is batch-first [N,V,H] here (the reference
    transposes to seq-first, mathematically identical).  key_pad: bool [N,V], True = padded."""
    cfg = ctx.cfg
    nh = cfg.num_attention_heads
    p = cfg.hidden_dropout_prob
    x2 = _ln(sd, pre + '.norm1', src, 1e-5)
    qkv = F.linear(x2, sd[pre + '.self_attn.in_proj_weight'], sd[pre + '.self_attn.in_proj_bias'])
    q, k, v = qkv.chunk(3, dim=-1)

    def split(x):
        return x.view(x.shape[0], x.shape[1], nh, -1).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    scores = scores.masked_fill(key_pad[:, None, None, :], float('-inf'))
    probs = ctx.drop(torch.softmax(scores, dim=-1), p)
    a = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous().view(src.shape)
    a = _lin(sd, pre + '.self_attn.out_proj', a)
    src = src + ctx.drop(a, p)
    x2 = _ln(sd, pre + '.norm2', src, 1e-5)
    x2 = _lin(sd, pre + '.linear2', ctx.drop(F.gelu(_lin(sd, pre + '.linear1', x2)), p))
    return src + ctx.drop(x2, p)


def pano_encoder(ctx, sd, pre, x, key_pad):
    # TransformerEncoder.forward (P/model/transformer.py:71-89), final norm eps 1e-12 (P/model/ops.py:19-22)
    for i in range(ctx.cfg.num_pano_layers):
        x = pano_layer(ctx, sd, f'{pre}.layers.{i}', x, key_pad)
    return _ln(sd, pre + '.norm', x, 1e-12)


def image_embeddings(ctx, sd, pre, batch):
    """CausalImageEmbeddings.forward, no BACL (P/model/vilmodel_goat.py:289-364): R2R branch, or the REVERIE/SOON
    branch that appends object tokens to every panorama (:322-349)."""
    reverie = getattr(ctx.cfg, 'name', 'R2R') in ('REVERIE', 'SOON')
    x = _ln(sd, pre + '.img_layer_norm', _lin(sd, pre + '.img_linear', batch['traj_view_img_fts']), 1e-12)
    lens = batch['traj_vp_view_lens']
    if not reverie:
        x = x + _ln(sd, pre + '.loc_layer_norm', _lin(sd, pre + '.loc_linear', batch['traj_loc_fts']), 1e-12)
        img_masks = gen_seq_masks(lens)
        x = ctx.drop(x, ctx.cfg.hidden_dropout_prob)
        x = pano_encoder(ctx, sd, pre + '.img_self_encoder', x, img_masks.logical_not())
    if batch['traj_obj_img_fts'] is not None:
        # positional call at :567-572 binds traj_obj_img_fts / traj_vp_obj_lens to the reverie_obj_* arguments
        o = _lin(sd, pre + '.obj_reverie_linear', batch['traj_obj_img_fts'])
        if ctx.cfg.use_obj_name:
            o = o + sd[pre + '.obj_name_linear.weight'][batch['traj_reverie_obj_names']]
        o = _ln(sd, pre + '.obj_reverie_layer_norm', o, 1e-12)
        obj_lens = batch['traj_vp_obj_lens']
        rows = []
        for xv, xo, vl, ol in zip(x, o, lens, obj_lens):
            rows.append(torch.cat([xv[:vl], xo[:ol]], 0) if ol > 0 else xv[:vl])
        x = pad_tensors_wgrad(rows)
        lens = lens + obj_lens
        x = x + sd[pre + '.nav_type_embedding.weight'][batch['traj_nav_types']] \
            + _ln(sd, pre + '.loc_layer_norm', _lin(sd, pre + '.loc_linear', batch['traj_loc_fts']), 1e-12)
        x = _ln(sd, pre + '.layer_norm', x, 1e-12)
        x = ctx.drop(x, ctx.cfg.hidden_dropout_prob)
        x = pano_encoder(ctx, sd, pre + '.pano_encoder', x, gen_seq_masks(lens).logical_not())
    step_lens = batch['traj_step_lens']
    split_embeds = torch.split(x, step_lens, 0)
    split_lens = torch.split(lens, step_lens, 0)
    fused = None
    if ctx.cfg.adaptive_pano_fusion:
        w = torch.softmax(torch.tanh(_lin(sd, pre + '.adaptive_pano_attn', x)), dim=1)  # all slots, no mask
        fused = torch.split(torch.sum(x * w, dim=1), step_lens, 0)
    return split_embeds, split_lens, fused


# ----------------------------------------------------------------------------- graph-map / vp inputs
def aggregate_gmap_features(split_embeds, split_lens, traj_vpids, traj_cand_vpids, gmap_vpids, split_fused):
    # GlobalMapEncoder._aggregate_gmap_features (P/model/vilmodel_goat.py:430-468)
    out = []
    for i in range(len(split_embeds)):
        visited, unvisited = {}, {}
        vp_masks = gen_seq_masks(split_lens[i])
        max_vp_len = int(max(split_lens[i]))
        emb = split_embeds[i][:, :max_vp_len] * vp_masks.unsqueeze(2)
        for t in range(len(split_embeds[i])):
            if split_fused is not None:
                visited[traj_vpids[i][t]] = split_fused[i][t]
            else:
                visited[traj_vpids[i][t]] = torch.sum(emb[t], 0) / split_lens[i][t]
            for j, vp in enumerate(traj_cand_vpids[i][t]):
                if vp not in visited:
                    unvisited.setdefault(vp, []).append(emb[t][j])
        fts = []
        for vp in gmap_vpids[i][1:]:
            if vp in visited:
                fts.append(visited[vp])
            else:
                fts.append(torch.mean(torch.stack(unvisited[vp], 0), 0))
        out.append(torch.stack(fts, 0))
    out = pad_tensors_wgrad(out)
    return torch.cat([out.new_zeros(out.shape[0], 1, out.shape[2]), out], 1)


def gmap_input_embedding(ctx, sd, pre, batch, split_embeds, split_lens, split_fused):
    # GlobalMapEncoder.gmap_input_embedding (P/model/vilmodel_goat.py:470-483)
    img = aggregate_gmap_features(split_embeds, split_lens, batch['traj_vpids'], batch['traj_cand_vpids'],
                                  batch['gmap_vpids'], split_fused)
    e = img + sd[pre + '.gmap_step_embeddings.weight'][batch['gmap_step_ids']] \
        + _ln(sd, pre + '.gmap_pos_embeddings.1', _lin(sd, pre + '.gmap_pos_embeddings.0', batch['gmap_pos_fts']), 1e-12)
    return e, gen_seq_masks(batch['gmap_lens'])


def vp_input_embedding(ctx, sd, pre, batch, split_embeds, split_lens):
    # LocalVPEncoder.vp_input_embedding (P/model/vilmodel_goat.py:377-391)
    vp_img = pad_tensors_wgrad([x[-1] for x in split_embeds])
    vp_lens = torch.stack([x[-1] + 1 for x in split_lens], 0)
    vp_masks = gen_seq_masks(vp_lens)
    max_vp_len = int(max(vp_lens))
    vp_img = torch.cat([vp_img.new_zeros(vp_img.shape[0], 1, vp_img.shape[2]), vp_img], 1)[:, :max_vp_len]
    e = vp_img + _ln(sd, pre + '.vp_pos_embeddings.1', _lin(sd, pre + '.vp_pos_embeddings.0', batch['vp_pos_fts']), 1e-12)
    return e, vp_masks


def _sprels(sd, pre, batch):
    # P/model/vilmodel_goat.py:496-499
    w, b = sd[pre + '.sprel_linear.weight'], sd[pre + '.sprel_linear.bias']
    return (batch['gmap_pair_dists'] * w.view(()) + b.view(())).unsqueeze(1)


# ----------------------------------------------------------------------------- GlocalTextPathCMT
def _text(ctx, sd, batch):
    txt_masks = gen_seq_masks(batch['txt_lens'])
    e = embeddings(ctx, sd, 'bert.embeddings', batch['txt_ids'])
    if getattr(ctx.cfg, 'do_back_txt', False):
        return lang_encoder_do(ctx, sd, 'bert.lang_encoder', e, txt_masks, batch), txt_masks
    return lang_encoder(ctx, sd, 'bert.lang_encoder', e, txt_masks), txt_masks


def bert_forward(ctx, sd, batch, return_gmap=True):
    # GlocalTextPathCMT.forward (P/model/vilmodel_goat.py:546-594)
    txt, txt_masks = _text(ctx, sd, batch)
    se, sl, sf = image_embeddings(ctx, sd, 'bert.img_embeddings', batch)
    gmap = None
    if return_gmap:
        g, gm = gmap_input_embedding(ctx, sd, 'bert.global_encoder', batch, se, sl, sf)
        gmap = crossmodal_encoder(ctx, sd, 'bert.global_encoder.encoder', g, gm, txt, txt_masks,
                                  _sprels(sd, 'bert.global_encoder', batch) if ctx.cfg.graph_sprels else None)
    v, vm = vp_input_embedding(ctx, sd, 'bert.local_encoder', batch, se, sl)
    vp = crossmodal_encoder(ctx, sd, 'bert.local_encoder.encoder', v, vm, txt, txt_masks)
    return gmap, vp, txt


def bert_forward_mlm(ctx, sd, batch):
    # GlocalTextPathCMT.forward_mlm (P/model/vilmodel_goat.py:597-648)
    txt, txt_masks = _text(ctx, sd, batch)
    ext_txt = extend_neg_masks(txt_masks)
    se, sl, sf = image_embeddings(ctx, sd, 'bert.img_embeddings', batch)
    g, gm = gmap_input_embedding(ctx, sd, 'bert.global_encoder', batch, se, sl, sf)
    gt = crossmodal_encoder(ctx, sd, 'bert.global_encoder.encoder', txt, ext_txt, g, extend_neg_masks(gm))
    v, vm = vp_input_embedding(ctx, sd, 'bert.local_encoder', batch, se, sl)
    vt = crossmodal_encoder(ctx, sd, 'bert.local_encoder.encoder', txt, ext_txt, v, extend_neg_masks(vm))
    return gt + vt


def bert_forward_cfp(ctx, sd, batch):
    # GlocalTextPathCMT.forward_cfp (P/model/vilmodel_goat.py:650-696); sprel bias computed but unused (:517-525)
    txt, _ = _text(ctx, sd, batch)
    se, sl, sf = image_embeddings(ctx, sd, 'bert.img_embeddings', batch)
    g, gm = gmap_input_embedding(ctx, sd, 'bert.global_encoder', batch, se, sl, sf)
    gmap = bert_attention(ctx, sd, 'bert.global_encoder.tim_self_encoder', g, extend_neg_masks(gm))
    v, vm = vp_input_embedding(ctx, sd, 'bert.local_encoder', batch, se, sl)
    vp = bert_attention(ctx, sd, 'bert.local_encoder.tim_self_encoder', v, extend_neg_masks(vm))
    return gmap, vp, txt


# ----------------------------------------------------------------------------- heads
def cls_prediction(sd, pre, x):
    # ClsPrediction (P/model/pretrain_goat.py:27-38)
    h = _ln(sd, pre + '.net.2', F.relu(_lin(sd, pre + '.net.0', x)), 1e-12)
    return _lin(sd, pre + '.net.3', h)


def head_transform(ctx, sd, pre, x):
    # BertPredictionHeadTransform (P/model/Bert_backbone.py:797-811)
    return _ln(sd, pre + '.LayerNorm', gelu(_lin(sd, pre + '.dense', x)), ctx.cfg.layer_norm_eps)


def forward_mlm(ctx, sd, batch, compute_loss=True):
    # GlocalTextPathCMTPreTraining.forward_mlm (P/model/pretrain_goat.py:188-224)
    txt = bert_forward_mlm(ctx, sd, batch)
    labels = batch['txt_labels']
    masked = txt[labels != -1]
    h = head_transform(ctx, sd, 'mlm_head.predictions.transform', masked)
    scores = F.linear(h, sd['bert.embeddings.word_embeddings.weight']) + sd['mlm_head.predictions.bias']
    if compute_loss:
        return F.cross_entropy(scores, labels[labels != -1], reduction='none')
    return scores


def _fuse_weights(sd, gmap_embeds, vp_embeds):
    return torch.sigmoid(cls_prediction(sd, 'sap_fuse_linear', torch.cat([gmap_embeds[:, 0], vp_embeds[:, 0]], 1)))


def forward_sap(ctx, sd, batch, compute_loss=True):
    # GlocalTextPathCMTPreTraining.forward_sap (P/model/pretrain_goat.py:286-354)
    gmap, vp, _ = bert_forward(ctx, sd, batch)
    B = gmap.shape[0]
    fw = _fuse_weights(sd, gmap, vp) if ctx.cfg.glocal_fuse else 0.5
    gl = cls_prediction(sd, 'global_sap_head', gmap).squeeze(2) * fw
    gl = gl.masked_fill(batch['gmap_visited_masks'], -float('inf'))
    gl = gl.masked_fill(gen_seq_masks(batch['gmap_lens']).logical_not(), -float('inf'))
    ll = cls_prediction(sd, 'local_sap_head', vp).squeeze(2) * (1 - fw)
    nav = pad_tensors_wgrad([x[-1] != 1 for x in torch.split(batch['traj_nav_types'], batch['traj_step_lens'])])
    nav = nav[:, :ll.size(1) - 1]
    nav = torch.cat([torch.zeros(len(nav), 1, dtype=torch.bool, device=nav.device), nav], 1)
    ll = ll.masked_fill(nav, -float('inf'))
    # fusion (:329-345): local candidate logits scattered into global slots by viewpoint id
    fused = gl.clone()
    rows = [[fused[i, j] for j in range(fused.shape[1])] for i in range(B)]
    for i in range(B):
        rows[i][0] = rows[i][0] + ll[i, 0]
        visited = set(vp_ for vp_, m in zip(batch['gmap_vpids'][i], batch['gmap_visited_masks'][i]) if m)
        tmp, bw = {}, 0
        for j, c in enumerate(batch['traj_cand_vpids'][i][-1]):
            if c in visited:
                bw = bw + ll[i, j + 1]
            else:
                tmp[c] = ll[i, j + 1]
        for j, vp_ in enumerate(batch['gmap_vpids'][i]):
            if j > 0 and vp_ not in visited:
                rows[i][j] = rows[i][j] + (tmp[vp_] if vp_ in tmp else bw)
    fused = torch.stack([torch.stack(r) for r in rows])
    if compute_loss:
        ga, la = batch['global_act_labels'], batch['local_act_labels']
        return F.cross_entropy(gl, ga, reduction='none') + F.cross_entropy(ll, la, reduction='none') \
            + F.cross_entropy(fused, ga, reduction='none')
    return gl, ll, fused


def attn_pool(x, w):
    # P/model/pretrain_goat.py:502-515  (softmax over all slots, no padding mask)
    a = torch.softmax(torch.matmul(torch.tanh(x), w), 1)
    return torch.tanh(torch.sum(x * a, 1))


def cfp_losses(gmap_o, vp_o, fused_o, txt_o, temperature, target=None, txt_all=None, img_offset=0):
    """Symmetric InfoNCE x3 (P/model/pretrain_goat.py:519-534).  With `txt_all`/`target` given it is the
    cross-rank (all-gathered negatives) extension: rows are local, columns are the gathered set."""
    B = gmap_o.shape[0]
    if target is None:
        target = torch.arange(B, device=gmap_o.device)

    def sym(x):
        sim = (x @ txt_o.T) / temperature
        return (F.cross_entropy(sim, target, reduction='none') + F.cross_entropy(sim.T, target, reduction='none')) / 2.0
    return sym(gmap_o) + sym(vp_o) + sym(fused_o)


def forward_cfp(ctx, sd, batch, compute_loss=True):
    # GlocalTextPathCMTPreTraining.forward_cfp (P/model/pretrain_goat.py:467-541)
    gmap, vp, txt = bert_forward_cfp(ctx, sd, batch)
    if batch.get('extra_heads'):
        gmap = head_transform(ctx, sd, 'tim_global_head', gmap)
        vp = head_transform(ctx, sd, 'tim_local_head', vp)
        txt = head_transform(ctx, sd, 'tim_txt_head', txt)
    fw = _fuse_weights(sd, gmap, vp) if ctx.cfg.glocal_fuse else 0.5
    go = attn_pool(gmap, sd['tim_global_attn'])
    vo = attn_pool(vp, sd['tim_local_attn'])
    to = attn_pool(txt, sd['tim_txt_attn'])
    fo = go * fw + vo * (1 - fw)
    if compute_loss:
        return cfp_losses(go, vo, fo, to, ctx.cfg.cfp_temperature)
    return go, vo, fo, to


def _last_step(batch, key):
    return [x[-1] for x in torch.split(batch[key], batch['traj_step_lens'], 0)]


def forward_og(ctx, sd, batch, compute_loss=True):
    # GlocalTextPathCMTPreTraining.forward_og (P/model/pretrain_goat.py:356-391)
    _, vp, _ = bert_forward(ctx, sd, batch, return_gmap=False)
    view_lens, obj_lens = _last_step(batch, 'traj_vp_view_lens'), _last_step(batch, 'traj_vp_obj_lens')
    obj = pad_tensors_wgrad([x[1 + vl:1 + vl + ol] for x, vl, ol in zip(vp, view_lens, obj_lens)])
    obj_masks = gen_seq_masks(torch.stack(obj_lens, 0))
    logits = cls_prediction(sd, 'og_head', obj).squeeze(2).masked_fill(obj_masks.logical_not(), -float('inf'))
    if compute_loss:
        return F.cross_entropy(logits, batch['obj_labels'], reduction='none')
    return logits


def region_classification(sd, pre, x):
    # RegionClassification (P/model/pretrain_goat.py:14-25)
    h = _ln(sd, pre + '.net.2', F.relu(_lin(sd, pre + '.net.0', x)), 1e-12)
    return _lin(sd, pre + '.net.3', h)


def _masked_hidden(hidden, mask):
    # _compute_masked_hidden (P/model/pretrain_goat.py:543-547)
    return hidden[mask.unsqueeze(-1).expand_as(hidden)].contiguous().view(-1, hidden.size(-1))


def forward_mrc(ctx, sd, batch, compute_loss=True):
    # GlocalTextPathCMTPreTraining.forward_mrc (P/model/pretrain_goat.py:226-284)
    _, vp, _ = bert_forward(ctx, sd, batch, return_gmap=False)
    view_lens = _last_step(batch, 'traj_vp_view_lens')
    view = pad_tensors_wgrad([x[1:vl + 1] for x, vl in zip(vp, view_lens)])
    v_pred = region_classification(sd, 'image_classifier', _masked_hidden(view, batch['vp_view_mrc_masks']))
    v_tgt = _masked_hidden(batch['vp_view_probs'], batch['vp_view_mrc_masks'])
    o_pred = o_tgt = None
    if batch['traj_obj_img_fts'] is not None:
        obj_lens = _last_step(batch, 'traj_vp_obj_lens')
        obj = pad_tensors_wgrad([x[vl + 1:vl + ol + 1] for x, vl, ol in zip(vp, view_lens, obj_lens)])
        head = 'obj_classifier' if ('obj_classifier.net.0.weight' in sd) else 'image_classifier'
        o_pred = region_classification(sd, head, _masked_hidden(obj, batch['vp_obj_mrc_masks']))
        o_tgt = _masked_hidden(batch['vp_obj_probs'], batch['vp_obj_mrc_masks'])
    if not compute_loss:
        return v_pred, v_tgt, o_pred, o_tgt
    loss = F.kl_div(F.log_softmax(v_pred, dim=-1), v_tgt, reduction='none').sum(dim=1)
    if o_pred is not None:
        loss = torch.cat([loss, F.kl_div(F.log_softmax(o_pred, dim=-1), o_tgt, reduction='none').sum(dim=1)], 0)
    return loss


def forward(cfg, sd, batch, task, compute_loss=True, training=False):
    """GlocalTextPathCMTPreTraining.forward dispatch (P/model/pretrain_goat.py:91-186)."""
    ctx = Ctx(cfg, training)
    batch = defaultdict(lambda: None, batch)
    if task.startswith('mlm'):
        return forward_mlm(ctx, sd, batch, compute_loss)
    if task.startswith('mrc'):
        return forward_mrc(ctx, sd, batch, compute_loss)
    if task.startswith('sap'):
        return forward_sap(ctx, sd, batch, compute_loss)
    if task.startswith('og'):
        return forward_og(ctx, sd, batch, compute_loss)
    if task.startswith('cfp'):
        return forward_cfp(ctx, sd, batch, compute_loss)
    raise ValueError('invalid task')
