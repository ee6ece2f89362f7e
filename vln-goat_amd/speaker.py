"""The "transpeaker" — the transformer speaker of the fine-tuning loop's back-translation augmentation (last item):
`Transpeaker` of M/models/transpeaker_model.py:232-257 (M = /root/reference/map_nav_src) on the HIP kernels, with the reference's
`state_dict` keys, plus the two uses of it in M/r2r/transpeaker.py: the teacher-forced loss (:209-246) and greedy / sampled decoding
(`infer_batch`, :248-318), and the feature walk along the ground-truth path (`from_shortest_path`, :158-199) on rollout.GraphSim.

Reference behaviour kept (file:line in M/models/transpeaker_model.py):
  * MultiHeadAttention (:91-119): bias-free projections, heads of size `aemb` (64: the head size of goat_attn_*), boolean masks filled
    with -1e9 BEFORE the softmax (a fully masked row becomes uniform, not NaN), `LayerNorm(out + residual)` with a LayerNorm built
    inside forward — never trained: unit gain, zero shift, eps 1e-5 — then dropout;
  * the attention-probability dropout lives in a module created inside forward (:98,114) and is therefore ACTIVE IN EVAL MODE too
    (p = speaker_dropout); reproduced: `attn_dropout_always`.  Q / K / V dropout only with `use_drop` (:110-113);
  * PositionalEncoding (:32-48): sinusoid table added over the sequence axis, dropout 0.1;
  * PoswiseFeedForwardNet (:121-134): Linear - ReLU - Dropout - Linear without biases, LayerNorm(out + x) as above;
  * encoder (:158-198): per step the action feature [F] attends over the 36 view features [36, F] (one query, no mask), the T step
    vectors then pass `speaker_layer_num` encoder layers WITHOUT a padding mask; decoder (:200-230): pad + causal self-attention
    mask, optional context mask, weight-tied to nothing (projection is its own Linear).
"""
import math

import numpy as np
import torch
from torch import nn

from . import hipops
from .layers import compute_dtype, _p


def _sinusoid(max_len, d_model):
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1)          # [max_len, 1, d_model] (the reference's buffer shape: state_dict key `pe`)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.register_buffer('pe', _sinusoid(max_len, d_model))

    def forward(self, x):
        """x [B, L, D] (batch first here; the reference transposes around the call)."""
        y = x + self.pe[:x.shape[1], 0].to(x.dtype).unsqueeze(0)
        return hipops.dropout(y, _p(self.dropout))


class _NoAffineLN(nn.Module):
    """nn.LayerNorm(hidden)(x) of a module built inside forward (never trained): gain 1, shift 0, eps 1e-5."""

    def __init__(self, hidden):
        super().__init__()
        self.register_buffer('one', torch.ones(hidden), persistent=False)
        self.register_buffer('zero', torch.zeros(hidden), persistent=False)

    def forward(self, x, residual, p=0.0):
        return hipops.layer_norm(x, self.one, self.zero, 1e-5, residual=residual)


class MultiHeadAttention(nn.Module):
    def __init__(self, q_hidden, k_hidden, n_heads, cfg):
        super().__init__()
        d_k = cfg.aemb
        if d_k != 64:
            raise NotImplementedError('GOAT HIP attention kernels are specialised for head size 64 (aemb)')
        self.n_heads, self.hidden = n_heads, q_hidden
        self.W_Q = nn.Linear(q_hidden, d_k * n_heads, bias=False)
        self.W_K = nn.Linear(k_hidden, d_k * n_heads, bias=False)
        self.W_V = nn.Linear(k_hidden, d_k * n_heads, bias=False)
        self.fc = nn.Linear(n_heads * d_k, q_hidden, bias=False)
        self.dropout = nn.Dropout(cfg.speaker_dropout)
        self.ln = _NoAffineLN(q_hidden)
        self.use_drop = bool(getattr(cfg, 'use_drop', False))
        self.attn_p = float(cfg.speaker_dropout) if getattr(cfg, 'attn_dropout_always', True) else None

    def forward(self, xq, xkv, bias=None):
        """xq [B, Lq, q_hidden], xkv [B, Lk, k_hidden]; bias: float32 [B, Lq, Lk] additive mask (0 / -1e9) or None."""
        q = hipops.linear(xq, self.W_Q.weight)
        k = hipops.linear(xkv, self.W_K.weight)
        v = hipops.linear(xkv, self.W_V.weight)
        if self.use_drop:
            q, k, v = (hipops.dropout(t, _p(self.dropout)) for t in (q, k, v))
        p = self.attn_p if self.attn_p is not None else _p(self.dropout)       # (the reference's always-on attention dropout)
        ctx = hipops.attention(q, torch.cat([k, v], -1), None, bias, self.n_heads, p)
        out = self.ln(hipops.linear(ctx, self.fc.weight), xq)
        return hipops.dropout(out, _p(self.dropout))


class PoswiseFeedForwardNet(nn.Module):
    def __init__(self, hidden, cfg):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(hidden, cfg.proj_hidden, bias=False), nn.ReLU(), nn.Dropout(cfg.speaker_dropout),
                                nn.Linear(cfg.proj_hidden, hidden, bias=False))
        self.ln = _NoAffineLN(hidden)

    def forward(self, x):
        h = hipops.linear(x, self.fc[0].weight, None, 'relu')
        h = hipops.dropout(h, _p(self.fc[2]))
        return self.ln(hipops.linear(h, self.fc[3].weight), x)


class EncoderLayer(nn.Module):
    def __init__(self, hidden, n_heads, cfg):
        super().__init__()
        self.enc_self_attn = MultiHeadAttention(hidden, hidden, n_heads, cfg)
        self.pos_ffn = PoswiseFeedForwardNet(hidden, cfg)

    def forward(self, x):
        return self.pos_ffn(self.enc_self_attn(x, x))


class DecoderLayer(nn.Module):
    def __init__(self, q_size, k_size, n_heads, cfg):
        super().__init__()
        self.dec_self_attn = MultiHeadAttention(q_size, q_size, n_heads, cfg)
        self.dec_enc_attn = MultiHeadAttention(q_size, k_size, n_heads, cfg)
        self.pos_ffn = PoswiseFeedForwardNet(q_size, cfg)

    def forward(self, x, enc, self_bias, enc_bias):
        x = self.dec_self_attn(x, x, self_bias)
        x = self.dec_enc_attn(x, enc, enc_bias)
        return self.pos_ffn(x)


class TranspeakerEncoder(nn.Module):
    def __init__(self, feature_size, hidden, cfg):
        super().__init__()
        self.hidden_size, self.feature_size, self.image_feat_size = hidden, feature_size, cfg.image_feat_size
        self.pos_emb = PositionalEncoding(hidden)
        self.drop_feat = nn.Dropout(p=cfg.featdropout)
        self.drop = nn.Dropout(p=cfg.speaker_dropout)
        self.down_size = nn.Linear(feature_size, hidden)
        self.layers = nn.ModuleList([EncoderLayer(hidden, cfg.speaker_head_num, cfg) for _ in range(cfg.speaker_layer_num)])
        self.image_self_attn = MultiHeadAttention(hidden, feature_size, cfg.speaker_head_num, cfg)

    def _drop_image_part(self, x):
        p = _p(self.drop_feat)
        if p <= 0:
            return x
        D = self.image_feat_size
        return torch.cat([hipops.dropout(x[..., :D].contiguous(), p), x[..., D:]], -1)

    def forward(self, action_inputs, feature_inputs, already_dropfeat=False):
        """action_inputs [B, T, F] (the view taken at each step + its angle feature), feature_inputs [B, T, 36, F]."""
        dt = compute_dtype()
        B, T, F = action_inputs.shape
        a, f = action_inputs.to(dt), feature_inputs.to(dt)
        if not already_dropfeat:
            a, f = self._drop_image_part(a), self._drop_image_part(f)
        ctx = hipops.linear(a, self.down_size.weight, self.down_size.bias).reshape(B * T, 1, self.hidden_size)
        enc_inputs = self.image_self_attn(ctx, f.reshape(B * T, 36, F)).view(B, T, self.hidden_size)
        x = self.pos_emb(enc_inputs)
        for layer in self.layers:
            x = layer(x)
        return enc_inputs, x


class TranspeakerDecoder(nn.Module):
    def __init__(self, vocab, word_size, hidden, padding_idx, cfg):
        super().__init__()
        self.embedding = nn.Embedding(vocab, word_size, padding_idx)
        self.pos_emb = PositionalEncoding(word_size)
        self.layers = nn.ModuleList([DecoderLayer(word_size, hidden, cfg.speaker_head_num, cfg) for _ in range(cfg.speaker_layer_num)])
        self.drop = nn.Dropout(p=cfg.speaker_dropout)
        self.use_drop = bool(getattr(cfg, 'use_drop', False))

    def forward(self, dec_inputs, enc_outputs, ctx_mask=None):
        """dec_inputs int64 [B, L]; ctx_mask bool [B, T] (True = padded context step) or None."""
        dt = compute_dtype()
        ids = dec_inputs.to(torch.int64)
        B, L = ids.shape
        x = hipops.embedding(ids, self.embedding.weight, out_dtype=dt, word_pad=self.embedding.padding_idx)
        if self.use_drop:
            x = hipops.dropout(x, _p(self.drop))
        x = self.pos_emb(x)
        # pad (key == 0) or future position -> -1e9 (:212-214)
        masked = ids.eq(0).unsqueeze(1) | torch.triu(torch.ones(L, L, dtype=torch.bool, device=ids.device), 1).unsqueeze(0)
        self_bias = torch.zeros(B, L, L, dtype=torch.float32, device=ids.device).masked_fill_(masked, -1e9)
        enc_bias = None
        if ctx_mask is not None:
            T = ctx_mask.shape[1]
            enc_bias = torch.zeros(B, L, T, dtype=torch.float32, device=ids.device).masked_fill_(ctx_mask.bool().unsqueeze(1), -1e9)
        enc = enc_outputs.to(dt)
        for layer in self.layers:
            x = layer(x, enc, self_bias, enc_bias)
        return x


class Transpeaker(nn.Module):
    """cfg: an object with h_dim, wemb, aemb, proj_hidden, speaker_layer_num, speaker_head_num, speaker_dropout, featdropout,
    image_feat_size, use_drop (the fields of M/r2r/parser.py:104-118 the reference reads at import time)."""

    def __init__(self, feature_size, hidden_size, word_size, tgt_vocab_size, cfg, padding_idx=0):
        super().__init__()
        self.encoder = TranspeakerEncoder(feature_size, hidden_size, cfg)
        self.decoder = TranspeakerDecoder(tgt_vocab_size, word_size, hidden_size, padding_idx, cfg)
        self.projection = nn.Linear(word_size, tgt_vocab_size, bias=False)
        self.dropout = nn.Dropout(cfg.speaker_dropout)
        self.use_drop = bool(getattr(cfg, 'use_drop', False))

    def forward(self, action_embeddings, world_state_embeddings, dec_inputs, ctx_mask=None, already_dropfeat=False):
        enc_inputs, enc_outputs = self.encoder(action_embeddings, world_state_embeddings, already_dropfeat)
        dec = self.decoder(dec_inputs, enc_outputs, ctx_mask)
        if self.use_drop:
            dec = hipops.dropout(dec, _p(self.dropout))
        return hipops.linear(dec, self.projection.weight, None, None, torch.float32)      # float32 logits [B, L, V]


def default_config(**over):
    from types import SimpleNamespace
    base = dict(h_dim=512, wemb=256, aemb=64, proj_hidden=1024, speaker_layer_num=3, speaker_head_num=4, speaker_dropout=0.2,
                featdropout=0.3, image_feat_size=768, speaker_angle_size=128, use_drop=False, maxDecode=120, attn_dropout_always=True)
    base.update(over)
    return SimpleNamespace(**base)


# ------------------------------------------------------------------------------------------------ the two uses (M/r2r/transpeaker.py)
def teacher_forcing_loss(model, can_feats, img_feats, insts, pad_id=0, ctx_mask=None):
    """M/r2r/transpeaker.py:233-239: cross-entropy of logits[:, :-1] against insts[:, 1:], <PAD> ignored, mean over the rest."""
    logits = model(can_feats, img_feats, insts, ctx_mask=ctx_mask)
    V = logits.shape[-1]
    return torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, V), insts[:, 1:].reshape(-1).to(torch.int64), ignore_index=pad_id)


@torch.no_grad()
def infer_batch(model, can_feats, img_feats, bos, eos, pad, unk, max_decode=120, sampling=False, already_dropfeat=False):
    """M/r2r/transpeaker.py:248-318: encode once, then decode word by word (the decoder re-reads the whole prefix every step, as
    the reference does), <UNK> never produced, finished sentences padded; -> int64 [B, <= max_decode + 1] including <BOS>."""
    enc_inputs, enc_outputs = model.encoder(can_feats, img_feats, already_dropfeat)
    B = can_feats.shape[0]
    word = torch.full((B, 1), bos, dtype=torch.int64, device=can_feats.device)
    ended = torch.zeros(B, dtype=torch.bool, device=can_feats.device)
    for _ in range(max_decode):
        dec = model.decoder(word, enc_outputs)
        logits = hipops.linear(dec[:, -1:].contiguous(), model.projection.weight, None, None, torch.float32)[:, 0]
        logits[:, unk] = -float('inf')
        nxt = torch.distributions.Categorical(logits=logits).sample() if sampling else logits.argmax(-1)
        nxt = torch.where(ended, torch.full_like(nxt, pad), nxt)
        word = torch.cat([word, nxt.unsqueeze(1)], 1)
        ended = ended | (nxt == eos)
        if bool(ended.all()):
            break
    return word


def path_features(sim, store, episodes, angle_size=128):
    """`from_shortest_path` (M/r2r/transpeaker.py:158-199) on the graph-only navigator: walk every ground-truth path; per step the
    36 view features + their relative angle features (`speaker_feature` of the observation, M/r2r/env.py:362-364) and the feature of
    the view in which the next viewpoint is seen (zeros at the stop step).  -> (img_feats [B, T, 36, F], can_feats [B, T, F]
    float32 on the store's device, lengths [B])."""
    from . import rollout
    obs = sim.reset(episodes)
    B = len(obs)
    D = store.table.shape[1]
    F = D + angle_size
    table = rollout.view_angle_feature_table(angle_size)
    ended = np.zeros(B, bool)
    lengths = np.zeros(B, np.int64)
    rows, angs, crow, cang = [], [], [], []
    while not ended.all():
        r = np.stack([ob['feature_row'] * 36 + np.arange(36) for ob in obs])
        a = np.stack([table[ob['viewIndex']] for ob in obs])
        cr = np.full(B, -1, np.int64)
        ca = np.zeros((B, angle_size), np.float32)
        moves = []
        for i, ob in enumerate(obs):
            path = ob['gt_path']
            nxt = None
            if not ended[i] and ob['viewpoint'] in path:
                k = path.index(ob['viewpoint'])
                nxt = path[k + 1] if k + 1 < len(path) else None
            if nxt is None:
                moves.append(None)
            else:
                c = next(c for c in ob['candidate'] if c['viewpointId'] == nxt)
                cr[i] = ob['feature_row'] * 36 + c['pointId']
                ca[i] = rollout.angle_feature(c['heading'], c['elevation'], angle_size)
                moves.append((nxt, c['pointId']))
        rows.append(r)
        angs.append(a)
        crow.append(cr)
        cang.append(ca)
        lengths += (~ended)
        ended = np.logical_or(ended, np.array([m is None for m in moves]))
        obs = sim.step(moves)
    dev = store.dev.device
    rows_t = torch.from_numpy(np.stack(rows, 1)).to(dev)                       # [B, T, 36]
    img = torch.cat([store.gather(rows_t).float(), torch.from_numpy(np.stack(angs, 1)).to(dev)], -1)
    crow_t = torch.from_numpy(np.stack(crow, 1)).to(dev)                       # [B, T], -1 at the stop step
    can = torch.cat([store.gather(crow_t).float(), torch.from_numpy(np.stack(cang, 1)).to(dev)], -1)
    return img, can, lengths
