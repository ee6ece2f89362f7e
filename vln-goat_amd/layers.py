"""BERT / RoBERTa / DETR-style blocks of GOAT with HIP forward+backward bodies.

Module tree, attribute names and parameter shapes follow the reference exactly (state_dict keys are a
drop-in contract); only the `forward` bodies differ: they call the hand-written gfx950
kernels through `hipops`.  Reference classes mirrored here (P/ = pretrain_src/, M/ = map_nav_src/):
  RobertaEmbeddings          P/model/Bert_backbone.py:56-121
  BertSelfAttention/-Output  P/model/Bert_backbone.py:157-310   (Roberta* twins :373-543)
  BertAttention              P/model/Bert_backbone.py:313-342
  BertIntermediate/-Output   P/model/Bert_backbone.py:345-370
  RobertaLayer               P/model/Bert_backbone.py:574-659
  BertCrossLayer             P/model/Bert_backbone.py:661-754
  CrossmodalEncoder          P/model/Bert_backbone.py:756-781
  BertPredictionHeadTransform / BertLMPredictionHead / BertOnlyMLMHead   :797-838
  TransformerEncoder(Layer)  P/model/transformer.py:62-89,133-191 (pre-LN, nn.MultiheadAttention)
"""
import os

import weakref

import torch
from torch import nn

from . import hipops

_COMPUTE_DTYPE = [torch.float32]


def set_compute_dtype(dtype):
    """torch.float32 (exact-f32 MFMA parity mode) or torch.bfloat16 (bf16 storage, f32 accumulate)."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError('compute dtype must be float32 or bfloat16')
    _COMPUTE_DTYPE[0] = dtype


def compute_dtype():
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_gpu_dtype()
        if dt == torch.bfloat16:
            return torch.bfloat16
        raise RuntimeError('GOAT HIP path supports bfloat16 autocast only (got %s)' % dt)
    return _COMPUTE_DTYPE[0]


# Masks depend on the batch's length tensors only; the blocks of a model ask for the same mask many times per step (every
# cross-modal layer converts the same bool mask into its additive form: 21 tiny ATen launches per pre-training step).  They are
# memoised per INPUT TENSOR OBJECT (identity checked through a weak reference, so a recycled address cannot alias, and the
# tensor's version counter, so an in-place edit invalidates) — results are read-only by convention.
# A captured step reads the memoised tensors at fixed addresses: after new data has been copied into a static batch
# (train_step.StaticBatch.commit) `refresh_masks()` recomputes the stale entries IN PLACE, sources before derived masks.
_MASK_MEMO = {}
_MEMO_LIMIT = [512]
_NO_MEMO = bool(os.environ.get('GOAT_NO_MASK_MEMO'))       # (diagnostics)


def _memo(tag, t, extra, compute):
    if _NO_MEMO:
        return compute(t)
    key = (tag, id(t), extra)
    ent = _MASK_MEMO.get(key)
    if ent is not None and ent[0]() is t and ent[1] == t._version:
        return ent[2]
    r = compute(t)
    if len(_MASK_MEMO) > _MEMO_LIMIT[0]:
        _prune_memo()
        _MEMO_LIMIT[0] = max(512, 2 * len(_MASK_MEMO))      # (every entry alive: look again after the memo has doubled)
    _MASK_MEMO[key] = [weakref.ref(t), t._version, r, compute]
    return r


def _prune_memo():
    """Drop the entries whose source tensor is gone — and only those.  An entry of a LIVE tensor may be read by a captured step at
    its fixed address (and must stay findable for refresh_masks()), so it is never evicted: the memo is bounded by the number of
    live batch tensors, not by a count."""
    for key in list(_MASK_MEMO):
        if _MASK_MEMO[key][0]() is None:
            del _MASK_MEMO[key]


def refresh_masks(only=None):
    """recompute, into the same storage, every memoised mask whose source tensor was edited in place (insertion order: a mask
    derived from another memoised mask is refreshed after it).  Entries of dead tensors are dropped.
    only: the edited source tensors (views of one buffer share a version counter: after a PARTIAL copy into an EpisodeBuffers every
    view looks edited) — just their masks, and the masks derived from those, are recomputed."""
    ids = None if only is None else {id(t) for t in only}
    for key in list(_MASK_MEMO):
        ent = _MASK_MEMO[key]
        t = ent[0]()
        if t is None:
            del _MASK_MEMO[key]
        elif ent[1] != t._version and (ids is None or id(t) in ids):
            ent[2].copy_(ent[3](t))
            ent[1] = t._version
            if ids is not None:
                ids.add(id(ent[2]))


def neg_mask(masks, value=-10000.0):
    """bool [N,L] -> additive float32 key mask [N,L] (P/model/ops.py:25-34 without the broadcast dims)."""
    return _memo('neg', masks, value, lambda m: (1.0 - m.float()) * value)


def inf_mask(masks):
    """bool [N,L] (True = valid) -> 0 / -inf (nn.MultiheadAttention key_padding_mask semantics)."""
    return _memo('inf', masks, None,
                 lambda m: torch.zeros(m.shape, dtype=torch.float32, device=m.device).masked_fill(~m, float('-inf')))


def gen_seq_masks(seq_lens, max_len=None):
    # P/model/ops.py:36-44
    if max_len is None:
        max_len = int(seq_lens.max())
    max_len = int(max_len)
    return _memo('seq', seq_lens, max_len,
                 lambda sl: torch.arange(max_len, device=sl.device).unsqueeze(0) < sl.unsqueeze(1))


class Linear(nn.Linear):
    def forward(self, x, act=None):
        return hipops.linear(x, self.weight, self.bias, act)


_NO_FORK_IN = bool(os.environ.get('GOAT_NO_LN_FORK_IN'))
_NO_FORK = bool(os.environ.get('GOAT_NO_LN_FORK'))      # (diagnostics: A/B of the forked LayerNorm outputs)
_NO_ZOUT = bool(os.environ.get('GOAT_NO_LN_ZOUT'))      # (diagnostics: the pre-LN residual junctions as kernels of their own)


class LayerNorm(nn.LayerNorm):
    def forward(self, x, residual=None, p=0.0, fork=False, fork_in=False, z_out=False, p_out=0.0, post_add=None):
        """p_out: dropout on the output, in the same launch (dropout(LayerNorm(x)) of the embedding blocks); post_add: a summand added
        behind the norm, in front of that dropout."""
        if (p_out or post_add is not None) and _NO_ZOUT:  # (diagnostics: the add / the output dropout as kernels of their own)
            y = self.forward(x, residual, p, fork, fork_in, z_out)
            y = y if post_add is None else post_add + y
            return hipops.dropout(y, p_out)
        if fork_in and (_NO_FORK or _NO_FORK_IN):
            return hipops.layer_norm(x, self.weight, self.bias, self.eps), x
        if z_out and (_NO_FORK or _NO_FORK_IN or _NO_ZOUT):          # (diagnostics: the junction as its own kernel again)
            z = hipops.dropout_add(x, residual, p)
            return hipops.layer_norm(z, self.weight, self.bias, self.eps), z
        return hipops.layer_norm(x, self.weight, self.bias, self.eps, residual, p, fork and not _NO_FORK, fork_in, z_out, p_out, post_add)


def _pair(h):
    """A hidden state travels between post-LN sub-layers as (for the next Linear, for the next residual add): two autograd
    handles on one buffer (hipops.layer_norm(fork=True)), so their gradients are summed inside the LayerNorm backward."""
    return h if isinstance(h, tuple) else (h, h)


BertLayerNorm = LayerNorm


def _p(drop):
    """dropout probability of an nn.Dropout at call time (the reference's set_dropout mutates `.p`)."""
    return drop.p if drop.training else 0.0


def _check_act(config):
    if getattr(config, 'hidden_act', 'gelu') != 'gelu':
        raise NotImplementedError('GOAT HIP path implements hidden_act="gelu" (erf) only')


class RobertaEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        pad = getattr(config, 'pad_token_id', None)
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=pad)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size, padding_idx=pad)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer('position_ids', torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.register_buffer('token_type_ids', torch.zeros(self.position_ids.size(), dtype=torch.long), persistent=False)
        self.padding_idx = pad

    def forward(self, input_ids, token_type_ids=None):
        # one gather-sum kernel for the three tables; position ids = arange(L) for every sample (:98-100)
        e = hipops.embedding(input_ids, self.word_embeddings.weight, self.token_type_embeddings.weight, token_type_ids,
                             self.position_embeddings.weight, out_dtype=compute_dtype(),
                             word_pad=self.word_embeddings.padding_idx, pos_pad=self.position_embeddings.padding_idx)
        return self.LayerNorm(e, p_out=_p(self.dropout))       # dropout(LayerNorm(e)) in one launch per direction


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        if self.attention_head_size != 64:
            raise NotImplementedError('GOAT HIP attention kernels are specialised for head_dim 64')
        self.all_head_size = config.hidden_size
        self.query = Linear(config.hidden_size, self.all_head_size)
        self.key = Linear(config.hidden_size, self.all_head_size)
        self.value = Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def project_kv(self, enc_hidden):
        """K|V projection of the attended sequence, [B, Lk, 2H].  Depends on enc_hidden only: a caller that attends to the SAME
        sequence many times (the instruction in every step of a navigation episode) computes it once and passes it as enc_kv."""
        return hipops.multi_linear(enc_hidden, [self.key.weight, self.value.weight], [self.key.bias, self.value.bias])

    def forward(self, hidden, kmask=None, enc_hidden=None, enc_kmask=None, bias=None, enc_kv=None):
        p = _p(self.dropout)
        if enc_hidden is None and enc_kv is None:
            qkv = hipops.multi_linear(hidden, [self.query.weight, self.key.weight, self.value.weight],
                                      [self.query.bias, self.key.bias, self.value.bias])
            return hipops.attention(qkv, None, kmask, bias, self.num_attention_heads, p)
        # cross-attention: the query-side mask is ignored (P/model/Bert_backbone.py:221-224)
        q = self.query(hidden)
        kv = enc_kv if enc_kv is not None else self.project_kv(enc_hidden)
        return hipops.attention(q, kv, enc_kmask, None, self.num_attention_heads, p)


class BertSelfOutput(nn.Module):
    def __init__(self, config, in_size=None):
        super().__init__()
        self.dense = Linear(in_size or config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden, input_tensor, fork=False):
        return self.LayerNorm(self.dense(hidden), residual=input_tensor, p=_p(self.dropout), fork=fork)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, hidden, kmask=None, enc_hidden=None, enc_kmask=None, bias=None, fork=False, enc_kv=None):
        h, h_res = _pair(hidden)
        return self.output(self.self(h, kmask, enc_hidden, enc_kmask, bias, enc_kv), h_res, fork)


RobertaAttention = BertAttention


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        _check_act(config)
        self.dense = Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


RobertaIntermediate, RobertaOutput = BertIntermediate, BertOutput


def _ffn_block(inter, out, x, fork=False):
    """BertIntermediate -> BertOutput (dense, dropout, LayerNorm(+residual))."""
    x, x_res = _pair(x)
    y = hipops.ffn(x, inter.dense.weight, inter.dense.bias, out.dense.weight, out.dense.bias, 'gelu', 0.0)
    return out.LayerNorm(y, residual=x_res, p=_p(out.dropout), fork=fork)


class RobertaLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert not getattr(config, 'is_decoder', False)
        self.attention = RobertaAttention(config)
        self.intermediate = RobertaIntermediate(config)
        self.output = RobertaOutput(config)

    def forward(self, hidden, kmask, fork=False):
        """hidden: a tensor or a _pair; fork=True returns a _pair (for the next layer of the same stack)."""
        return _ffn_block(self.intermediate, self.output, self.attention(hidden, kmask, fork=True), fork)


class BertCrossLayer(nn.Module):
    """self-attn (+graph bias) -> cross-attn -> FFN.  `with_lang_branch` creates the reference's unused
    lang_self_attn / lang_inter / lang_output parameters (pretrain checkpoints carry them)."""

    def __init__(self, config, with_lang_branch=True):
        super().__init__()
        self.attention = BertAttention(config)
        self.crossattention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        if with_lang_branch and getattr(config, 'use_lang2visn_attn', False):
            self.lang_self_attn = BertAttention(config)
            self.lang_inter = RobertaIntermediate(config)
            self.lang_output = RobertaOutput(config)

    def forward(self, hidden, enc_hidden, kmask, enc_kmask, bias=None, fork=False, enc_kv=None):
        a = self.attention(hidden, kmask, bias=bias, fork=True)
        a = self.crossattention(a, None, enc_hidden, enc_kmask, fork=True, enc_kv=enc_kv)
        return _ffn_block(self.intermediate, self.output, a, fork)


def init_weights(module):
    # P/model/Bert_backbone.py:840-848
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class CrossmodalEncoder(nn.Module):
    def __init__(self, config, with_lang_branch=True):
        super().__init__()
        self.num_top_layer = config.num_top_layer
        self.crossattention = nn.ModuleList([BertCrossLayer(config, with_lang_branch) for _ in range(self.num_top_layer)])
        self.crossattention.apply(init_weights)

    def kv_groups(self):
        """[(key weight, bias), (value weight, bias)] of every layer's cross-attention: the operands of hipops.linear_bank."""
        out = []
        for layer in self.crossattention:
            sa = layer.crossattention.self
            out.append([(sa.key.weight, sa.key.bias), (sa.value.weight, sa.value.bias)])
        return out

    def project_kv(self, kv_embeds):
        """the K|V projections of every layer's cross-attention for one attended sequence (see BertSelfAttention.project_kv): every layer
        attends to the SAME `kv_embeds` (P/model/Bert_backbone.py:765-781), so the projections are ONE GEMM [rows, n_layers * 2H]
        (hipops.linear_bank) and, in backward, one dgrad over the concatenated contraction instead of a dgrad per layer and an add."""
        return project_kv_bank(kv_embeds, [self])[0]

    def forward(self, q_embeds, q_kmask, kv_embeds, kv_kmask, bias=None, kv_cache=None):
        """q_kmask / kv_kmask: additive float32 [B,L] key masks (already -10000-style).  kv_cache: project_kv(kv_embeds)."""
        n = len(self.crossattention)
        # the attended sequence (and the graph-distance bias) is read by every layer: one autograd handle per layer, their gradients
        # meet in ONE launch (hipops.fanout) instead of n - 1 pairwise adds of the autograd engine
        if kv_cache is None and hipops.LINEAR_BANK:
            kv_cache = self.project_kv(kv_embeds)
        kvs = hipops.fanout(kv_embeds, n) if kv_cache is None else [kv_embeds] * n
        biases = hipops.fanout(bias, n) if torch.is_tensor(bias) else [bias] * n
        if not isinstance(q_embeds, tuple):
            q_embeds = tuple(hipops.fanout(q_embeds, 2))          # layer 0 reads it twice: first Linear + residual (a _pair)
        for i, layer in enumerate(self.crossattention):
            q_embeds = layer(q_embeds, kvs[i], q_kmask, kv_kmask, biases[i], fork=i + 1 < n, enc_kv=None if kv_cache is None else kv_cache[i])
        return q_embeds


def project_kv_bank(kv_embeds, encoders):
    """K|V projections of every cross-attention layer of SEVERAL CrossmodalEncoders that attend to one sequence (the instruction, read
    by the global-map and the local encoder: P/model/vilmodel_goat.py:399,501-504): -> one kv_cache list per encoder."""
    n = [len(e.crossattention) for e in encoders]
    if hipops.LINEAR_BANK and kv_embeds.is_cuda:
        kvs = hipops.linear_bank(kv_embeds, [g for e in encoders for g in e.kv_groups()])
    else:       # (diagnostics: one projection per layer)
        hs = hipops.fanout(kv_embeds, sum(n))
        kvs = [layer.crossattention.self.project_kv(h) for layer, h in zip([l for e in encoders for l in e.crossattention], hs)]
    out, at = [], 0
    for k in n:
        out.append(kvs[at:at + k])
        at += k
    return out


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden, location=0):
        return torch.tanh(self.dense(hidden[:, location]))          # (the row-strided [CLS] view goes to the GEMM as it is: hipops.linear)


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        _check_act(config)
        self.dense = Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden):
        return self.LayerNorm(self.dense(hidden, act='gelu'))


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))

    def forward(self, hidden):
        h = self.transform(hidden)
        return hipops.linear(h, self.decoder.weight, self.bias, None, torch.float32)

    def loss(self, hidden, targets):
        """per-token cross-entropy without materialising softmax / f32 dlogits (fused decoder + CE)."""
        return hipops.decoder_cross_entropy(self.transform(hidden), self.decoder.weight, self.bias, targets)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


# ----------------------------------------------------------------------------- panorama encoder (DETR style)
class PanoSelfAttention(nn.Module):
    """Parameter-compatible with nn.MultiheadAttention (in_proj_weight / in_proj_bias / out_proj)."""

    def __init__(self, d_model, nhead, dropout):
        super().__init__()
        if d_model // nhead != 64:
            raise NotImplementedError('GOAT HIP attention kernels are specialised for head_dim 64')
        self.embed_dim, self.num_heads, self.dropout = d_model, nhead, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, x, kmask, p):
        qkv = hipops.linear(x, self.in_proj_weight, self.in_proj_bias)
        return self.out_proj(hipops.attention(qkv, None, kmask, None, self.num_heads, p))


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout):
        super().__init__()
        self.self_attn = PanoSelfAttention(d_model, nhead, dropout)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)   # eps 1e-5 (torch default), P/model/transformer.py:143-144
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, src, kmask, normed=None, next_norm=None):
        """forward_pre (P/model/transformer.py:170-182); attention-prob dropout = the layer's dropout value.
        Each residual junction `src + dropout(sub-layer)` is computed by the LayerNorm kernel that FOLLOWS it (z_out): norm2 for the
        attention junction, `next_norm` (the next layer's norm1 or the stack's final norm) for the feed-forward junction.
        normed: norm1(src) if the previous layer already computed it.  -> (src, next_norm(src) or None)"""
        p_attn = self.self_attn.dropout if self.training else 0.0
        if normed is None:
            # (fork_in: the LayerNorm hands its input back for the skip connection, so both gradients of `src` meet inside its backward)
            normed, src = self.norm1(src, fork_in=True)
        a = self.self_attn(normed, kmask, p_attn)
        n2, src = self.norm2(a, residual=src, p=_p(self.dropout1), z_out=True)
        y = hipops.ffn(n2, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                       'gelu', _p(self.dropout))
        if next_norm is None:
            return hipops.dropout_add(y, src, _p(self.dropout2)), None
        nxt, src = next_norm(y, residual=src, p=_p(self.dropout2), z_out=True)
        return src, nxt


class TransformerEncoder(nn.Module):
    def __init__(self, config, num_layers, norm=True):
        super().__init__()
        _check_act(config)
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(config.hidden_size, config.num_attention_heads, config.intermediate_size,
                                    config.hidden_dropout_prob) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = LayerNorm(config.hidden_size, eps=1e-12) if norm else None

    def forward(self, src, valid_masks):
        """src [N,V,H] batch-first; valid_masks bool [N,V] (True = real view)."""
        kmask = inf_mask(valid_masks)
        out, normed = src, None
        for i, layer in enumerate(self.layers):
            nxt = self.layers[i + 1].norm1 if i + 1 < len(self.layers) else self.norm
            out, normed = layer(out, kmask, normed, nxt)
        return normed if self.norm is not None else out


def create_transformer_encoder(config, num_layers, norm=False):
    # P/model/ops.py:11-23
    return TransformerEncoder(config, num_layers, norm=norm)


class RegionClassification(nn.Module):
    # MRC head (P/model/pretrain_goat.py:14-25)
    def __init__(self, hidden_size, label_dim):
        super().__init__()
        self.net = nn.Sequential(Linear(hidden_size, hidden_size), nn.ReLU(),
                                 LayerNorm(hidden_size, eps=1e-12), Linear(hidden_size, label_dim))

    def forward(self, x):
        return self.net[3](self.net[2](self.net[0](x, act='relu')))


class ClsPrediction(nn.Module):
    # P/model/pretrain_goat.py:27-38
    def __init__(self, hidden_size, input_size=None):
        super().__init__()
        if input_size is None:
            input_size = hidden_size
        self.net = nn.Sequential(Linear(input_size, hidden_size), nn.ReLU(),
                                 LayerNorm(hidden_size, eps=1e-12), Linear(hidden_size, 1))

    def forward(self, x):
        h = self.net[2](self.net[0](x, act='relu'))
        return self.net[3](h)
