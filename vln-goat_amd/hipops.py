"""torch.autograd.Function wrappers over the C ABI of libgoat_hip.so (hand-written backward passes).

Every op here launches hand-written gfx950 kernels on the *current torch HIP stream* through ctypes
(`_lib.py`); nothing synchronises or allocates outside torch's caching allocator, so a whole
forward+backward step can be captured into a hipGraph (`torch.cuda.graph`).  There is no eager/CPU
fallback: CPU tensors raise.

Compute dtype: float32 (exact-f32 MFMA, the 1e-3 parity mode) or bfloat16 (bf16 storage, f32
accumulate).  Parameters stay float32 masters; bf16 / transposed "shadows" of the weights are cached on
the Parameter object and refreshed when its version counter moves (i.e. once per optimizer step).
"""
import ctypes
import json
import math
import os

import torch

from . import _lib
from ._lib import GOAT_BF16, GOAT_F32, EPI_ACCUM, EPI_GELU, EPI_MUL_DGELU, EPI_MUL_DRELU, EPI_NONE, EPI_RELU

_ACT_EPI = {None: EPI_NONE, 'none': EPI_NONE, 'gelu': EPI_GELU, 'relu': EPI_RELU}
_ACT_DEPI = {'gelu': EPI_MUL_DGELU, 'relu': EPI_MUL_DRELU}


# ----------------------------------------------------------------------------- plumbing
from ._plumbing import _dt, _epc, _need_gpu, _ptr, _stream          # noqa: E402,F401


class RngState:
    """Dropout counter state.  Eager: host counter.  Graph capture: `dev` (uint64 on device) is bumped
    by one in-graph add per replay so every replay draws fresh masks (see goat_hip.h)."""
    seed = None         # None: taken from torch.initial_seed() at first use (so torch.manual_seed / the reference trainer's
    rank_salt = 0       # set_random_seed(seed + rank) steer it); rank_salt: GoatDataParallel folds the rank in
    base = None         # what manual_seed() was given
    counter = 0
    dev = None

    @classmethod
    def next(cls, numel):
        if cls.seed is None:
            cls.seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + 0x5EED + cls.rank_salt * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF
        off = cls.counter
        cls.counter += (int(numel) + 7) & ~7      # multiples of 8: row kernels draw masks per even-aligned counter pair
        return cls.seed, off, (cls.dev.data_ptr() if cls.dev is not None else None)


def manual_seed(seed=None):
    """Seed of the dropout masks of the HIP kernels.  None: re-derive from torch.initial_seed() (call after torch.manual_seed)."""
    RngState.base = seed
    RngState.seed = None if seed is None else (int(seed) + RngState.rank_salt * 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF
    RngState.counter = 0


# ----------------------------------------------------------------------------- weight shadows
# Cached operand copies of the float32 master weights (bf16 casts, K-/row-padded and concatenated images).  A captured
# hipGraph has their ADDRESSES baked in, so a stale copy is always rebuilt INTO ITS OWN STORAGE, never replaced by a new
# tensor: graphs captured before an optimizer step read the updated weights after it (optim.FusedAdamW refreshes most copies
# inside its update kernel and the rest through `refresh_shadows`).
def _store_shadow(cache, key, ver, w):
    ent = cache.get(key)
    if ent is not None and ent[1].shape == w.shape and ent[1].dtype == w.dtype and ent[1].device == w.device:
        ent[1].copy_(w)
        cache[key] = (ver, ent[1])
        return ent[1]
    cache[key] = (ver, w)
    return w


def _build_shadow(param, dtype, transposed, pad_k):
    with torch.no_grad():
        w = param.detach()
        if pad_k:
            w = torch.nn.functional.pad(w, (0, pad_k))
        if transposed:
            w = w.t()
        return w.to(dtype).contiguous()


def _shadow(param, dtype, transposed=False, pad_k=0):
    """dtype-cast (and optionally transposed / K-padded) copy of a float32 master weight, cached."""
    key = (dtype, transposed, pad_k)
    cache = param.__dict__.setdefault('_goat_shadow', {})
    ent = cache.get(key)
    ver = param._version
    if ent is not None and ent[0] == ver and ent[1].device == param.device:
        return ent[1]
    return _store_shadow(cache, key, ver, _build_shadow(param, dtype, transposed, pad_k))


def _build_cat(params, dtype, transposed):
    with torch.no_grad():
        w = torch.cat([p.detach() for p in params], 0)
        if transposed:
            w = w.t()
        return w.to(dtype).contiguous()


def _shadow_cat(params, dtype, transposed=False):
    """Row-concatenation of several [N_i,K] weights (fused QKV / KV projection), cached on the first."""
    key = ('cat', dtype, transposed, tuple(id(p) for p in params))
    cache = params[0].__dict__.setdefault('_goat_shadow', {})
    ver = tuple(p._version for p in params)
    ent = cache.get(key)
    if ent is not None and ent[0] == ver and ent[1].device == params[0].device:
        return ent[1]
    return _store_shadow(cache, key, ver, _build_cat(params, dtype, transposed))


def _build_catb(biases):
    with torch.no_grad():
        return torch.cat([x.detach().float() for x in biases], 0).contiguous()


def _cat_bias(biases):
    key = ('catb', tuple(id(b) for b in biases))
    cache = biases[0].__dict__.setdefault('_goat_shadow', {})
    ver = tuple(b._version for b in biases)
    ent = cache.get(key)
    if ent is not None and ent[0] == ver and ent[1].device == biases[0].device:
        return ent[1]
    return _store_shadow(cache, key, ver, _build_catb(biases))


def _build_rows_padded(param, dtype, rows):
    with torch.no_grad():
        w = torch.zeros((rows, param.shape[1]), dtype=dtype, device=param.device)
        w[:param.shape[0]] = param.detach().to(dtype)
        return w


def refresh_shadows(param, by_id, done=()):
    """Rebuild, in place, every cached copy hanging off `param` whose storage address is not in `done` (the copies an update
    kernel has already refreshed).  For an optimizer that writes the masters through raw pointers (no version bump).
    by_id: {id(parameter): parameter} of every parameter that may be a member of a concatenated copy."""
    cache = param.__dict__.get('_goat_shadow')
    if not cache:
        return 0
    n = 0
    for key, (ver, t) in list(cache.items()):
        if t.data_ptr() in done:
            continue
        if key[0] == 'cat':
            members = [by_id.get(i) for i in key[3]]
            if any(m is None for m in members):
                del cache[key]          # a member is gone: nothing can read this copy any more
                continue
            t.copy_(_build_cat(members, key[1], key[2]))
        elif key[0] == 'catb':
            members = [by_id.get(i) for i in key[1]]
            if any(m is None for m in members):
                del cache[key]
                continue
            t.copy_(_build_catb(members))
        elif key[0] == 'rowpad':
            t.copy_(_build_rows_padded(param, key[1], key[2]))
        elif key[0] == 'bpad':
            t.copy_(_build_bias_padded(param, key[1]))
        else:
            t.copy_(_build_shadow(param, key[0], key[1], key[2]))
        n += 1
    return n


# ----------------------------------------------------------------------------- raw kernels
# (PROFILE — bench.py's per-launch HIP-event timing — lives in tuning.py beside the autotuner state; `hipops.PROFILE` is an alias)


def gemm_nt(a, b, out, bias=None, epi=EPI_NONE, aux=None, split_k=1):
    """out[M,N] = epi(a[M,K] @ b[N,K]^T + bias)."""
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and out.shape[0] == M and out.shape[1] == N
    if tuning.PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    st = _lib.lib().goat_gemm_nt(_stream(), _dt(a), _dt(out), _ptr(a), a.stride(0), _ptr(b), b.stride(0),
                                 _ptr(out), out.stride(0), M, N, K,
                                 _ptr(bias) if bias is not None else None, epi,
                                 _ptr(aux) if aux is not None else None,
                                 aux.stride(0) if aux is not None else 0, split_k)
    _lib.check(st, 'goat_gemm_nt(M=%d,N=%d,K=%d)' % (M, N, K))
    if tuning.PROFILE is not None:
        e1.record()
        tuning.PROFILE.append((e0, e1, 2.0 * M * N * K, (M, N, K, epi, split_k, str(a.dtype)),
                        ('goat_gemm_nt', (_dt(a), _dt(out), _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), out.stride(0),
                                          M, N, K, _ptr(bias) if bias is not None else None, epi,
                                          _ptr(aux) if aux is not None else None, aux.stride(0) if aux is not None else 0,
                                          split_k), (a, b, out, bias, aux))))
    return out


def transpose_pad(x, colsum=None):
    """x[R,C] -> out[C, ld] with ld = R rounded up to the 16-byte chunk, padding zero-filled."""
    R, C = x.shape
    e = _epc(x)
    ld = (R + e - 1) // e * e
    out = torch.empty((C, ld), dtype=x.dtype, device=x.device)
    st = _lib.lib().goat_transpose(_stream(), _dt(x), _ptr(x), x.stride(0), _ptr(out), ld, R, C,
                                   _ptr(colsum) if colsum is not None else None)
    _lib.check(st, 'goat_transpose')
    return out


def _split_k(n_out_tiles, k_tiles, target=384):
    return max(1, min(k_tiles, int(round(float(target) / max(1, n_out_tiles)))))


def colsum(x, out=None):
    """float32 column sums of x[R,C] (bias gradient)."""
    R, C = x.shape
    if out is None:
        out = torch.zeros(C, dtype=torch.float32, device=x.device)
    st = _lib.lib().goat_colsum(_stream(), _dt(x), _ptr(x), x.stride(0), R, C, _ptr(out))
    _lib.check(st, 'goat_colsum')
    return out


# ----------------------------------------------------------------------------- GEMM configuration / autotuner: tuning.py
from . import tuning                                                   # noqa: E402
from .tuning import (BALANCED, EIGHT_WAVES, PERSIST, PINGPONG, TUNED_FILE, TUNE_EVENTS, USE_PERSIST, USE_PP, _FLUSH, _TUNED,      # noqa: E402,F401
                     _heuristic_cfg, _launch_gemm_bf16, _tile_candidates, _time_cfg, _tune_gemm, load_tuned, n_cu, save_tuned, stage_name, tile, tile_name)


# (Dropout inside the FFN GEMM epilogues — round 2's goat_gemm_bf16_dropout — measured 6.25 vs 6.22 ms per step: removed in round 3.)
def gemm(a, b, out, ta=False, tb=False, bias=None, epi=EPI_NONE, aux=None, split_k=1, colsum_out=None, split_opts=None,
         accumulate=False, zero_first=False):
    """out[M,N] = epi(op(a) @ op(b)^T + bias); ta: a is [Kc,M] (else [M,Kc]); tb: b is [Kc,N] (else [N,Kc]).
    bf16 -> pipelined LDS-DMA kernel (goat_gemm_bf16) whenever its layout rules hold; otherwise (f32 parity
    path, odd contraction lengths) explicit transposes + goat_gemm_nt.
    split_opts: candidate split-K factors the autotuner may choose from (the caller must have zero-filled
    `out` if any of them is > 1); returns `out`.
    accumulate: out (float32) += product — split-K launches add atomically anyway, unsplit ones use the
    read-modify-write epilogue GOAT_EPI_ACCUM (gradient-arena sinks).
    zero_first: `out` holds stale data: clear it here if (and only if) the launch ends up split (atomics)."""
    Kc = a.shape[0] if ta else a.shape[1]
    M = a.shape[1] if ta else a.shape[0]
    N = b.shape[1] if tb else b.shape[0]
    assert (b.shape[0] if tb else b.shape[1]) == Kc and out.shape[0] == M and out.shape[1] == N
    if M == 0 or N == 0:          # empty problem (e.g. an MRC batch without masked rows): the reference returns an empty tensor
        return out
    if Kc == 0:
        return out if accumulate else out.zero_()
    fast = (a.dtype == torch.bfloat16 and not (ta and not tb) and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
            and ((ta and tb) or Kc % 64 == 0) and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and a.stride(1) == 1 and b.stride(1) == 1)
    if not fast:
        if accumulate:       # f32 parity path / odd shapes: product into a temporary, then one add
            tmp = torch.zeros_like(out) if split_k > 1 else torch.empty_like(out)
            gemm(a, b, tmp, ta, tb, bias, epi, aux, split_k, colsum_out, None, False)
            return out.add_(tmp)
        if zero_first and split_k > 1:
            out.zero_()
        if colsum_out is not None:
            colsum(a, colsum_out)
        if ta:
            a = transpose_pad(a)
            if tb:
                b = transpose_pad(b)
            elif b.shape[1] != a.shape[1]:
                b = torch.nn.functional.pad(b, (0, a.shape[1] - b.shape[1]))
        elif tb:
            b = transpose_pad(b)
            if b.shape[1] != a.shape[1]:
                a = torch.nn.functional.pad(a, (0, b.shape[1] - a.shape[1]))
        if split_k > 1:
            bk = 64 if a.dtype == torch.bfloat16 else 32
            split_k = max(2, min(split_k, (a.shape[1] + bk - 1) // bk))
        return gemm_nt(a, b, out, bias, epi, aux, split_k)
    key = (ta, tb, M, N, Kc, epi, out.dtype == torch.float32, split_k, bias is not None)
    cfg = _TUNED.get(key)
    if cfg is None:
        if tuning.AUTOTUNE and not torch.cuda.is_current_stream_capturing() and tuning.PROFILE is None:
            cfg = _tune_gemm(key, a, b, out, ta, tb, M, N, Kc, bias, epi, aux, split_opts or (split_k,), colsum_out)
        else:
            cfg = _heuristic_cfg(ta, tb, M, N, Kc, split_k) + (split_k,)
            tuning.STATS['gemm_heuristic'] += 1          # a launch on a configuration nobody measured (tests assert a captured step has none)
            if len(tuning.STATS_LOG) < 256:
                tuning.STATS_LOG.append(('gemm', key, bool(torch.cuda.is_current_stream_capturing())))
            if os.environ.get('GOAT_GEMM_CFG_LOG'):      # (diagnostics: shapes that run on the heuristic, e.g. first seen inside a capture)
                import sys
                print('[gemm cfg] heuristic %s capturing=%s autotune=%s' % (key, torch.cuda.is_current_stream_capturing(), tuning.AUTOTUNE), file=sys.stderr)
    else:
        tuning.STATS['gemm_tuned'] += 1
    bm, nstage, split_cfg = cfg
    # a split launch accumulates with atomics: only allowed when the caller zero-filled `out` (split requested / split_opts
    # given), asked for the clear (zero_first) or accumulates anyway.  A tuned entry with split > 1 must never be applied to a
    # buffer the caller left uninitialised (found by the full-size gradient pins of round 2).
    may_split = split_k > 1 or split_opts is not None or zero_first or accumulate
    split_k = split_cfg if may_split else 1
    if zero_first and split_k > 1:
        out.zero_()
    if accumulate and split_k == 1:
        assert epi == EPI_NONE and out.dtype == torch.float32 and bias is None
        epi = EPI_ACCUM
    if tuning.PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _launch_gemm_bf16(a, b, out, ta, tb, M, N, Kc, bias, epi, aux, split_k, bm, nstage, colsum_out)
    if tuning.PROFILE is not None:
        e1.record()
        cargs = (int(ta), int(tb), _dt(out), _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out),
                 out.stride(0), M, N, Kc, _ptr(bias) if bias is not None else None, epi,
                 _ptr(aux) if aux is not None else None, aux.stride(0) if aux is not None else 0,
                 split_k, bm, nstage, _ptr(colsum_out) if colsum_out is not None else None)
        tuning.PROFILE.append((e0, e1, 2.0 * M * N * Kc, (M, N, Kc, epi, split_k, 'v2 t%d%d %s %s' % (ta, tb, tile_name(bm), stage_name(nstage))),
                        ('goat_gemm_bf16', cargs, (a, b, out, bias, aux, colsum_out))))
    return out


# ----------------------------------------------------------------------------- hipGraph capture / parallel branches: streams.py;
# deferred weight gradients and LayerNorm column reductions: wgrad_queue.py
from .streams import Branch, graph, note_parallel_branch, _RETAINED_GRAPHS          # noqa: E402,F401
from .wgrad_queue import LnReduceQueue, WgradQueue                                   # noqa: E402,F401


def _sink(param, keep_queued=False):
    """Gradient-arena slice bound to `param` (dp.GradArena.attach), or None.  When it is bound — i.e. still the
    object behind param.grad — backward passes accumulate the parameter's gradient straight into it and return
    None to autograd (no temporary, no zero-fill, no `grad += dW` kernel).  Setting param.grad = None (or to any
    other tensor) silently restores the ordinary autograd path.
    keep_queued: the caller is about to queue ANOTHER weight-gradient problem for this slice on this stream and the two may be merged
    (WgradQueue.mergeable): a queued write of the same stream then stays queued."""
    if param is None:
        return None
    if WgradQueue.pending_ids and id(param) in WgradQueue.pending_ids:
        if not (keep_queued and WgradQueue.pending_ids[id(param)] == torch.cuda.current_stream().cuda_stream):
            WgradQueue.flush_param(id(param))      # a queued write of this slice must land before anything else touches it
    s = param.__dict__.get('_goat_sink')
    return s if (s is not None and param.grad is s) else None


def _sink_cat(params, keep_queued=False):
    """One [sum(rows), ...] view over the arena slices of several parameters if they are adjacent in the arena
    (query/key/value weights of a block), else None."""
    sinks = [_sink(p, keep_queued) for p in params]
    if any(t is None for t in sinks):
        return None
    for a, b in zip(sinks, sinks[1:]):
        if a.data_ptr() + a.numel() * a.element_size() != b.data_ptr() or a.shape[1:] != b.shape[1:]:
            return None
    s0 = sinks[0]
    return torch.as_strided(s0, (sum(t.shape[0] for t in sinks),) + tuple(s0.shape[1:]), s0.stride())


ARENA_EPOCH = [0]       # bumped by dp.GradArena.zero(): a sink's first use in a step overwrites / clears its slice


def _first_touch(*params):
    """True if none of `params` has been written through its sink yet in this step (marks them written).
    Small parameters (`_goat_prezero`: biases, LayerNorm, ...) are cleared by GradArena.zero() at the start of the step:
    they never count as a first touch — writers just accumulate.  Mixed states among the others (some written, some
    not) cannot be served by one kernel launch: the unwritten slices are cleared here and the call is an accumulation."""
    cur = ARENA_EPOCH[0]
    params = [p for p in params if not p.__dict__.get('_goat_prezero')]
    if not params:
        return False
    seen = [p.__dict__.get('_goat_epoch') == cur for p in params]
    for p, was in zip(params, seen):
        p.__dict__['_goat_epoch'] = cur
        if not was and any(seen):
            p.__dict__['_goat_sink'].zero_()
    return not any(seen)


def _prep_fallback(*params):
    """A Function is about to return ordinary gradients for `params` (autograd will add them into .grad): if a
    .grad is an arena slice nobody has written yet in this step it still holds the previous step's values."""
    for p in params:
        t = _sink(p)
        if t is not None and _first_touch(p):
            t.zero_()


def _wgrad_impl(dy, x, want_bias, w_sink=None, b_sink=None, first=False, b_first=False, defer_ids=None):
    M, N = dy.shape
    K = x.shape[1]
    # default (bm 64, split) from scripts/wgrad_sweep.py; the autotuner may pick another split (output is zero-filled)
    tiles = ((N + 63) // 64) * ((K + 127) // 128)
    kt = (M + 63) // 64
    if tiles < 128:
        split = max(1, min(int(round(250.0 / tiles)), kt // 8))
    else:
        split = max(1, min(int(round(500.0 / tiles)), kt // 24))
    nb = N if want_bias else 0
    tunable = tuning.AUTOTUNE and dy.dtype == torch.bfloat16
    if w_sink is not None:
        # gradient-arena slice.  First use in this step: clear it right here (the fill leaves the lines in the
        # Infinity Cache for the split-K atomics) or, unsplit, simply overwrite it; later uses accumulate.
        db = b_sink
        if want_bias and db is None:
            db = torch.zeros(N, dtype=torch.float32, device=dy.device)
        elif want_bias and b_first:
            db.zero_()
        if (defer_ids is not None and WgradQueue.enabled and dy.dtype == torch.bfloat16 and (b_sink is not None or not want_bias)
                and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and w_sink.is_contiguous()
                and not torch.is_grad_enabled()):
            WgradQueue.push(dy, x, w_sink, db if want_bias else None, defer_ids, accumulate=not first)
            return None, None
        for i in (defer_ids or ()):
            WgradQueue.flush_param(i)           # (a write left queued for merging must land before this direct one)
        gemm(dy, x, w_sink, ta=True, tb=True, split_k=split, colsum_out=db if want_bias else None,
             split_opts=(1, 2, 3, 4, 6, 8) if tunable else None, accumulate=not first, zero_first=first)
        return None, (db if (want_bias and b_sink is None) else None)
    if split > 1 or tunable:
        buf = torch.zeros(N * K + nb, dtype=torch.float32, device=dy.device)
        dw = buf[:N * K].view(N, K)
        db = buf[N * K:] if want_bias else None
        split = max(split, 2) if not tunable else split
    else:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
        db = torch.zeros(N, dtype=torch.float32, device=dy.device) if want_bias else None
    gemm(dy, x, dw, ta=True, tb=True, split_k=split, colsum_out=db,
         split_opts=(1, 2, 3, 4, 6, 8) if tunable else None)
    return dw, db


def wgrad(dy, x, want_bias, w_sink=None, b_sink=None, first=False, b_first=False, defer_ids=None):
    """dW[N,K] (f32) = dy[M,N]^T @ x[M,K] ; db[N] (f32) = colsum(dy), fused into the same kernel.
    One zero-fill covers both outputs (split-K partial tiles and the bias sums are accumulated atomically).
    With sinks (gradient-arena slices) the results are accumulated in place — deferred into a grouped launch when
    possible (WgradQueue) — and (None, None) is returned.  (Measured and dropped: running each weight-gradient GEMM on a
    second stream next to the dgrad chain was slower, 9.9 vs 9.3 ms per step.)"""
    return _wgrad_impl(dy, x, want_bias, w_sink, b_sink, first, b_first, defer_ids)


# ----------------------------------------------------------------------------- Linear
def _pad_k(x, e):
    K = x.shape[1]
    pad = (-K) % e
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    return x, pad


SMALLK_WGRAD = os.environ.get('GOAT_NO_SMALLK', '0') != '1'       # (diagnostics: A/B of the short-input weight-gradient kernel)
PANO_SINK = os.environ.get('GOAT_NO_PANO_SINK', '0') != '1'


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b)   (F.linear + activation; P/model/Bert_backbone.py:302,348-357,362)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, out_dtype):
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous() and not (x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0
                                           and x2.dtype == torch.bfloat16 and x2.shape[1] % 64 == 0
                                           and x2.shape[0] * x2.stride(0) * 2 < (1 << 31)):      # (the GEMM's operand window spans rows * lda: < 2 GiB)
            x2 = x2.contiguous()          # (a row-strided view — the [CLS] rows hidden[:, 0] of a pooler — goes to the GEMM as it is: lda = its row stride)
        Kw = weight.shape[1]
        if x2.shape[1] == Kw:
            x2, pad = _pad_k(x2, _epc(x2))
            prepad = False
        else:
            # input already K-padded with zero columns to the GEMM's 16-byte chunk (train_step.prepare_position_features: the 7- / 14-wide
            # position features are cast and padded ONCE per batch on the host instead of by two launches per Linear and step)
            pad = (-Kw) % _epc(x2)
            if x2.shape[1] != Kw + pad:
                raise ValueError('linear: input width %d matches neither the weight (%d) nor its padded width (%d)' % (x2.shape[1], Kw, Kw + pad))
            prepad = True
        w = _shadow(weight, x2.dtype, False, pad)
        N = w.shape[0]
        out = torch.empty((x2.shape[0], N), dtype=out_dtype or x2.dtype, device=x2.device)
        aux = None
        epi = _ACT_EPI[act]
        if epi != EPI_NONE and (ctx.needs_input_grad[0] or weight.requires_grad):
            aux = torch.empty_like(out)
        gemm(x2, w, out, bias=bias.detach() if bias is not None else None, epi=epi, aux=aux)
        ctx.save_for_backward(x2, aux)
        ctx.weight, ctx.bias, ctx.has_bias, ctx.act, ctx.pad, ctx.xshape, ctx.prepad = weight, bias, bias is not None, act, pad, x.shape, prepad
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        weight = ctx.weight
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.act not in (None, 'none'):
            dy2 = act_bwd(dy2, aux, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            w = _shadow(weight, x2.dtype, False, ctx.pad)  # [N, Kp]
            dx = torch.empty_like(x2)
            gemm(dy2, w, dx, ta=False, tb=True)        # dx[M,Kp] = dy[M,N] @ W[N,Kp]
            if ctx.pad and not ctx.prepad:
                dx = dx[:, :x2.shape[1] - ctx.pad]
            dx = dx.reshape(ctx.xshape)
        if SMALLK_WGRAD and ctx.pad and x2.shape[1] - ctx.pad <= 16 and (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])):
            # short-input Linear (7- / 14-wide position features, K padded to a 16-byte chunk for the GEMM): one kernel adds the
            # weight and bias gradients straight into the arena slices (otherwise: padded GEMM into a temporary, slice copy, grad += dW)
            K = x2.shape[1] - ctx.pad
            w_sink = _sink(weight)
            b_sink = _sink(ctx.bias) if (ctx.has_bias and w_sink is not None) else None
            if w_sink is not None and (b_sink is not None or not ctx.has_bias):
                if _first_touch(weight):
                    w_sink.zero_()
                if b_sink is not None and _first_touch(ctx.bias):
                    b_sink.zero_()
                dwt, dbt = w_sink, b_sink
            else:
                _prep_fallback(weight, *([ctx.bias] if ctx.has_bias else []))
                dwt = torch.zeros(weight.shape, dtype=torch.float32, device=dy2.device)
                dbt = torch.zeros(weight.shape[0], dtype=torch.float32, device=dy2.device) if ctx.has_bias else None
            st = _lib.lib().goat_wgrad_smallk(_stream(), _dt(dy2), _ptr(dy2), dy2.stride(0), _ptr(x2), x2.stride(0), dy2.shape[0],
                                              dy2.shape[1], K, _ptr(dwt), dwt.stride(0), _ptr(dbt) if dbt is not None else None)
            _lib.check(st, 'goat_wgrad_smallk')
            if dwt is not w_sink:
                dw, db = dwt, dbt
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            keep = WgradQueue.mergeable(dy2.shape[0], dy2.shape[1], x2.shape[1])      # small (BPTT-step) problems of one weight are merged into one
            w_sink = None if ctx.pad else _sink(weight, keep)
            b_sink = _sink(ctx.bias, keep) if w_sink is not None else None
            first = w_sink is not None and _first_touch(weight)
            b_first = b_sink is not None and _first_touch(ctx.bias)
            _prep_fallback(*(([] if w_sink is not None else [weight]) + ([] if b_sink is not None else [ctx.bias])))
            ids = [id(weight)] + ([id(ctx.bias)] if b_sink is not None else [])
            dw, db = wgrad(dy2, x2, ctx.has_bias, w_sink, b_sink, first, b_first, ids if w_sink is not None else None)
            if ctx.pad:
                dw = dw[:, :x2.shape[1] - ctx.pad].contiguous()
        return dx, dw, db, None, None


class _RowDotFn(torch.autograd.Function):
    """y[..., 0] = x[..., :] . w + b for a Linear(H, 1) (the last layer of ClsPrediction, P/model/pretrain_goat.py:27-38) without the GEMM
    machinery: goat_rowdot_fwd / _bwd (one wave per row; dW / db by block partials + atomics straight into the arena slices)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, H = x2.shape
        y = torch.empty(M, dtype=x2.dtype, device=x2.device)
        st = _lib.lib().goat_rowdot_fwd(_stream(), _dt(x2), _ptr(x2), _ptr(weight), _ptr(bias) if bias is not None else None, _ptr(y), M, H)
        _lib.check(st, 'goat_rowdot_fwd')
        ctx.save_for_backward(x2)
        ctx.weight, ctx.bias, ctx.xshape = weight, bias, x.shape
        return y.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dy):
        x2, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        M, H = x2.shape
        dy2 = dy.reshape(-1)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
        dw = db = dwt = dbt = None
        if need_w:
            w_sink = _sink(weight)
            b_sink = _sink(bias) if (bias is not None and w_sink is not None) else None
            if w_sink is not None and (b_sink is not None or bias is None):
                if _first_touch(weight):
                    w_sink.zero_()
                if b_sink is not None and _first_touch(bias):
                    b_sink.zero_()
                dwt, dbt = w_sink, b_sink
            else:
                _prep_fallback(weight, *([bias] if bias is not None else []))
                buf = torch.zeros(H + 1, dtype=torch.float32, device=x2.device)
                dwt, dbt = buf[:H], (buf[H:] if bias is not None else None)
                dw, db = dwt.view(weight.shape), (dbt.view(bias.shape) if bias is not None else None)
        st = _lib.lib().goat_rowdot_bwd(_stream(), _dt(x2), _ptr(x2), _ptr(weight), _ptr(dy2), _ptr(dx) if dx is not None else None,
                                        _ptr(dwt) if dwt is not None else None, _ptr(dbt) if dbt is not None else None, M, H)
        _lib.check(st, 'goat_rowdot_bwd')
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db


ROWDOT = os.environ.get('GOAT_NO_ROWDOT', '0') != '1'        # (diagnostics: Linear(H, 1) through the GEMM path again)


def linear(x, weight, bias=None, act=None, out_dtype=None):
    if (ROWDOT and weight.shape[0] == 1 and act in (None, 'none') and out_dtype is None and x.is_cuda and x.shape[-1] == weight.shape[1]
            and weight.shape[1] % 8 == 0 and weight.shape[1] <= 1024 and weight.dtype == torch.float32 and weight.is_contiguous()
            and (bias is None or bias.dtype == torch.float32) and x.dtype in (torch.float32, torch.bfloat16)):
        return _RowDotFn.apply(x, weight, bias)
    return _LinearFn.apply(x, weight, bias, act, out_dtype)


def act_bwd(dy, u, act, p=0.0, rng=(0, 0, None)):
    """dx = dropmask_p(dy) * act'(u)  (goat_act_bwd)."""
    dy = dy if dy.is_contiguous() else dy.contiguous()
    dx = torch.empty_like(dy)
    st = _lib.lib().goat_act_bwd(_stream(), _dt(dy), _ptr(dy), _ptr(u), _ptr(dx), dy.numel(), _ACT_EPI[act],
                                 p, rng[0], rng[1], rng[2])
    _lib.check(st, 'goat_act_bwd')
    return dx


class _FfnFn(torch.autograd.Function):
    """y = dropout_p(act(x @ W1^T + b1)) @ W2^T + b2.  With p == 0 (BERT blocks: BertIntermediate +
    BertOutput.dense, P/model/Bert_backbone.py:345-368) the activation derivative is fused into the dgrad
    GEMM epilogue; with p > 0 (panorama encoder FFN, P/model/transformer.py:179) one elementwise kernel
    applies mask and derivative."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, p):
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        W1 = _shadow(w1, x2.dtype)
        W2 = _shadow(w2, x2.dtype)
        M, F_ = x2.shape[0], W1.shape[0]
        u = torch.empty((M, F_), dtype=x2.dtype, device=x2.device)
        h = torch.empty_like(u)
        rng = RngState.next(h.numel()) if p > 0 else (0, 0, None)
        gemm(x2, W1, h, bias=b1.detach(), epi=_ACT_EPI[act], aux=u)
        if p > 0:
            st = _lib.lib().goat_dropout_add_fwd(_stream(), _dt(h), _ptr(h), None, _ptr(h), h.numel(), p,
                                                 rng[0], rng[1], rng[2])
            _lib.check(st, 'goat_dropout_add_fwd')
        y = torch.empty((M, W2.shape[0]), dtype=x2.dtype, device=x2.device)
        gemm(h, W2, y, bias=b2.detach())
        ctx.save_for_backward(x2, u, h)
        ctx.w1, ctx.w2, ctx.act, ctx.xshape, ctx.p, ctx.rng = w1, w2, act, x.shape, p, rng
        ctx.b1, ctx.b2 = b1, b2
        return y.view(*x.shape[:-1], W2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, u, h = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        W2 = _shadow(ctx.w2, x2.dtype)  # [H, F]
        du = torch.empty_like(u)
        if ctx.p > 0:
            gemm(dy2, W2, du, tb=True)
            du = act_bwd(du, u, ctx.act, ctx.p, ctx.rng)
        else:
            gemm(dy2, W2, du, tb=True, epi=_ACT_DEPI[ctx.act], aux=u)
        k2 = WgradQueue.mergeable(dy2.shape[0], dy2.shape[1], h.shape[1])
        s2 = _sink(ctx.w2, k2)
        sb2 = _sink(ctx.b2, k2) if s2 is not None else None
        f2 = s2 is not None and _first_touch(ctx.w2)
        bf2 = sb2 is not None and _first_touch(ctx.b2)
        _prep_fallback(*(([] if s2 is not None else [ctx.w2]) + ([] if sb2 is not None else [ctx.b2])))
        dw2, db2 = wgrad(dy2, h, True, s2, sb2, f2, bf2, [id(ctx.w2), id(ctx.b2)] if s2 is not None else None)
        W1 = _shadow(ctx.w1, x2.dtype)  # [F, H]
        dx = torch.empty_like(x2)
        gemm(du, W1, dx, tb=True)
        k1 = WgradQueue.mergeable(du.shape[0], du.shape[1], x2.shape[1])
        s1 = _sink(ctx.w1, k1)
        sb1 = _sink(ctx.b1, k1) if s1 is not None else None
        f1 = s1 is not None and _first_touch(ctx.w1)
        bf1 = sb1 is not None and _first_touch(ctx.b1)
        _prep_fallback(*(([] if s1 is not None else [ctx.w1]) + ([] if sb1 is not None else [ctx.b1])))
        dw1, db1 = wgrad(du, x2, True, s1, sb1, f1, bf1, [id(ctx.w1), id(ctx.b1)] if s1 is not None else None)
        return dx.view(ctx.xshape), dw1, db1, dw2, db2, None, None


def ffn(x, w1, b1, w2, b2, act='gelu', p=0.0):
    return _FfnFn.apply(x, w1, b1, w2, b2, act, float(p))


# ----------------------------------------------------------------------------- tied decoder + cross-entropy (MLM)
def _build_bias_padded(bias, n):
    with torch.no_grad():
        b = torch.zeros(n, dtype=torch.float32, device=bias.device)
        b[:bias.numel()] = bias.detach().float()
        return b


def _bias_padded(bias, n):
    """float32 [n] copy of a bias with zeros behind it (the 64-padded vocabulary of the MLM decoder), cached like _shadow: built
    once, refreshed in place — not a fill + a copy in every step."""
    key = ('bpad', n)
    cache = bias.__dict__.setdefault('_goat_shadow', {})
    ent = cache.get(key)
    if ent is not None and ent[0] == bias._version and ent[1].device == bias.device:
        return ent[1]
    return _store_shadow(cache, key, bias._version, _build_bias_padded(bias, n))


def _shadow_rows_padded(param, dtype, rows):
    """[rows, K] copy of a [N, K] weight (N <= rows, extra rows zero), cached like _shadow."""
    key = ('rowpad', dtype, rows)
    cache = param.__dict__.setdefault('_goat_shadow', {})
    ent = cache.get(key)
    if ent is not None and ent[0] == param._version and ent[1].device == param.device:
        return ent[1]
    return _store_shadow(cache, key, param._version, _build_rows_padded(param, dtype, rows))


class _DecoderCeFn(torch.autograd.Function):
    """loss[m] = CE(h[m] @ W^T + b, target[m])  — BertLMPredictionHead.decoder + F.cross_entropy
    (P/model/Bert_backbone.py:826-829, P/model/pretrain_goat.py:210-216).  The vocabulary dimension is padded
    to a multiple of 64 (zero weight rows) so the logits GEMM, its dgrad (contraction over the vocabulary,
    split-K) and its wgrad all run on the LDS-DMA kernel; cross-entropy forward/backward are two row kernels
    that never materialise a softmax."""

    @staticmethod
    def forward(ctx, h, weight, bias, targets):
        _need_gpu(h)
        h2 = h.reshape(-1, h.shape[-1])
        h2 = h2 if h2.is_contiguous() else h2.contiguous()
        M, K = h2.shape
        N = weight.shape[0]
        Np = (N + 63) // 64 * 64
        W = _shadow_rows_padded(weight, h2.dtype, Np)
        Nl = W.shape[0]
        bpad = _bias_padded(bias, Nl)
        logits = torch.empty((M, Nl), dtype=torch.float32, device=h.device)
        gemm(h2, W, logits, bias=bpad)
        loss = torch.empty(M, dtype=torch.float32, device=h.device)
        lse = torch.empty(M, dtype=torch.float32, device=h.device)
        tg = targets.contiguous()
        st = _lib.lib().goat_ce_fwd(_stream(), _ptr(logits), Nl, M, N, _ptr(tg), _ptr(loss), _ptr(lse))
        _lib.check(st, 'goat_ce_fwd')
        ctx.save_for_backward(h2, logits, lse, tg)
        ctx.weight, ctx.bias, ctx.N, ctx.hshape = weight, bias, N, h.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        h2, logits, lse, tg = ctx.saved_tensors
        weight, N = ctx.weight, ctx.N
        M, K = h2.shape
        Nl = logits.shape[1]
        dl = torch.empty((M, Nl), dtype=h2.dtype, device=h2.device)
        dloss = dloss.contiguous().float()
        st = _lib.lib().goat_ce_bwd(_stream(), _dt(dl), _ptr(logits), Nl, M, N, _ptr(tg), _ptr(lse), _ptr(dloss), _ptr(dl), Nl)
        _lib.check(st, 'goat_ce_bwd')
        if h2.dtype == torch.bfloat16:
            W = _shadow_rows_padded(weight, h2.dtype, Nl)
            dh32 = torch.zeros((M, K), dtype=torch.float32, device=h2.device)
            gemm(dl, W, dh32, tb=True, split_k=8, split_opts=(4, 6, 8, 12, 16))     # contraction over the vocabulary
            dh = dh32.to(h2.dtype)
        else:
            dh = torch.empty_like(h2)
            gemm(dl, _shadow_rows_padded(weight, h2.dtype, Nl), dh, tb=True)
        w_sink = _sink(weight)
        if w_sink is not None:      # the tied word-embedding table: accumulate next to the embedding scatter-add
            b_sink = _sink(ctx.bias)
            first = _first_touch(weight)
            b_first = b_sink is not None and _first_touch(ctx.bias)
            if b_sink is None:
                _prep_fallback(ctx.bias)
            dw, db = wgrad(dl[:, :N], h2, True, w_sink, b_sink, first, b_first, [id(weight), id(ctx.bias)])
            return dh.view(ctx.hshape), None, db, None
        _prep_fallback(weight, ctx.bias)
        dw, db = wgrad(dl, h2, True)
        return dh.view(ctx.hshape), dw[:N], db[:N], None


def decoder_cross_entropy(h, weight, bias, targets):
    return _DecoderCeFn.apply(h, weight, bias, targets)


# ----------------------------------------------------------------------------- fused multi-weight projection
class _MultiLinearFn(torch.autograd.Function):
    """y = x @ cat(W_i)^T + cat(b_i): one GEMM for the separate query/key/value Linears of a BERT block
    (P/model/Bert_backbone.py:210,231-232).  Gradients are returned per original parameter."""

    @staticmethod
    def forward(ctx, x, *wb):
        n = len(wb) // 2
        ws, bs = wb[:n], wb[n:]
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        W = _shadow_cat(ws, x2.dtype)
        b = _cat_bias(bs)
        out = torch.empty((x2.shape[0], W.shape[0]), dtype=x2.dtype, device=x2.device)
        gemm(x2, W, out, bias=b)
        ctx.save_for_backward(x2)
        ctx.ws, ctx.bs, ctx.xshape = ws, bs, x.shape
        return out.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        ws = ctx.ws
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            W = _shadow_cat(ws, x2.dtype)
            dx = torch.empty_like(x2)
            gemm(dy2, W, dx, tb=True)
            dx = dx.view(ctx.xshape)
        keep = WgradQueue.mergeable(dy2.shape[0], dy2.shape[1], x2.shape[1])
        w_sink = _sink_cat(ws, keep)
        b_sink = _sink_cat(ctx.bs, keep) if w_sink is not None else None
        first = w_sink is not None and _first_touch(*ws)
        b_first = b_sink is not None and _first_touch(*ctx.bs)
        _prep_fallback(*(([] if w_sink is not None else list(ws)) + ([] if b_sink is not None else list(ctx.bs))))
        ids = [id(t) for t in ws] + [id(t) for t in ctx.bs]
        dw, db = wgrad(dy2, x2, True, w_sink, b_sink, first, b_first, ids if w_sink is not None else None)
        sizes = [w.shape[0] for w in ws]
        none = (None,) * len(ws)
        return (dx,) + (tuple(torch.split(dw, sizes, 0)) if dw is not None else none) \
            + (tuple(torch.split(db, sizes, 0)) if db is not None else none)


def multi_linear(x, weights, biases):
    return _MultiLinearFn.apply(x, *weights, *biases)


class _GradSlots:
    """One [rows, n * width] gradient buffer shared by the n output views of a projection bank: the consumer of view i (an attention
    backward) writes its gradient straight into columns [i * width, (i + 1) * width) and hands that view back to autograd."""

    def __init__(self, out, n, lead):
        self.buf = torch.empty_like(out)
        self.n, self.width, self.lead = n, out.shape[1] // n, lead

    def view(self, i):
        return self.buf.view(*self.lead, self.buf.shape[1])[..., i * self.width:(i + 1) * self.width]


class _LinearBankFn(torch.autograd.Function):
    """y = x @ cat(W_j)^T + cat(b_j) for the projections of SEVERAL modules that read one activation — the key | value projections of
    every cross-attention layer of the cross-modal encoders on the sequence they all attend to (P/model/Bert_backbone.py:765-781: the same
    `kv_embeds` goes to every layer; P/model/vilmodel_goat.py:501-504,631-647) — as ONE GEMM [rows, n * width] instead of one per layer,
    and in backward ONE dgrad over the concatenated contraction (K = n * width) instead of n dgrads and an n-way add.
    Returns n views [..., width] of the one result (row stride n * width); hipops.attention reads them through strides and writes the
    gradient of view i into its columns of one shared buffer (_GradSlots) — no copies on either side."""

    @staticmethod
    def forward(ctx, x, n, *wb):
        m = len(wb) // 2
        ws, bs = wb[:m], wb[m:]
        _need_gpu(x)
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        W = _shadow_cat(ws, x2.dtype)
        b = _cat_bias(bs)
        out = torch.empty((x2.shape[0], W.shape[0]), dtype=x2.dtype, device=x2.device)
        gemm(x2, W, out, bias=b)
        ctx.save_for_backward(x2)
        ctx.ws, ctx.bs, ctx.xshape, ctx.n = ws, bs, x.shape, n
        need = any(ctx.needs_input_grad)          # (grad mode is off inside forward: the flags say whether a backward pass can come)
        ctx.slots = _GradSlots(out, n, x.shape[:-1]) if need else None
        width = W.shape[0] // n
        full = out.view(*x.shape[:-1], W.shape[0])
        return tuple(full[..., i * width:(i + 1) * width] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        (x2,) = ctx.saved_tensors
        ws, n, slots = ctx.ws, ctx.n, ctx.slots
        width = slots.width
        for i, g in enumerate(grads):          # a gradient that is not already in its slot (another producer than attention, a fan-out sum)
            dst = slots.view(i)
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
        dy2 = slots.buf
        dx = None
        if ctx.needs_input_grad[0]:
            W = _shadow_cat(ws, x2.dtype)
            dx = torch.empty_like(x2)
            gemm(dy2, W, dx, tb=True)
            dx = dx.view(ctx.xshape)
        # weight gradients: one problem per run of parameters that are adjacent in the gradient arena (the key | value pair of a layer),
        # reading its columns of the shared gradient buffer through the leading dimension
        per = len(ws) // n
        dws, dbs = [], []
        for i in range(n):
            wi, bi = ws[i * per:(i + 1) * per], ctx.bs[i * per:(i + 1) * per]
            dyi = dy2[:, i * width:(i + 1) * width]
            keep = WgradQueue.mergeable(dyi.shape[0], dyi.shape[1], x2.shape[1])
            w_sink = _sink_cat(wi, keep)
            b_sink = _sink_cat(bi, keep) if w_sink is not None else None
            first = w_sink is not None and _first_touch(*wi)
            b_first = b_sink is not None and _first_touch(*bi)
            _prep_fallback(*(([] if w_sink is not None else list(wi)) + ([] if b_sink is not None else list(bi))))
            ids = [id(t) for t in wi] + [id(t) for t in bi]
            dw, db = wgrad(dyi, x2, True, w_sink, b_sink, first, b_first, ids if w_sink is not None else None)
            sizes = [w.shape[0] for w in wi]
            dws += list(torch.split(dw, sizes, 0)) if dw is not None else [None] * per
            dbs += list(torch.split(db, sizes, 0)) if db is not None else [None] * per
        return (dx, None) + tuple(dws) + tuple(dbs)


LINEAR_BANK = os.environ.get('GOAT_NO_KV_BANK', '0') != '1'       # (diagnostics: A/B against one projection per layer)


def linear_bank(x, groups):
    """groups: n lists of (weight, bias) pairs — the projections of n modules that read x (e.g. [(key, value)] per cross-attention layer).
    -> n views [..., sum of the group's rows] of ONE GEMM result, each carrying `_goat_grad_slot` for its consumer's backward."""
    n = len(groups)
    ws = [w for g in groups for (w, _) in g]
    bs = [b for g in groups for (_, b) in g]
    outs = _LinearBankFn.apply(x, n, *ws, *bs)
    fn = outs[0].grad_fn
    slots = getattr(fn, 'slots', None) if fn is not None else None
    if slots is not None:
        for i, o in enumerate(outs):
            o._goat_grad_slot = (slots, i)
    return list(outs)


# ----------------------------------------------------------------------------- LayerNorm / dropout
LN_DETERMINISTIC = os.environ.get('GOAT_LN_DETERMINISTIC', '0') == '1'
LN_ATOMIC_MAX_ROWS = int(os.environ.get('GOAT_LN_ATOMIC_MAX_ROWS', '4096'))


class _LnFn(torch.autograd.Function):
    """y = LayerNorm(residual + dropout_p(x)) (P/model/Bert_backbone.py:306-310).
    fork=True returns y twice (two autograd outputs over one buffer): one for the next sub-layer's first Linear, one for
    the next LayerNorm's residual input.  Their gradients then reach backward() separately and the kernel sums them on
    load (goat_ln_bwd's dy2), which replaces the elementwise add autograd would otherwise launch for the shared tensor."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, p, fork=False, fork_in=False, z_out=False, p_out=0.0, post_add=None):
        _need_gpu(x)
        ctx.set_materialize_grads(False)
        if fork_in and (fork or residual is not None):
            raise ValueError('fork_in is for the plain LayerNorm(x) of a pre-LN block')
        if z_out and (fork or fork_in or residual is None):
            raise ValueError('z_out returns the pre-norm sum residual + dropout(x): it needs a residual')
        ctx.fork_in, ctx.z_out = fork_in, z_out
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, H)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        need_z = (r2 is not None) or p > 0
        z = torch.empty_like(x2) if need_z else None
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        seed, off, dev = RngState.next(x2.numel()) if p > 0 else (0, 0, None)
        off_out = 0
        if p_out > 0:                        # dropout on the output: a second counter range of the same stream
            s2, off_out, d2 = RngState.next(x2.numel())
            if p > 0:
                assert s2 == seed
            seed, dev = s2, d2
        pa = None
        if post_add is not None:             # y = dropout_out(LayerNorm(z) + post_add)
            pa = post_add.reshape(-1, H).to(x2.dtype)
            pa = pa if pa.is_contiguous() else pa.contiguous()
        ctx.has_post = pa is not None
        st = _lib.lib().goat_ln_fwd_do(_stream(), _dt(x2), _ptr(x2), _ptr(r2) if r2 is not None else None,
                                       _ptr(gamma), _ptr(beta), eps, p, seed, off, dev,
                                       _ptr(y), _ptr(z) if z is not None else None, _ptr(mean), _ptr(rstd), M, H, p_out, off_out,
                                       _ptr(pa) if pa is not None else None)
        _lib.check(st, 'goat_ln_fwd')
        ctx.save_for_backward(z if z is not None else x2, gamma, mean, rstd)
        ctx.rng = (p, seed, off, dev)
        ctx.out_drop = (p_out, off_out)
        ctx.has_res = residual is not None
        ctx.gb = (gamma, beta)
        ctx.shape = x.shape
        yv = y.view(x.shape)
        if z_out:       # second output: the pre-norm sum z = residual + dropout(x), the hidden state of a pre-LN stack; the gradient it
            return yv, z.view(x.shape)      # collects behind this LayerNorm comes back to THIS node (GOAT_LN_ADD_BEFORE)
        if fork_in:     # second output: x itself for the skip connection; its gradient comes back to THIS node (dx_add of the kernel)
            return yv, x.view_as(x)
        return (yv, yv.view_as(yv)) if fork else yv

    @staticmethod
    def backward(ctx, dy, dyb=None):
        dskip = None
        if ctx.fork_in or ctx.z_out:
            dskip, dyb = dyb, None
            if dy is None and ctx.fork_in:     # only the skip connection carried a gradient
                return dskip, None, None, None, None, None, None, None, None, None, None
        z, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:
            if ctx.z_out:                      # the normalised output went unused: only the pre-norm sum carried a gradient
                dy = torch.zeros(ctx.shape, dtype=z.dtype, device=z.device)
            else:
                dy, dyb = dyb, None
        p, seed, off, dev = ctx.rng
        H = z.shape[-1]
        dy2 = dy.reshape(-1, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if dskip is not None:
            dskip = dskip.reshape(-1, H)
            if dskip.dtype != z.dtype:
                dskip = dskip.to(z.dtype)
            if not dskip.is_contiguous():
                dskip = dskip.contiguous()
        if dyb is not None:
            dyb = dyb.reshape(-1, H)
            if dyb.dtype != dy2.dtype:
                dyb = dyb.to(dy2.dtype)
            if not dyb.is_contiguous():
                dyb = dyb.contiguous()
        M = z.shape[0]
        L = _lib.lib()
        dx = torch.empty_like(z)
        dres = torch.empty_like(z) if (ctx.has_res and p > 0) else None
        # gradient of the summand behind the norm: the (masked) dy — without output dropout it IS dy: handed on as such, no copy
        post_alias = ctx.has_post and ctx.out_drop[0] == 0.0 and dyb is None
        dpost = torch.empty_like(z) if (ctx.has_post and not post_alias) else None
        sg, sb = _sink(ctx.gb[0]), _sink(ctx.gb[1])
        sunk = sg is not None and sb is not None
        if not sunk:
            _prep_fallback(*ctx.gb)
        dg = sg if sunk else torch.empty(H, dtype=torch.float32, device=z.device)
        db = sb if sunk else torch.empty(H, dtype=torch.float32, device=z.device)
        # LN_DETERMINISTIC: per-block partials in a workspace + a second (reduction) launch; default: the blocks add their column
        # partials to dgamma / dbeta with float atomics (43 fewer launches per step; summation order is not reproducible)
        # (measured, scripts/ln_bench.py: atomics 15.8 vs 16.7 us at 3840 rows and one launch fewer; 26.3 vs 22.2 us at 8640 rows)
        acc = int(sunk and not _first_touch(*ctx.gb))
        defer = acc == 1 and LnReduceQueue.enabled and M >= LnReduceQueue.MIN_ROWS
        if defer:       # arena slices (pre-zeroed, accumulating): leave the column partials behind, one reduction per backward pass
            nparts = L.goat_ln_bwd_nparts(M)
            ws = torch.empty(nparts * 2 * H, dtype=torch.float32, device=z.device)
            acc = 2
        else:
            ws = torch.empty(L.goat_ln_bwd_ws_floats(H), dtype=torch.float32, device=z.device) if (LN_DETERMINISTIC or M > LN_ATOMIC_MAX_ROWS) else None
        st = L.goat_ln_bwd_do(_stream(), _dt(z), _ptr(dy2), _ptr(dyb) if dyb is not None else None, _ptr(z), _ptr(gamma), _ptr(mean), _ptr(rstd),
                              p, seed, off, dev, _ptr(dx), _ptr(dres) if dres is not None else None,
                              _ptr(dg), _ptr(db), _ptr(ws) if ws is not None else None, M, H,
                              acc | (4 if (ctx.z_out and dskip is not None) else 0), _ptr(dskip) if dskip is not None else None,
                              ctx.out_drop[0], ctx.out_drop[1], _ptr(dpost) if dpost is not None else None)
        _lib.check(st, 'goat_ln_bwd')
        if defer:
            LnReduceQueue.push(ws, dg, db, nparts, H)
        if sunk:
            dg = db = None
        dxv = dx.view(ctx.shape)
        if ctx.has_res:
            dr = dres.view(ctx.shape) if dres is not None else dxv
        else:
            dr = None
        return dxv, dr, dg, db, None, None, None, None, None, None, (dy2.view(ctx.shape) if post_alias else (dpost.view(ctx.shape) if dpost is not None else None))


def layer_norm(x, gamma, beta, eps, residual=None, p=0.0, fork=False, fork_in=False, z_out=False, p_out=0.0, post_add=None):
    """fork_in=True (pre-LN blocks): returns (LayerNorm(x), x) — use the second output for the skip connection; the gradient it
    receives is added inside the LayerNorm backward kernel instead of by an autograd add.
    z_out=True (pre-LN blocks, with residual): returns (LayerNorm(z), z) with z = residual + dropout_p(x) — the residual junction in
    FRONT of the LayerNorm and the LayerNorm in one launch per direction; z continues as the block's hidden state and the gradient
    it collects later joins inside this LayerNorm's backward kernel.
    p_out: dropout on the LayerNorm's output in the same launch (the embedding blocks' dropout(LayerNorm(e)));
    post_add: a summand added behind the norm and in front of that dropout: dropout(LayerNorm(z) + post_add)."""
    return _LnFn.apply(x, residual, gamma, beta, float(eps), float(p), bool(fork), bool(fork_in), bool(z_out), float(p_out), post_add)


# ----------------------------------------------------------------------------- gradient fan-in / fills (csrc/glue.hip)
FANOUT = os.environ.get('GOAT_NO_FANOUT', '0') != '1'        # (diagnostics: A/B against the autograd engine's pairwise adds)


def add_n(tensors, out=None):
    """out = sum(tensors) (same shape / dtype, contiguous), float32 accumulation, one launch (goat_add_n)."""
    ts = [t if t.is_contiguous() else t.contiguous() for t in tensors]
    out = torch.empty_like(ts[0]) if out is None else out
    while len(ts) > 8:                       # (never on the GOAT paths: at most 7 consumers)
        head = add_n(ts[:8])
        ts = [head] + ts[8:]
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    st = _lib.lib().goat_add_n(_stream(), _dt(ts[0]), arr, len(ts), _ptr(out), ts[0].numel())
    _lib.check(st, 'goat_add_n')
    return out


class _FanoutFn(torch.autograd.Function):
    """n autograd handles on ONE buffer; backward = the sum of the handles' gradients in one launch."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        dt = gs[0].dtype
        gs = [g if g.dtype == dt else g.to(dt) for g in gs]
        return add_n(gs), None


def fanout(x, n):
    """A tensor with n consumers: returns n handles on x's buffer, one per consumer.  Their gradients meet in ONE goat_add_n launch
    (float32 accumulation) instead of n - 1 pairwise `add` kernels issued by the autograd engine as the gradients trickle in."""
    if n <= 1 or not FANOUT or not (torch.is_tensor(x) and x.requires_grad and x.is_cuda):
        return [x] * max(n, 1)
    return list(_FanoutFn.apply(x, n))


def fanout_tree(obj, n):
    """fanout() over every tensor of a nested list / tuple / dict: -> list of n objects of the same structure (the K|V projections of an
    instruction, read by every step of a navigation episode: their gradients meet in one launch per tensor instead of T - 1 adds)."""
    if torch.is_tensor(obj):
        return fanout(obj, n)
    if isinstance(obj, dict):
        parts = {k: fanout_tree(v, n) for k, v in obj.items()}
        return [{k: parts[k][i] for k in obj} for i in range(n)]
    if isinstance(obj, (list, tuple)):
        parts = [fanout_tree(v, n) for v in obj]
        return [type(obj)(p[i] for p in parts) for i in range(n)]
    return [obj] * n


class _CeRowsFn(torch.autograd.Function):
    """loss[m] = logsumexp(logits[m, :]) - logits[m, target[m]] (float32; a negative target = ignored row, loss 0): F.cross_entropy(reduction=
    'none', ignore_index=-100) on the action logits of a navigation step (M/r2r/agent.py:614-616) as one launch per direction
    (goat_ce_fwd / _bwd) instead of log_softmax + nll_loss and their two backward kernels.  Masked actions carry -inf logits."""

    @staticmethod
    def forward(ctx, logits, targets):
        lg = logits.float().contiguous()
        M, N = lg.shape
        tg = targets.to(torch.int64).contiguous()
        loss = torch.empty(M, dtype=torch.float32, device=lg.device)
        lse = torch.empty(M, dtype=torch.float32, device=lg.device)
        st = _lib.lib().goat_ce_fwd(_stream(), _ptr(lg), N, M, N, _ptr(tg), _ptr(loss), _ptr(lse))
        _lib.check(st, 'goat_ce_fwd')
        ctx.save_for_backward(lg, tg, lse)
        ctx.in_dtype = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, tg, lse = ctx.saved_tensors
        M, N = lg.shape
        dl = torch.empty_like(lg)
        st = _lib.lib().goat_ce_bwd(_stream(), GOAT_F32, _ptr(lg), N, M, N, _ptr(tg), _ptr(lse), _ptr(dloss.float().contiguous()), _ptr(dl), N)
        _lib.check(st, 'goat_ce_bwd')
        return (dl if ctx.in_dtype == torch.float32 else dl.to(ctx.in_dtype)), None


def cross_entropy_rows(logits, targets, ignore_index=-100):
    """per-row cross-entropy; rows whose target is NEGATIVE are ignored by the kernel (loss 0, zero gradient): `ignore_index` must be
    negative, and is what the torch route below (row widths that are no multiple of 4: goat_ce_* reads 16-byte pieces) is told.
    A target >= the row width is a caller bug: torch raises, the kernel returns NaN for that row (it never reads out of bounds)."""
    if ignore_index >= 0:
        raise ValueError('cross_entropy_rows ignores negative targets only (ignore_index = %d)' % ignore_index)
    _need_gpu(logits)
    if logits.shape[1] % 4:
        return torch.nn.functional.cross_entropy(logits.float(), targets, reduction='none', ignore_index=ignore_index)
    return _CeRowsFn.apply(logits, targets)


def zero_ranges(tensors):
    """clear up to 16 contiguous tensors (16-byte aligned, sizes multiples of 16 bytes) per launch (goat_zero_ranges)."""
    ts = [t for t in tensors if t.numel()]
    for i in range(0, len(ts), 16):
        grp = ts[i:i + 16]
        if any((t.data_ptr() & 15) or ((t.numel() * t.element_size()) & 15) or not t.is_contiguous() or not t.is_cuda for t in grp):
            for t in grp:
                t.zero_()
            continue
        ptrs = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        nb = (ctypes.c_int64 * len(grp))(*[t.numel() * t.element_size() for t in grp])
        st = _lib.lib().goat_zero_ranges(_stream(), ptrs, nb, len(grp))
        _lib.check(st, 'goat_zero_ranges')


class _DropAddFn(torch.autograd.Function):
    """y = residual + dropout_p(x)."""

    @staticmethod
    def forward(ctx, x, residual, p):
        _need_gpu(x)
        xc = x if x.is_contiguous() else x.contiguous()
        rc = None
        if residual is not None:
            rc = residual if residual.is_contiguous() else residual.contiguous()
        y = torch.empty_like(xc)
        seed, off, dev = RngState.next(xc.numel()) if p > 0 else (0, 0, None)
        st = _lib.lib().goat_dropout_add_fwd(_stream(), _dt(xc), _ptr(xc), _ptr(rc) if rc is not None else None,
                                             _ptr(y), xc.numel(), p, seed, off, dev)
        _lib.check(st, 'goat_dropout_add_fwd')
        ctx.rng = (p, seed, off, dev)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, off, dev = ctx.rng
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        if p > 0:
            dx = torch.empty_like(dyc)
            st = _lib.lib().goat_dropout_bwd(_stream(), _dt(dyc), _ptr(dyc), _ptr(dx), dyc.numel(), p, seed, off, dev)
            _lib.check(st, 'goat_dropout_bwd')
        else:
            dx = dyc
        return dx, (dyc if ctx.has_res else None), None


def dropout_add(x, residual, p):
    if p <= 0 and residual is None:
        return x
    return _DropAddFn.apply(x, residual, float(p))


def dropout(x, p):
    return dropout_add(x, None, p)


# ----------------------------------------------------------------------------- attention
class _AttnFn(torch.autograd.Function):
    """Masked MHA on packed projections.
    mode 'self' : a = qkv [B,L,3H]              (q|k|v)
    mode 'cross': a = q [B,Lq,H], b = kv [B,Lk,2H]   (k|v)
    kmask: float32 [B,Lk] additive (0 / -10000 / -inf) or None; bias: float32 [B,Lq,Lk] or None."""

    @staticmethod
    def forward(ctx, a, b, kmask, bias, nh, p):
        _need_gpu(a)
        a = a if a.is_contiguous() else a.contiguous()
        slot = None
        if b is not None:
            # a view of a projection bank (linear_bank): rows at the bank's leading dimension; its gradient goes into the bank's buffer
            # (read through its strides; the FIRST consumer of the view claims the slot — a second reader of the same view, e.g. the steps of
            # an eager episode that share one projection of the instruction, gets a buffer of its own and autograd adds the two)
            if b.dim() == 3 and b.stride(2) == 1 and b.stride(0) == b.shape[1] * b.stride(1) and b.stride(1) % 8 == 0 and b.data_ptr() % 16 == 0:
                slot = b.__dict__.pop('_goat_grad_slot', None)
            else:
                b = b.contiguous()
        ctx.slot = slot
        if b is None:
            B, Lq, H3 = a.shape
            H = H3 // 3
            Lk = Lq
            q = (a, 0, H3, Lq * H3)
            k = (a, H, H3, Lq * H3)
            v = (a, 2 * H, H3, Lq * H3)
        else:
            B, Lq, H = a.shape
            Lk = b.shape[1]
            ldb = b.stride(1)
            q = (a, 0, H, Lq * H)
            k = (b, 0, ldb, Lk * ldb)
            v = (b, H, ldb, Lk * ldb)
        assert H == nh * 64, 'head_dim must be 64'
        o = torch.empty((B, Lq, H), dtype=a.dtype, device=a.device)
        lse = torch.empty((B, nh, Lq), dtype=torch.float32, device=a.device)
        if kmask is not None:
            kmask = kmask.contiguous().float()
        if bias is not None:
            bias = bias.contiguous().float()
        seed, off, dev = RngState.next(B * nh * Lq * Lk) if p > 0 else (0, 0, None)
        scale = 1.0 / math.sqrt(64.0)
        st = _lib.lib().goat_attn_fwd(
            _stream(), _dt(a),
            _ptr(q[0], q[1]), q[2], q[3], _ptr(k[0], k[1]), k[2], k[3], _ptr(v[0], v[1]), v[2], v[3],
            _ptr(o), H, Lq * H,
            _ptr(kmask) if kmask is not None else None, _ptr(bias) if bias is not None else None, _ptr(lse),
            B, nh, Lq, Lk, scale, p, seed, off, dev)
        _lib.check(st, 'goat_attn_fwd(Lq=%d,Lk=%d)' % (Lq, Lk))
        ctx.save_for_backward(a, b, kmask, bias, o, lse)
        ctx.cfg = (nh, p, seed, off, dev, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        a, b, kmask, bias, o, lse = ctx.saved_tensors
        nh, p, seed, off, dev, scale = ctx.cfg
        do = do if do.is_contiguous() else do.contiguous()
        da = torch.empty_like(a)
        db = None
        if b is not None:
            db = ctx.slot[0].view(ctx.slot[1]) if ctx.slot is not None else torch.empty(b.shape, dtype=b.dtype, device=b.device)
        if b is None:
            B, Lq, H3 = a.shape
            H = H3 // 3
            Lk = Lq
            q, k, v = (a, 0, H3, Lq * H3), (a, H, H3, Lq * H3), (a, 2 * H, H3, Lq * H3)
            dq, dk, dv = (da, 0, H3, Lq * H3), (da, H, H3, Lq * H3), (da, 2 * H, H3, Lq * H3)
        else:
            B, Lq, H = a.shape
            Lk = b.shape[1]
            ldb, ldg = b.stride(1), db.stride(1)
            q, k, v = (a, 0, H, Lq * H), (b, 0, ldb, Lk * ldb), (b, H, ldb, Lk * ldb)
            dq, dk, dv = (da, 0, H, Lq * H), (db, 0, ldg, Lk * ldg), (db, H, ldg, Lk * ldg)
        dbias = None
        if bias is not None and ctx.needs_input_grad[3]:
            dbias = torch.zeros_like(bias)
        st = _lib.lib().goat_attn_bwd(
            _stream(), _dt(a),
            _ptr(q[0], q[1]), q[2], q[3], _ptr(k[0], k[1]), k[2], k[3], _ptr(v[0], v[1]), v[2], v[3],
            _ptr(o), H, Lq * H, _ptr(do), H, Lq * H,
            _ptr(dq[0], dq[1]), dq[2], dq[3], _ptr(dk[0], dk[1]), dk[2], dk[3], _ptr(dv[0], dv[1]), dv[2], dv[3],
            _ptr(kmask) if kmask is not None else None, _ptr(bias) if bias is not None else None, _ptr(lse),
            _ptr(dbias) if dbias is not None else None,
            B, nh, Lq, Lk, scale, p, seed, off, dev)
        _lib.check(st, 'goat_attn_bwd(Lq=%d,Lk=%d)' % (Lq, Lk))
        return da, db, None, dbias, None, None


def attention(a, b, kmask, bias, nh, p):
    return _AttnFn.apply(a, b, kmask, bias, int(nh), float(p))


# ----------------------------------------------------------------------------- pano fusion / gather
class _PanoFusionFn(torch.autograd.Function):
    """fused[n] = sum_v softmax_v(tanh(x[n,v]·a + a0)) x[n,v]  (P/model/vilmodel_goat.py:354-361)."""

    @staticmethod
    def forward(ctx, x, a_w, a_b):
        _need_gpu(x)
        x = x if x.is_contiguous() else x.contiguous()
        N, V, H = x.shape
        fused = torch.empty((N, H), dtype=x.dtype, device=x.device)
        wsave = torch.empty((N, V), dtype=torch.float32, device=x.device)
        av = a_w.detach().reshape(-1).contiguous()
        st = _lib.lib().goat_pano_fusion_fwd(_stream(), _dt(x), _ptr(x), _ptr(av), _ptr(a_b), _ptr(fused),
                                             _ptr(wsave), N, V, H)
        _lib.check(st, 'goat_pano_fusion_fwd')
        ctx.save_for_backward(x, av, a_b, wsave)
        ctx.wshape = a_w.shape
        ctx.params = (a_w, a_b)
        return fused

    @staticmethod
    def backward(ctx, df):
        x, av, a_b, wsave = ctx.saved_tensors
        N, V, H = x.shape
        df = df if df.is_contiguous() else df.contiguous()
        dx = torch.empty_like(x)
        sa, sb = _sink(ctx.params[0]), _sink(ctx.params[1])
        sunk = PANO_SINK and sa is not None and sb is not None and sa.is_contiguous()
        if sunk:                 # arena slices: the kernel's atomics add straight into them
            for prm, snk in zip(ctx.params, (sa, sb)):
                if _first_touch(prm):
                    snk.zero_()
            da, da0 = sa.view(-1), sb.view(-1)
        else:
            _prep_fallback(*ctx.params)
            da = torch.zeros(H, dtype=torch.float32, device=x.device)
            da0 = torch.zeros(1, dtype=torch.float32, device=x.device)
        st = _lib.lib().goat_pano_fusion_bwd(_stream(), _dt(x), _ptr(x), _ptr(av), _ptr(a_b), _ptr(wsave), _ptr(df),
                                             _ptr(dx), _ptr(da), _ptr(da0), N, V, H)
        _lib.check(st, 'goat_pano_fusion_bwd')
        if sunk:
            return dx, None, None
        return dx, da.view(ctx.wshape), da0


def pano_fusion(x, a_w, a_b):
    return _PanoFusionFn.apply(x, a_w, a_b)


class _GatherFn(torch.autograd.Function):
    """out[i] = scale[i] * sum_{j in seg i} src[idx[j]]  (index tensors built on the host per batch)."""

    @staticmethod
    def forward(ctx, src, idx, start, scale, n_out, inverse):
        _need_gpu(src)
        src = src if src.is_contiguous() else src.contiguous()
        H = src.shape[-1]
        s2 = src.reshape(-1, H)
        out = torch.empty((n_out, H), dtype=src.dtype, device=src.device)
        st = _lib.lib().goat_gather_segmean_fwd(_stream(), _dt(s2), _ptr(s2), s2.shape[0], _ptr(idx), _ptr(start),
                                                _ptr(scale) if scale is not None else None, _ptr(out), n_out, H, None)
        _lib.check(st, 'goat_gather_segmean_fwd')
        ctx.save_for_backward(idx, start, scale)
        ctx.sshape, ctx.sdtype = src.shape, src.dtype
        ctx.inverse = inverse
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, start, scale = ctx.saved_tensors
        dout = dout if dout.is_contiguous() else dout.contiguous()
        n_out, H = dout.shape
        rows = 1
        for s in ctx.sshape[:-1]:
            rows *= s
        if ctx.inverse is not None:          # inverse index built on the host with the forward one: the backward pass is a gather too
            inv_idx, inv_start = ctx.inverse[0], ctx.inverse[1]
            inv_w = ctx.inverse[2] if len(ctx.inverse) > 2 else None
            if inv_start.numel() != rows + 1:
                raise ValueError('inverse gather index covers %d rows, the source has %d' % (inv_start.numel() - 1, rows))
            d2 = dout if dout.dtype == ctx.sdtype else dout.to(ctx.sdtype)
            dsrc = torch.empty((rows, H), dtype=ctx.sdtype, device=dout.device)
            st = _lib.lib().goat_gather_segmean_fwd(_stream(), _dt(d2), _ptr(d2), n_out, _ptr(inv_idx), _ptr(inv_start), None, _ptr(dsrc),
                                                    rows, H, _ptr(inv_w) if inv_w is not None else None)
            _lib.check(st, 'goat_gather_segmean_fwd (inverse)')
            return dsrc.view(ctx.sshape), None, None, None, None, None
        d32 = torch.zeros((rows, H), dtype=torch.float32, device=dout.device)
        st = _lib.lib().goat_gather_segmean_bwd(_stream(), _dt(dout), _ptr(dout), _ptr(idx), _ptr(start),
                                                _ptr(scale) if scale is not None else None, _ptr(d32), n_out, H)
        _lib.check(st, 'goat_gather_segmean_bwd')
        return d32.to(ctx.sdtype).view(ctx.sshape), None, None, None, None, None


def gather_segmean(src, idx, start, scale, n_out, inverse=None):
    """inverse: graphmap.inverse_index(idx, start, scale, n_src) on the device (optional) — the backward pass then gathers instead
    of scatter-adding with float atomics into a zero-filled float32 buffer."""
    return _GatherFn.apply(src, idx, start, scale, int(n_out), inverse)


# ----------------------------------------------------------------------------- embedding tables
_EMBED_ERR = {}


def _embed_err(dev):
    t = _EMBED_ERR.get(dev)
    if t is None:
        t = _EMBED_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def check_embed_errors():
    """Raises if any embedding lookup since the last call saw an id outside its table (synchronises)."""
    for dev, t in _EMBED_ERR.items():
        if int(t.item()):
            t.zero_()
            raise IndexError('embedding id out of range on %s' % (dev,))


class SparseEmbedGrad:
    """Data-parallel hook for the word-embedding table.  In a step whose only gradient of the table comes from the lookups
    (sap, cfp: at most B*L of its 50 265 rows are touched) all-reducing the dense 154 MB table gradient is waste: with
    `sink_list` set (dp.GoatDataParallel.begin_step), _EmbedFn.backward does not scatter into a table listed in `params`
    but hands (d_out rows, ids, table, padding index) over; dp all-gathers rows + ids (5.9 MB per rank) and scatter-adds
    every rank's rows locally (GoatDataParallel.reduce_gradients).  Inactive at world size 1 and in mlm steps (the tied
    decoder makes the table gradient dense)."""
    params = set()          # id(parameter) of the tables handled this way
    sink_list = None        # list to append to during this step's backward, or None


class _EmbedFn(torch.autograd.Function):
    """out = word[ids] (+ type[type_ids or 0]) (+ pos[arange(L)]) in one pass; backward scatters into f32 table
    gradients (P/model/Bert_backbone.py:98-113).  Replaces three F.embedding calls and torch's sort-based
    embedding backward."""

    @staticmethod
    def forward(ctx, ids, word, type_tab, type_ids, pos_tab, out_dtype, word_pad, pos_pad):
        _need_gpu(word)
        ids = ids if ids.is_contiguous() else ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.long()
        if type_ids is not None:
            type_ids = type_ids.long().contiguous()
        V, H = word.shape
        rows = ids.numel()
        L = ids.shape[-1]
        if pos_tab is not None and L > pos_tab.shape[0]:
            raise IndexError('sequence length %d exceeds the position table (%d rows)' % (L, pos_tab.shape[0]))
        out = torch.empty(tuple(ids.shape) + (H,), dtype=out_dtype, device=word.device)
        st = _lib.lib().goat_embed_fwd(_stream(), _dt(out), _ptr(word), _ptr(ids), _ptr(type_tab) if type_tab is not None else None,
                                       _ptr(type_ids) if type_ids is not None else None,
                                       _ptr(pos_tab) if pos_tab is not None else None, L, _ptr(out), rows, H, V,
                                       _ptr(_embed_err(word.device)))
        _lib.check(st, 'goat_embed_fwd')
        ctx.tabs = (word, type_tab, pos_tab)
        ctx.save_for_backward(ids, type_ids)
        ctx.meta = (V, H, L, None if type_tab is None else type_tab.shape[0], None if pos_tab is None else pos_tab.shape[0],
                    -1 if word_pad is None else int(word_pad), -1 if pos_pad is None else int(pos_pad))
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, type_ids = ctx.saved_tensors
        V, H, L, TV, P, word_pad, pos_pad = ctx.meta
        dout = dout if dout.is_contiguous() else dout.contiguous()
        d2 = dout.view(-1, H)
        rows = d2.shape[0]
        need_w, need_t, need_p = ctx.needs_input_grad[1], TV is not None and ctx.needs_input_grad[2], \
            P is not None and ctx.needs_input_grad[4]
        dev = dout.device
        sw, st_, sp = (_sink(t) for t in ctx.tabs)
        if need_w and sw is not None and SparseEmbedGrad.sink_list is not None and id(ctx.tabs[0]) in SparseEmbedGrad.params:
            SparseEmbedGrad.sink_list.append((d2, ids, ctx.tabs[0], word_pad))      # exchanged and scattered by dp after backward
            need_w = False
        for t, sk, need in zip(ctx.tabs, (sw, st_, sp), (need_w, need_t, need_p)):
            if sk is not None and need and _first_touch(t):
                sk.zero_()              # first writer of this slice in the step: clear it (scatter-adds follow)
        dword = (sw if sw is not None else torch.zeros((V, H), dtype=torch.float32, device=dev)) if need_w else None
        dtab = (st_ if st_ is not None else torch.zeros((TV, H), dtype=torch.float32, device=dev)) if need_t else None
        dpos = (sp if sp is not None else torch.zeros((P, H), dtype=torch.float32, device=dev)) if need_p else None
        if need_w or (need_t and type_ids is not None):
            st = _lib.lib().goat_embed_bwd(_stream(), _dt(d2), _ptr(d2), _ptr(ids), _ptr(type_ids) if type_ids is not None else None,
                                           L, _ptr(dword) if need_w else None,
                                           _ptr(dtab) if (need_t and type_ids is not None) else None,
                                           None, rows, H, V, word_pad, pos_pad)
            _lib.check(st, 'goat_embed_bwd')
        if need_p:
            # position ids are arange(L) for every sample: d pos[:L] = column sums of dout viewed as [rows/L, L*H]
            # (the padding row, if any, is skipped: nn.Embedding(padding_idx) gives it no gradient)
            flat, of = d2.view(rows // L, L * H), dpos.view(-1)
            if 0 <= pos_pad < L:
                if pos_pad > 0:
                    colsum(flat[:, :pos_pad * H], out=of[:pos_pad * H])
                if pos_pad + 1 < L:
                    colsum(flat[:, (pos_pad + 1) * H:], out=of[(pos_pad + 1) * H:L * H])
            else:
                colsum(flat, out=of[:L * H])
        if need_t and type_ids is None:
            colsum(d2, out=dtab[0])            # every token has type 0: one column sum instead of `rows` atomics per column
        return (None, None if (dword is sw or dword is None) else dword, None if dtab is st_ else dtab, None,
                None if dpos is sp else dpos, None, None, None)


def embedding_scatter_add(dword, rows, ids, word_pad=-1):
    """dword[ids[r], :] += rows[r, :] (float32 table gradient, atomics) — goat_embed_bwd on the word table only."""
    rows = rows if rows.is_contiguous() else rows.contiguous()
    ids = ids.reshape(-1).contiguous()
    st = _lib.lib().goat_embed_bwd(_stream(), _dt(rows), _ptr(rows), _ptr(ids), None, 1, _ptr(dword), None, None,
                                   rows.shape[0], rows.shape[1], dword.shape[0], -1 if word_pad is None else int(word_pad), -1)
    _lib.check(st, 'goat_embed_bwd')


def embedding(ids, word, type_tab=None, type_ids=None, pos_tab=None, out_dtype=None, word_pad=None, pos_pad=None):
    return _EmbedFn.apply(ids, word, type_tab, type_ids, pos_tab, out_dtype or word.dtype, word_pad, pos_pad)


# ----------------------------------------------------------------------------- causal-learning heads
class _AttnPoolFn(torch.autograd.Function):
    """out = tanh(sum_l softmax_l(tanh(x_l)·w) x_l), all slots, no padding mask (P/model/pretrain_goat.py:502-515)."""

    @staticmethod
    def forward(ctx, x, w, smask=None):
        _need_gpu(x)
        x = x if x.is_contiguous() else x.contiguous()
        B, L, H = x.shape
        wv = w.detach().reshape(-1).float().contiguous()
        out = torch.empty((B, H), dtype=torch.float32, device=x.device)
        attn = torch.empty((B, L), dtype=torch.float32, device=x.device)
        ws = torch.empty(B * L, dtype=torch.float32, device=x.device)
        if smask is not None:
            assert smask.shape == (B, L) and smask.dtype == torch.float32 and smask.is_contiguous()
        st = _lib.lib().goat_attn_pool_fwd(_stream(), _dt(x), _ptr(x), _ptr(wv), _ptr(out), _ptr(attn), _ptr(ws), B, L, H,
                                           _ptr(smask) if smask is not None else None)
        _lib.check(st, 'goat_attn_pool_fwd')
        ctx.save_for_backward(x, wv, attn, out)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wv, attn, out = ctx.saved_tensors
        B, L, H = x.shape
        dout = dout.float().contiguous()
        dx = torch.empty_like(x)
        dw = torch.zeros(H, dtype=torch.float32, device=x.device)
        ws = torch.empty(B * L, dtype=torch.float32, device=x.device)
        st = _lib.lib().goat_attn_pool_bwd(_stream(), _dt(x), _ptr(x), _ptr(wv), _ptr(attn), _ptr(out), _ptr(dout), _ptr(dx),
                                           _ptr(dw), _ptr(ws), B, L, H)
        _lib.check(st, 'goat_attn_pool_bwd')
        return dx, dw.view(ctx.wshape), None


def attn_pool(x, w, smask=None):
    """smask: optional float32 [B, L] of 0 / -inf added to the scores (slots beyond the batch's own padded width in a
    shape-bucketed static batch); masked slots get weight 0 and a zero gradient."""
    return _AttnPoolFn.apply(x, w, smask)


class _DoorGateFn(torch.autograd.Function):
    """s = sigmoid(Linear_a(aug) + Linear_o(ori)) per token; out = s*aug + (1-s)*ori (P/model/vilmodel_goat.py:137-143)."""

    @staticmethod
    def forward(ctx, aug, ori, wa, ba, wo, bo):
        _need_gpu(aug)
        H = aug.shape[-1]
        a2 = aug.reshape(-1, H)
        o2 = ori.reshape(-1, H).to(a2.dtype)
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        o2 = o2 if o2.is_contiguous() else o2.contiguous()
        wav, wov = wa.detach().reshape(-1).contiguous(), wo.detach().reshape(-1).contiguous()
        rows = a2.shape[0]
        out = torch.empty_like(a2)
        gate = torch.empty(rows, dtype=torch.float32, device=a2.device)
        st = _lib.lib().goat_door_gate_fwd(_stream(), _dt(a2), _ptr(a2), _ptr(o2), _ptr(wav), _ptr(wov), _ptr(ba.detach()),
                                           _ptr(bo.detach()), _ptr(out), _ptr(gate), rows, H)
        _lib.check(st, 'goat_door_gate_fwd')
        ctx.save_for_backward(a2, o2, wav, wov, gate)
        ctx.shapes = (aug.shape, ori.shape, ori.dtype, wa.shape, wo.shape)
        ctx.params = (wa, ba, wo, bo)
        return out.view(aug.shape)

    @staticmethod
    def backward(ctx, dout):
        a2, o2, wav, wov, gate = ctx.saved_tensors
        ashape, oshape, odtype, washape, woshape = ctx.shapes
        rows, H = a2.shape
        d2 = dout.reshape(rows, H).to(a2.dtype)
        d2 = d2 if d2.is_contiguous() else d2.contiguous()
        daug, dori = torch.empty_like(a2), torch.empty_like(a2)
        sinks = [_sink(p_) for p_ in ctx.params]
        if all(t is not None for t in sinks):
            # gradient arena: the four gate parameters are small (cleared by GradArena.zero at the start of the step, never a first touch) —
            # the kernel's atomics add straight into their slices; a gate used in every step of an episode would otherwise cost a zero fill,
            # a clone and four AccumulateGrad adds per use
            for p_ in ctx.params:
                if _first_touch(p_):
                    _sink(p_).zero_()
            st = _lib.lib().goat_door_gate_bwd(_stream(), _dt(a2), _ptr(a2), _ptr(o2), _ptr(wav), _ptr(wov), _ptr(gate), _ptr(d2),
                                               _ptr(daug), _ptr(dori), _ptr(sinks[0]), _ptr(sinks[2]), _ptr(sinks[1]), rows, H, _ptr(sinks[3]))
            _lib.check(st, 'goat_door_gate_bwd')
            return daug.view(ashape), dori.view(oshape).to(odtype), None, None, None, None
        _prep_fallback(*ctx.params)
        buf = torch.zeros(2 * H + 1, dtype=torch.float32, device=a2.device)
        st = _lib.lib().goat_door_gate_bwd(_stream(), _dt(a2), _ptr(a2), _ptr(o2), _ptr(wav), _ptr(wov), _ptr(gate), _ptr(d2),
                                           _ptr(daug), _ptr(dori), _ptr(buf), _ptr(buf, H), _ptr(buf, 2 * H), rows, H, None)
        _lib.check(st, 'goat_door_gate_bwd')
        db = buf[2 * H:]          # both biases receive the same gradient (distinct tensors: autograd may keep them as .grad)
        return daug.view(ashape), dori.view(oshape).to(odtype), buf[:H].view(washape), db, buf[H:2 * H].view(woshape), db.clone()


def door_gate(aug_lin, ori_lin, aug, ori):
    """aug_lin / ori_lin: the two Linear(H,1) modules of the gate (their weights [1,H] and biases [1])."""
    return _DoorGateFn.apply(aug, ori, aug_lin.weight, aug_lin.bias, ori_lin.weight, ori_lin.bias)


class _DictWsumFn(torch.autograd.Function):
    """out[b,1,:] = sum_k p[b,k] z[b,k,:] (BACL type_1 confounder expectation, P/model/vilmodel_goat.py:115-118)."""

    @staticmethod
    def forward(ctx, z, p, out_dtype):
        _need_gpu(z)
        zf = z.float().contiguous()
        pf = p.float().reshape(p.shape[0], p.shape[1]).contiguous()
        B, K, H = zf.shape
        out = torch.empty((B, 1, H), dtype=out_dtype, device=z.device)
        st = _lib.lib().goat_dict_wsum_fwd(_stream(), _dt(out), _ptr(zf), _ptr(pf), _ptr(out), B, K, H)
        _lib.check(st, 'goat_dict_wsum_fwd')
        ctx.save_for_backward(zf, pf)
        ctx.meta = (z.dtype, p.shape, p.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        zf, pf = ctx.saved_tensors
        zdtype, pshape, pdtype = ctx.meta
        B, K, H = zf.shape
        d2 = dout.reshape(B, H)
        d2 = d2 if d2.is_contiguous() else d2.contiguous()
        dz = torch.empty_like(zf) if ctx.needs_input_grad[0] else None
        dp = torch.empty_like(pf) if ctx.needs_input_grad[1] else None
        st = _lib.lib().goat_dict_wsum_bwd(_stream(), _dt(d2), _ptr(d2), _ptr(zf), _ptr(pf), _ptr(dz) if dz is not None else None,
                                           _ptr(dp) if dp is not None else None, B, K, H)
        _lib.check(st, 'goat_dict_wsum_bwd')
        return (dz.to(zdtype) if dz is not None else None, dp.view(pshape).to(pdtype) if dp is not None else None, None)


def dict_weighted_sum(z, p, out_dtype):
    return _DictWsumFn.apply(z, p, out_dtype)


if not os.environ.get('GOAT_RETUNE'):       # (GOAT_RETUNE=1: start from an empty table and time every shape again)
    load_tuned()


# ----------------------------------------------------------------------------- SAP logits / loss
def _u8(t):
    """bool / uint8 mask -> contiguous byte tensor (a view for bool: same storage)."""
    if t is None:
        return None
    t = t if t.is_contiguous() else t.contiguous()
    return t.view(torch.uint8) if t.dtype == torch.bool else t


class _SapFuseFn(torch.autograd.Function):
    """Tail of the single-action-prediction head in one launch per direction (goat_sap_fuse_fwd / _bwd; formulas in include/goat_hip.h;
    P/model/pretrain_goat.py:375-413, M/models/vilmodel_GOAT.py:803-839).  -> (gl, ll, fused, loss); loss is None without labels."""

    @staticmethod
    def forward(ctx, gs, ls, fwl, fw_sigmoid, gvis, gvalid, glens, lmask, lmask_is_valid, M, add_stop, ga, la):
        _need_gpu(gs)
        ctx.set_materialize_grads(False)
        B, G = gs.shape
        W = ls.shape[1]
        dt = gs.dtype
        gs = gs.contiguous()
        ls = ls.to(dt).contiguous()
        ctx.fw_shape = None if fwl is None else fwl.shape
        fwl = None if fwl is None else fwl.reshape(B).to(dt).contiguous()
        gvis, gvalid, lmask = _u8(gvis), _u8(gvalid), _u8(lmask)
        glens = None if glens is None else glens.to(torch.int64).contiguous()
        M = None if M is None else M.float().contiguous()
        ga = None if ga is None else ga.to(torch.int64).contiguous()
        la = None if la is None else la.to(torch.int64).contiguous()
        dev = gs.device
        gl = torch.empty((B, G), dtype=torch.float32, device=dev)
        ll = torch.empty((B, W), dtype=torch.float32, device=dev)
        fused = torch.empty((B, G), dtype=torch.float32, device=dev)
        labels = ga is not None
        loss = torch.empty(B, dtype=torch.float32, device=dev) if labels else None
        lse = torch.empty((B, 3), dtype=torch.float32, device=dev) if labels else None
        P = lambda t: _ptr(t) if t is not None else None
        ctx.cfg = (_dt(gs), int(bool(fw_sigmoid)), int(bool(lmask_is_valid)), int(bool(add_stop)), B, G, W)
        st = _lib.lib().goat_sap_fuse_fwd(_stream(), ctx.cfg[0], _ptr(gs), _ptr(ls), P(fwl), ctx.cfg[1], P(gvis), P(gvalid), P(glens),
                                          P(lmask), ctx.cfg[2], P(M), ctx.cfg[3], P(ga), P(la), _ptr(gl), _ptr(ll), _ptr(fused),
                                          P(loss), P(lse), B, G, W)
        _lib.check(st, 'goat_sap_fuse_fwd')
        ctx.opt = (fwl, gvis, gvalid, glens, lmask, M, ga, la, lse)
        ctx.save_for_backward(gs, ls, gl, ll, fused)
        return gl, ll, fused, loss

    @staticmethod
    def backward(ctx, dgl, dll, dfused, dloss):
        gs, ls, gl, ll, fused = ctx.saved_tensors
        fwl, gvis, gvalid, glens, lmask, M, ga, la, lse = ctx.opt
        dtc, fw_sig, lvalid, add_stop, B, G, W = ctx.cfg
        f = lambda t: None if t is None else t.float().contiguous()
        dgl, dll, dfused, dloss = f(dgl), f(dll), f(dfused), f(dloss)
        dgs, dls = torch.empty_like(gs), torch.empty_like(ls)
        dfw = torch.empty_like(fwl) if fwl is not None else None
        P = lambda t: _ptr(t) if t is not None else None
        st = _lib.lib().goat_sap_fuse_bwd(_stream(), dtc, _ptr(gs), _ptr(ls), P(fwl), fw_sig, P(gvis), P(gvalid), P(glens), P(lmask), lvalid,
                                          P(M), add_stop, P(ga), P(la), _ptr(gl), _ptr(ll), _ptr(fused), P(lse), P(dloss), P(dgl), P(dll),
                                          P(dfused), _ptr(dgs), _ptr(dls), P(dfw), B, G, W)
        _lib.check(st, 'goat_sap_fuse_bwd')
        return dgs, dls, (None if dfw is None else dfw.view(ctx.fw_shape)), None, None, None, None, None, None, None, None, None, None


def sap_fuse(gs, ls, fwl=None, fw_sigmoid=True, gvis=None, gvalid=None, glens=None, lmask=None, lmask_is_valid=False, M=None,
             add_stop=False, labels=None):
    """gs [B,G] / ls [B,W] head scores, fwl [B] or [B,1] fusion logit (None: weight 0.5) -> (global logits, local logits, fused logits,
    loss [B] or None), all float32.  labels = (global_act_labels, local_act_labels) adds the three cross-entropies."""
    ga, la = labels if labels is not None else (None, None)
    return _SapFuseFn.apply(gs, ls, fwl, fw_sigmoid, gvis, gvalid, glens, lmask, lmask_is_valid, M, add_stop, ga, la)


# ----------------------------------------------------------------------------- CFP contrastive losses
class _InfoNceFn(torch.autograd.Function):
    """loss[i] = sum_{x in gmap, vp, fused} 1/2 [CE(x_loc[i]·txt_all^T/tau, t0+i) + CE(txt_loc[i]·x_all^T/tau, t0+i)]
    (P/model/pretrain_goat.py:519-534) in one forward and one backward launch (goat_infonce_fwd / _bwd).  `*_all` are the
    candidates of every data-parallel rank (the same tensors as `*_loc` on one rank)."""

    @staticmethod
    def forward(ctx, g_loc, v_loc, f_loc, t_loc, g_all, v_all, f_all, t_all, target0, temperature):
        for t in (g_loc, v_loc, f_loc, t_loc, g_all, v_all, f_all, t_all):
            _need_gpu(t)
        ins = [t.float().contiguous() for t in (g_loc, v_loc, f_loc, t_loc, g_all, v_all, f_all, t_all)]
        Bl, H = ins[0].shape
        Ba = ins[4].shape[0]
        loss = torch.zeros(Bl, dtype=torch.float32, device=ins[0].device)
        prob = torch.empty((6, Bl, Ba), dtype=torch.float32, device=ins[0].device)
        xl = (ctypes.c_void_p * 3)(*[_ptr(t) for t in ins[0:3]])
        xa = (ctypes.c_void_p * 3)(*[_ptr(t) for t in ins[4:7]])
        st = _lib.lib().goat_infonce_fwd(_stream(), xl, xa, _ptr(ins[3]), _ptr(ins[7]), _ptr(loss), _ptr(prob), Bl, Ba, H,
                                         int(target0), float(temperature))
        _lib.check(st, 'goat_infonce_fwd')
        ctx.save_for_backward(prob, *ins)
        ctx.same = tuple(a is b for a, b in zip((g_loc, v_loc, f_loc, t_loc), (g_all, v_all, f_all, t_all)))
        ctx.cfg = (Bl, Ba, H, int(target0), float(temperature))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        prob, *ins = ctx.saved_tensors
        Bl, Ba, H, target0, temperature = ctx.cfg
        dev = prob.device
        dloc = torch.zeros((4, Bl, H), dtype=torch.float32, device=dev)
        need_all = [not s for s in ctx.same]
        dall = torch.zeros((4, Ba, H), dtype=torch.float32, device=dev) if any(need_all) else None
        gl = [dloc[k] for k in range(4)]
        ga = [(dall[k] if need_all[k] else dloc[k]) for k in range(4)]      # same tensor passed as loc and all: one gradient buffer
        xl = (ctypes.c_void_p * 3)(*[_ptr(t) for t in ins[0:3]])
        xa = (ctypes.c_void_p * 3)(*[_ptr(t) for t in ins[4:7]])
        dxl = (ctypes.c_void_p * 3)(*[_ptr(t) for t in gl[0:3]])
        dxa = (ctypes.c_void_p * 3)(*[_ptr(t) for t in ga[0:3]])
        st = _lib.lib().goat_infonce_bwd(_stream(), xl, xa, _ptr(ins[3]), _ptr(ins[7]), _ptr(dloss.float().contiguous()), _ptr(prob),
                                         dxl, dxa, _ptr(gl[3]), _ptr(ga[3]), Bl, Ba, H, target0, temperature)
        _lib.check(st, 'goat_infonce_bwd')
        return tuple(gl) + tuple((ga[k] if need_all[k] else None) for k in range(4)) + (None, None)


def _cfp_mix_fwd(go, vo, fwl):
    B, H = go.shape
    fo = torch.empty_like(go)
    fw = torch.empty(B, dtype=torch.float32, device=go.device)
    st = _lib.lib().goat_cfp_mix_fwd(_stream(), _dt(fwl), _ptr(go), _ptr(vo), _ptr(fwl), _ptr(fo), _ptr(fw), B, H)
    _lib.check(st, 'goat_cfp_mix_fwd')
    return fo, fw


def _cfp_mix_bwd(go, vo, fw, dfo, dgo, dvo, fwl_dtype, accumulate):
    B, H = go.shape
    dfwl = torch.empty(B, dtype=fwl_dtype, device=go.device)
    st = _lib.lib().goat_cfp_mix_bwd(_stream(), GOAT_BF16 if fwl_dtype == torch.bfloat16 else GOAT_F32, _ptr(go), _ptr(vo), _ptr(fw), _ptr(dfo),
                                     _ptr(dgo), _ptr(dvo), _ptr(dfwl), B, H, int(accumulate))
    _lib.check(st, 'goat_cfp_mix_bwd')
    return dfwl


class _CfpMixFn(torch.autograd.Function):
    """fo = go * sigmoid(fwl) + vo * (1 - sigmoid(fwl))  (float32 [B,H] pooled vectors, fusion logit [B] / [B,1] in the compute dtype):
    the fused CFP vector of P/model/pretrain_goat.py:486-499 in one launch per direction."""

    @staticmethod
    def forward(ctx, go, vo, fwl):
        go, vo = go.float().contiguous(), vo.float().contiguous()
        f1 = fwl.reshape(-1).contiguous()
        fo, fw = _cfp_mix_fwd(go, vo, f1)
        ctx.save_for_backward(go, vo, fw)
        ctx.fshape, ctx.fdtype = fwl.shape, f1.dtype
        return fo

    @staticmethod
    def backward(ctx, dfo):
        go, vo, fw = ctx.saved_tensors
        dgo, dvo = torch.empty_like(go), torch.empty_like(vo)
        dfwl = _cfp_mix_bwd(go, vo, fw, dfo.float().contiguous(), dgo, dvo, ctx.fdtype, False)
        return dgo, dvo, dfwl.view(ctx.fshape)


_MEAN_SEED = {}


def backward_mean(loss_vec, scale=1.0):
    """`(loss_vec.mean() * scale).backward()` as the trainer spells it (P/train_r2r_goat.py:333-338), minus the autograd engine's seed
    launches: the mean is still computed (one launch: trainers log it; returned detached), the backward pass starts from a cached
    constant vector scale / n instead of ones_like + div (+ mul) kernels."""
    n = loss_vec.numel()
    key = (n, float(scale), loss_vec.device, loss_vec.dtype)
    g = _MEAN_SEED.get(key)
    if g is None:
        g = _MEAN_SEED[key] = torch.full((n,), float(scale) / n, dtype=loss_vec.dtype, device=loss_vec.device)
    m = loss_vec.detach().mean()
    torch.autograd.backward(loss_vec, grad_tensors=g.view_as(loss_vec))
    return m


def cfp_mix(go, vo, fwl):
    return _CfpMixFn.apply(go, vo, fwl)


class _CfpTailFn(torch.autograd.Function):
    """loss = InfoNCE(go, vo, fo, to) with fo = mix(go, vo, fwl) — the whole CFP tail behind the three pooled vectors as ONE autograd node
    (single rank: candidates = the batch's own vectors): two launches per direction, and the gradients of go / vo from the loss and from
    the fused vector meet inside the backward kernels instead of in autograd-engine adds."""

    @staticmethod
    def forward(ctx, go, vo, fwl, to, temperature):
        go, vo, to = go.float().contiguous(), vo.float().contiguous(), to.float().contiguous()
        f1 = fwl.reshape(-1).contiguous()
        fo, fw = _cfp_mix_fwd(go, vo, f1)
        Bl, H = go.shape
        loss = torch.zeros(Bl, dtype=torch.float32, device=go.device)
        prob = torch.empty((6, Bl, Bl), dtype=torch.float32, device=go.device)
        xs = (ctypes.c_void_p * 3)(_ptr(go), _ptr(vo), _ptr(fo))
        st = _lib.lib().goat_infonce_fwd(_stream(), xs, xs, _ptr(to), _ptr(to), _ptr(loss), _ptr(prob), Bl, Bl, H, 0, float(temperature))
        _lib.check(st, 'goat_infonce_fwd')
        ctx.save_for_backward(go, vo, fo, to, fw, prob)
        ctx.fshape, ctx.fdtype, ctx.temperature = fwl.shape, f1.dtype, float(temperature)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        go, vo, fo, to, fw, prob = ctx.saved_tensors
        Bl, H = go.shape
        d = torch.zeros((4, Bl, H), dtype=torch.float32, device=go.device)          # dgo | dvo | dfo | dto, accumulated by the kernels
        xs = (ctypes.c_void_p * 3)(_ptr(go), _ptr(vo), _ptr(fo))
        dx = (ctypes.c_void_p * 3)(_ptr(d[0]), _ptr(d[1]), _ptr(d[2]))
        st = _lib.lib().goat_infonce_bwd(_stream(), xs, xs, _ptr(to), _ptr(to), _ptr(dloss.float().contiguous()), _ptr(prob), dx, dx, _ptr(d[3]), _ptr(d[3]),
                                         Bl, Bl, H, 0, ctx.temperature)
        _lib.check(st, 'goat_infonce_bwd')
        dfwl = _cfp_mix_bwd(go, vo, fw, d[2], d[0], d[1], ctx.fdtype, True)
        return d[0], d[1], dfwl.view(ctx.fshape), d[3], None


def cfp_tail(go, vo, fwl, to, temperature):
    return _CfpTailFn.apply(go, vo, fwl, to, temperature)


def infonce(g_loc, v_loc, f_loc, t_loc, g_all, v_all, f_all, t_all, target0, temperature):
    return _InfoNceFn.apply(g_loc, v_loc, f_loc, t_loc, g_all, v_all, f_all, t_all, target0, temperature)


# ----------------------------------------------------------------------------- aliases of the state that moved to tuning.py
class _HipopsModule(type(os)):
    """`hipops.AUTOTUNE` / `hipops.PROFILE` read and write tuning.AUTOTUNE / tuning.PROFILE (bench.py, scripts and trainers written against
    the round-5 layout switch them through this module)."""
    _FORWARD = ('AUTOTUNE', 'PROFILE')

    def __getattr__(self, name):
        if name in _HipopsModule._FORWARD:
            return getattr(tuning, name)
        raise AttributeError('module %r has no attribute %r' % (self.__name__, name))

    def __setattr__(self, name, value):
        if name in _HipopsModule._FORWARD:
            setattr(tuning, name, value)
        else:
            super().__setattr__(name, value)


import sys as _sys            # noqa: E402
_sys.modules[__name__].__class__ = _HipopsModule
