"""Fused optimizer step on the gradient arena: the update either side of the forward/backward path.

Reference behaviour (P = /root/reference/pretrain_src):
  * P/optim/misc.py:13-37    build_optimizer: two parameter groups by NAME — weight decay for everything except names
                             containing 'bias', 'LayerNorm.bias', 'LayerNorm.weight' — lr / betas from the options
  * P/optim/adamw.py:53-110  HF-style AdamW: bias-corrected step size from the PARAMETER's own step count, decoupled decay
                             applied after the update, parameters whose .grad is None skipped (no moment update, no decay)
  * P/train_r2r_goat.py:349-366   per update: lr of the schedule into every group, clip_grad_norm_(5.0), optimizer.step()
Here the gradients already sit in ONE float32 buffer (dp.GradArena); the moments get two buffers of the same layout and the
whole update is two kernels (goat_grad_sqnorm, goat_adamw_step in csrc/optim.hip) that also refresh the bf16 operand
"shadows" of the weights, so the next forward needs no cast kernels.  No host synchronisation: the clip coefficient is
computed on the device; `last_grad_norm()` reads it back on request.
"""
import ctypes

import torch

from . import _lib

NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')      # P/optim/misc.py:13
CHUNK = 65536


class FusedAdamW:
    """optimizer over the parameters of a dp.GradArena.

        opt = FusedAdamW(model.named_parameters(), arena, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01)
        ... backward of `task` (gradients in the arena) ...
        opt.step(task, max_norm=5.0)

    `param_groups` mirrors the reference's two groups ({'lr', 'weight_decay', 'names'}) so that a trainer's
    `for g in optimizer.param_groups: g['lr'] = lr_this_step` keeps working."""

    def __init__(self, named_params, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0:
            raise ValueError('invalid AdamW hyper-parameters')                       # P/optim/adamw.py:44-51
        self.arena = arena
        self.betas, self.eps, self.correct_bias = (float(betas[0]), float(betas[1])), float(eps), bool(correct_bias)
        named = [(n, p) for n, p in named_params if id(p) in arena.views]
        self.names = {id(p): n for n, p in named}
        decay = [n for n, _ in named if not any(nd in n for nd in NO_DECAY)]
        no_decay = [n for n, _ in named if any(nd in n for nd in NO_DECAY)]
        self.param_groups = [{'lr': float(lr), 'weight_decay': float(weight_decay), 'names': decay},
                             {'lr': float(lr), 'weight_decay': 0.0, 'names': no_decay}]
        self._group_of = {n: 0 for n in decay}
        self._group_of.update({n: 1 for n in no_decay})
        dev = arena.flat.device
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.steps = {id(p): 0 for _, p in named}                                   # state['step'] of every parameter
        self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._plans = {}

    # -- per-task plan (static): which tensors, their chunks, the arena ranges of the norm --------------------------------------
    def _plan(self, task):
        key = task.split('_')[0] if task is not None else None
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        arena = self.arena
        plist = [p for p in arena.params if id(p) in self.names and (arena.tasks_of[id(p)] is None or key is None or key in arena.tasks_of[id(p)])]
        chunks = []
        for i, p in enumerate(plist):
            for first in range(0, p.numel(), CHUNK):
                chunks += [i, first]
        dev = arena.flat.device
        # the norm runs over the parameters' own elements (the alignment padding between arena slices is never written: zeros)
        rng = []
        for p in plist:
            a = arena.offsets[id(p)]
            rng += [a, a + p.numel()]
        nbytes = max(1, len(plist)) * ctypes.sizeof(_lib.AdamwTensor)
        pl = {'params': plist, 'chunks': torch.tensor(chunks, dtype=torch.int32, device=dev), 'nchunks': len(chunks) // 2,
              'ranges': torch.tensor(rng, dtype=torch.int64, device=dev), 'n_ranges': len(rng) // 2,
              'host': (_lib.AdamwTensor * max(1, len(plist)))(), 'dev': torch.empty(nbytes, dtype=torch.uint8, device=dev),
              'pinned': torch.empty(nbytes, dtype=torch.uint8).pin_memory() if dev.type == 'cuda' else None}
        self._plans[key] = pl
        return pl

    @staticmethod
    def _shadows(p):
        """bf16 copies of `p` the kernel can refresh in place (same element order): the plain shadow and the row-padded
        decoder shadow.  Every other cached single-parameter shadow (transposed, K-padded, float32) is dropped so that its
        next use rebuilds it from the updated master."""
        out = []
        cache = p.__dict__.get('_goat_shadow')
        if cache:
            for k in list(cache):
                if k[0] in ('cat', 'catb'):
                    continue
                ver, t = cache[k]
                ok = False
                if t.dtype == torch.bfloat16 and t.is_contiguous():
                    if len(k) == 3 and k[0] == torch.bfloat16 and k[1] is False and k[2] == 0:
                        ok = True
                    elif k[0] == 'rowpad' and t.shape[1:] == p.shape[1:] and t.shape[0] >= p.shape[0]:
                        ok = True
                if ok and len(out) < 2:
                    out.append(t.data_ptr())
                else:
                    del cache[k]
        return out

    def _cat_members(self, plist):
        """{id(param): [pointer into a row-concatenated bf16 shadow]} for the members of cached 'cat' shadows (fused QKV / KV
        projections: the rows of parameter i start at sum of the rows before it)."""
        by_id = {id(p): p for p in plist}
        out = {}
        for p in plist:
            cache = p.__dict__.get('_goat_shadow')
            if not cache:
                continue
            for k in list(cache):
                if k[0] == 'catb':             # concatenated float32 biases: rebuilt on next use (tiny)
                    del cache[k]
                    continue
                if k[0] != 'cat':
                    continue
                ver, t = cache[k]
                ids = k[3]
                if k[1] != torch.bfloat16 or k[2] or not t.is_contiguous() or any(i not in by_id for i in ids):
                    del cache[k]              # cannot be refreshed in place: rebuilt on next use
                    continue
                row = 0
                for i in ids:
                    out.setdefault(i, []).append(t.data_ptr() + row * t.shape[1] * 2)
                    row += by_id[i].shape[0]
        return out

    def step(self, task=None, max_norm=5.0):
        """clip_grad_norm_(max_norm) + AdamW on the parameters `task` uses (all arena parameters if None)."""
        pl = self._plan(task)
        if not pl['params']:
            return
        arena, lib = self.arena, _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        b1, b2 = self.betas
        cat = self._cat_members(pl['params'])
        host = pl['host']
        for i, p in enumerate(pl['params']):
            self.steps[id(p)] += 1
            t = self.steps[id(p)]
            g = self.param_groups[self._group_of[self.names[id(p)]]]
            lr = g['lr']
            step_size = lr * ((1.0 - b2 ** t) ** 0.5) / (1.0 - b1 ** t) if self.correct_bias else lr
            sh = self._shadows(p) + cat.get(id(p), [])
            if len(sh) > 2:                    # more bf16 copies than the kernel refreshes: drop the caches, rebuilt lazily
                p.__dict__.pop('_goat_shadow', None)
                sh = []
            e = host[i]
            e.param, e.arena_off, e.numel = p.data_ptr(), arena.offsets[id(p)], p.numel()
            e.shadow0 = sh[0] if len(sh) > 0 else None
            e.shadow1 = sh[1] if len(sh) > 1 else None
            e.step_size, e.decay = step_size, lr * g['weight_decay']
        nbytes = ctypes.sizeof(host)
        raw = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ctypes.addressof(host)), dtype=torch.uint8)
        if pl['pinned'] is not None:
            pl['pinned'].copy_(raw)
            pl['dev'].copy_(pl['pinned'], non_blocking=True)
        else:
            pl['dev'].copy_(raw)
        self._sq.zero_()
        clip = max_norm is not None and max_norm > 0
        _lib.check(lib.goat_grad_sqnorm(st, arena.flat.data_ptr(), pl['ranges'].data_ptr(), pl['n_ranges'], self._sq.data_ptr()), 'goat_grad_sqnorm')
        _lib.check(lib.goat_adamw_step(st, arena.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), pl['dev'].data_ptr(),
                                       pl['chunks'].data_ptr(), pl['nchunks'], b1, b2, self.eps, float(max_norm) if clip else 0.0,
                                       self._sq.data_ptr()), 'goat_adamw_step')

    def last_grad_norm(self):
        """total gradient norm of the last step() (synchronises)."""
        return float(self._sq.sqrt().item())

    def state_of(self, p):
        a = self.arena.offsets[id(p)]
        return {'step': self.steps[id(p)], 'exp_avg': self.exp_avg[a:a + p.numel()].view_as(p), 'exp_avg_sq': self.exp_avg_sq[a:a + p.numel()].view_as(p)}


def build_optimizer(model, opts, arena):
    """P/optim/misc.py:11-37 on the arena: opts.optim must be 'adamw' (the shipped configuration); learning_rate, betas,
    weight_decay as there."""
    if getattr(opts, 'optim', 'adamw') != 'adamw':
        raise ValueError('invalid optimizer')
    return FusedAdamW(model.named_parameters(), arena, lr=opts.learning_rate, betas=tuple(opts.betas), weight_decay=opts.weight_decay)
