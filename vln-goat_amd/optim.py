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
        self._by_id = {id(p): p for _, p in named}
        self.last_refreshed = 0          # copies rebuilt by torch ops after the last step (0 in the steady bf16 state)

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

    def _copies(self, plist):
        """Which cached operand copies the update kernel refreshes itself.  -> ({id(p): slots}, done) with slots =
        {'s0', 's1': bf16 pointers in the parameter's element order, 'cols', 'ld0': s0 is a row-padded image, 'f32': float32
        pointer} and done = the storage addresses of the copies that are COMPLETELY covered by the slots handed out (a
        concatenated copy only when each of its members got one).  Everything else is rebuilt in place after the kernel by
        hipops.refresh_shadows — a cached copy is never dropped: a captured hipGraph may read it at its address."""
        by_id = {id(p): p for p in plist}
        slots = {id(p): {'bf16': [], 'pad': None, 'f32': None} for p in plist}
        whole = []            # (tensor, [(param id, kind, pointer)]) per cached copy
        for p in plist:
            cache = p.__dict__.get('_goat_shadow')
            if not cache:
                continue
            for k, (ver, t) in cache.items():
                if not t.is_contiguous():
                    continue
                if k[0] == 'cat':
                    if k[1] != torch.bfloat16 or k[2] or any(i not in by_id for i in k[3]):
                        continue
                    row, members = 0, []
                    for i in k[3]:
                        members.append((i, 'bf16', t.data_ptr() + row * t.shape[1] * 2))
                        row += by_id[i].shape[0]
                    whole.append((t, members))
                elif k[0] == 'catb':
                    if t.dtype != torch.float32 or any(i not in by_id for i in k[1]):
                        continue
                    off, members = 0, []
                    for i in k[1]:
                        members.append((i, 'f32', t.data_ptr() + off * 4))
                        off += by_id[i].numel()
                    whole.append((t, members))
                elif k[0] == 'rowpad':
                    if t.dtype == torch.bfloat16 and t.shape[1:] == p.shape[1:] and t.shape[0] >= p.shape[0]:
                        whole.append((t, [(id(p), 'bf16', t.data_ptr())]))
                elif k[0] == 'bpad':          # zero-padded float32 image of a bias: the leading numel(p) floats are the bias
                    if t.dtype == torch.float32 and t.numel() >= p.numel():
                        whole.append((t, [(id(p), 'f32', t.data_ptr())]))
                elif k[0] == torch.bfloat16 and k[1] is False and t.dtype == torch.bfloat16:
                    if k[2] == 0:
                        whole.append((t, [(id(p), 'bf16', t.data_ptr())]))
                    elif p.dim() == 2:
                        whole.append((t, [(id(p), 'pad', (t.data_ptr(), p.shape[1], p.shape[1] + k[2]))]))
        done = set()
        for t, members in whole:
            ok = True
            for i, kind, _ in members:        # does every member still have a free slot of that kind?
                sl = slots[i]
                if kind == 'bf16':
                    ok &= len(sl['bf16']) < (1 if sl['pad'] is not None else 2)
                elif kind == 'pad':
                    ok &= sl['pad'] is None and len(sl['bf16']) < 2
                else:
                    ok &= sl['f32'] is None
            if not ok:
                continue
            for i, kind, ptr in members:
                if kind == 'bf16':
                    slots[i]['bf16'].append(ptr)
                else:
                    slots[i][kind] = ptr
            done.add(t.data_ptr())
        return slots, done

    def step(self, task=None, max_norm=5.0):
        """clip_grad_norm_(max_norm) + AdamW on the parameters `task` uses (all arena parameters if None)."""
        from . import hipops
        pl = self._plan(task)
        if not pl['params']:
            return
        arena, lib = self.arena, _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        b1, b2 = self.betas
        slots, done = self._copies(pl['params'])
        host = pl['host']
        for i, p in enumerate(pl['params']):
            self.steps[id(p)] += 1
            t = self.steps[id(p)]
            g = self.param_groups[self._group_of[self.names[id(p)]]]
            lr = g['lr']
            step_size = lr * ((1.0 - b2 ** t) ** 0.5) / (1.0 - b1 ** t) if self.correct_bias else lr
            sl = slots[id(p)]
            e = host[i]
            e.param, e.arena_off, e.numel = p.data_ptr(), arena.offsets[id(p)], p.numel()
            bf = list(sl['bf16'])
            if sl['pad'] is not None:
                e.shadow0, e.cols, e.ld0 = sl['pad']
                e.shadow1 = bf[0] if bf else None
            else:
                e.cols, e.ld0 = 0, 0
                e.shadow0 = bf[0] if len(bf) > 0 else None
                e.shadow1 = bf[1] if len(bf) > 1 else None
            e.shadow_f32 = sl['f32']
            e.step_size, e.decay = step_size, lr * g['weight_decay']
        nbytes = ctypes.sizeof(host)
        raw = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ctypes.addressof(host)), dtype=torch.uint8)
        if pl['pinned'] is not None:
            if pl.get('copied') is not None:
                pl['copied'].synchronize()       # the previous step's H2D copy has read the pinned table (graph-replay loops never sync)
            pl['pinned'].copy_(raw)
            pl['dev'].copy_(pl['pinned'], non_blocking=True)
            pl['copied'] = torch.cuda.Event()
            pl['copied'].record()
        else:
            pl['dev'].copy_(raw)
        self._sq.zero_()
        clip = max_norm is not None and max_norm > 0
        _lib.check(lib.goat_grad_sqnorm(st, arena.flat.data_ptr(), pl['ranges'].data_ptr(), pl['n_ranges'], self._sq.data_ptr()), 'goat_grad_sqnorm')
        _lib.check(lib.goat_adamw_step(st, arena.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), pl['dev'].data_ptr(),
                                       pl['chunks'].data_ptr(), pl['nchunks'], b1, b2, self.eps, float(max_norm) if clip else 0.0,
                                       self._sq.data_ptr()), 'goat_adamw_step')
        # the copies the kernel did not cover (transposed / float32 images of the parity mode, a third bf16 copy): same storage, new values
        by_id = self._by_id
        self.last_refreshed = sum(hipops.refresh_shadows(p, by_id, done) for p in pl['params'])

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
