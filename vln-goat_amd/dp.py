"""Data-parallel engine for GOAT pre-training: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The reference wraps the model in torch DDP with find_unused_parameters=True (P/utils/misc.py:52-65) and
relies on its implicit bucketed all-reduce; tasks touch different parameter subsets.
This engine keeps the semantics (gradients averaged over ranks, every rank runs the same task) but owns
the communication:
  * a static per-task participation list (parameters that receive a gradient for that task, discovered
    on the task's first step — identical on all ranks because the task is), replacing the per-iteration
    unused-parameter bitmap all-reduce;
  * GradArena: `.grad` of every used parameter is a view into ONE float32 HBM buffer ordered by task usage; the HIP
    backward kernels accumulate into it directly (hipops._sink), weight gradients are deferred and run as grouped
    launches (hipops.WgradQueue), and the all-reduce (mean) runs in place on the task's 1-3 contiguous ranges in
    128 MB chunks on a dedicated communication stream — no pack/unpack copies, ReduceOp.AVG inside RCCL;
  * GradBuckets (legacy path, used when no arena is attached): gradients packed bucket by bucket (reverse registration
    order ~ reverse autograd order) into flat buffers and all-reduced on the communication stream, so the pack of
    bucket i+1 overlaps the all-reduce of bucket i; optional bf16 wire format;
  * CfpGather: the CFP contrastive negatives are shared across ranks with ONE all-gather of the packed
    [4,B,H] pooled vectors; its backward is the matching reduce-scatter (sum) so the result equals the
    single-process loss on the concatenated batch (the reference computes CFP per rank only,
    P/model/pretrain_goat.py:519-538 — this is a build-side extension, exact at world_size 1).
Everything is device-agnostic (works with gloo on CPU tensors), which is how it is unit-tested here.
"""
import torch
import torch.distributed as dist


# Run the collectives even when the process group has ONE rank (they are identities there): the only way to exercise the RCCL
# launch / hipGraph-capture path of the gradient exchange on a single-GPU box (RCCL refuses two ranks per device).  Set by the
# in-graph communication tests and by `bench.py --in-graph-comm` at N = 1; never needed at N > 1.
FORCE_COLLECTIVES = [False]


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _comm_on():
    """is there a gradient exchange to run?  (more than one rank, or one rank with FORCE_COLLECTIVES)"""
    return _world() > 1 or (FORCE_COLLECTIVES[0] and dist.is_available() and dist.is_initialized())


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _gloo_gather_rows(x):
    """[W, n] = the vector x of every rank (gloo, tests only): all_gather on CPU tensors, a zero-padded float32 all-reduce on GPU tensors
    (gloo has no GPU all-gather; adding zeros is exact, so the values are the senders' bits)."""
    W = _world()
    if x.is_cuda:
        allv = torch.zeros((W, x.numel()), dtype=torch.float32, device=x.device)
        allv[_rank()] = x.float()
        dist.all_reduce(allv)
        return allv.to(x.dtype)
    parts = [torch.empty_like(x) for _ in range(W)]
    dist.all_gather(parts, x)
    return torch.stack(parts, 0)


class _AllGatherCat(torch.autograd.Function):
    """y = cat_r(x_r) over ranks along dim 1; backward: dx_r = sum over ranks of dy[:, r-th slice]."""

    @staticmethod
    def forward(ctx, x):
        W = _world()
        ctx.shape = x.shape
        if dist.get_backend() == 'gloo' and x.is_cuda:          # gloo has no GPU all-gather: zero-padded all-reduce
            out = torch.zeros((W,) + tuple(x.shape), dtype=x.dtype, device=x.device)
            out[_rank()] = x
            dist.all_reduce(out)
            return out
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _note_collective(x)
        dist.all_gather_into_tensor(out, x.contiguous())       # concatenated along dim 0
        return out.view((W,) + tuple(x.shape))

    @staticmethod
    def backward(ctx, dy):
        W = _world()
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        dy = dy.contiguous()
        if dist.get_backend() == 'gloo':   # gloo has no reduce_scatter_tensor
            dist.all_reduce(dy)
            dx.copy_(dy[_rank()])
        else:
            _note_collective(dy)
            dist.reduce_scatter_tensor(dx, dy.view((W * ctx.shape[0],) + tuple(ctx.shape[1:])))
        return dx


class CfpGather:
    """Callable plugged into GlocalTextPathCMTPreTraining.cfp_gather."""

    def __call__(self, gmap_o, vp_o, fused_o, txt_o):
        W = _world()
        if W == 1:
            return gmap_o, vp_o, fused_o, txt_o, 0
        B = gmap_o.shape[0]
        packed = torch.stack([gmap_o, vp_o, fused_o, txt_o], 0)         # [4,B,H]
        allv = _AllGatherCat.apply(packed)                               # [W,4,B,H]
        allv = allv.permute(1, 0, 2, 3).reshape(4, W * B, -1)
        return allv[0], allv[1], allv[2], allv[3], _rank() * B


class GradBuckets:
    """Flat all-reduce buckets over a fixed list of parameters (one instance per task)."""

    def __init__(self, params, bucket_bytes=64 << 20, wire_dtype=None):
        self.params = list(params)
        self.wire_dtype = wire_dtype
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self.flat = None
        self.comm_stream = None

    def _alloc(self, device):
        dt = self.wire_dtype or torch.float32
        self.flat = [torch.empty(sum(p.numel() for p in b), dtype=dt, device=device) for b in self.buckets]
        if device.type == 'cuda':
            self.comm_stream = torch.cuda.Stream(device=device)

    def all_reduce_mean(self, grads=None):
        """Average `p.grad` (or the given per-parameter tensors) over ranks, in place."""
        W = _world()
        if W == 1:
            return
        glist = grads if grads is not None else [p.grad for p in self.params]
        gmap = {id(p): g for p, g in zip(self.params, glist)}
        dev = glist[0].device
        if self.flat is None:
            self._alloc(dev)
        cuda = dev.type == 'cuda'
        works = []
        for flat, bucket in zip(self.flat, self.buckets):
            gs = [gmap[id(p)] for p in bucket]
            views, off = [], 0
            for g in gs:
                views.append(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            torch._foreach_copy_(views, gs)                      # pack (casts to the wire dtype)
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    dist.all_reduce(flat)
                    flat.div_(W)
            else:
                dist.all_reduce(flat)
                flat.div_(W)
            works.append((views, gs))
        if cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for views, gs in works:
            torch._foreach_copy_(gs, views)                      # unpack reduced values back into the grads


class GradArena:
    """Flat float32 gradient storage in HBM: every parameter's `.grad` is a view into ONE buffer.

    * the HIP autograd Functions (hipops._sink) accumulate weight / bias / LayerNorm / embedding-table gradients
      straight into these views — no per-parameter zero-fill, no temporaries, no `grad += dW` kernels; anything that
      still comes back through autograd is added in place by AccumulateGrad, so the result is always `.grad` as
      torch would have produced it;
    * one fill per step zeroes the slices a task touches (`zero(task)`), and the gradient all-reduce runs in place
      on a few large contiguous ranges (`all_reduce_mean(task)`), without pack/unpack copies;
    * parameters are ordered by the set of tasks that use them (`usage`), so a task's parameters form a handful of
      contiguous ranges; within a group the large matrices come first, in registration order, so the query/key/value
      weights of a block stay adjacent (one fused wgrad GEMM writes all three), then the small parameters.
    Parameters that no task uses are left out (their .grad stays None, as under the reference's
    DDP(find_unused_parameters=True), P/utils/misc.py:52-65).
    Call `zero()` instead of `optimizer.zero_grad(set_to_none=True)`; if some code does set a .grad to None, the
    Functions notice (the sink is no longer bound) and fall back to returning ordinary gradients."""

    ALIGN = 64          # elements (256 B): slices stay 16-B aligned for the GEMM epilogue and vector fills
    SMALL = 100000      # parameters below this size (biases, LayerNorm, 1-column heads) sit together at the end of their
                        # group and are cleared by ONE fill per group in zero(); the large matrices are cleared / overwritten
                        # by their first writer (cache-warm for the split-K atomics)

    def __init__(self, params, usage=None, bucket_bytes=128 << 20, phase_of=None, wire_dtype=None):
        """wire_dtype: None — float32 on the wire, as the reference's DDP — or torch.bfloat16: half the bytes per step over xGMI
        (the 8-GPU target needs it), accumulated in float32 on receipt (_reduce_mean_wire).
        phase_of: optional {id(param): 0 | 1} — backward phase in which the parameter's gradient becomes final (0: the part of
        the backward pass that runs first, e.g. heads / cross-modal / panorama encoders; 1: the rest, e.g. text encoder and
        embeddings).  Phases are kept contiguous inside every usage group so that the phase-0 ranges can be all-reduced
        while the later phases are still computing (GoatDataParallel.backward_phase)."""
        params = [p for p in params if p.requires_grad]
        phase_of = phase_of or {}
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        if usage is not None:
            uniq = [p for p in uniq if usage.get(id(p))]
            key = lambda p: tuple(sorted(usage[id(p)]))
        else:
            key = lambda p: ()
        groups = {}
        for p in uniq:
            groups.setdefault(key(p), []).append(p)
        self.params, self.offsets, self.tasks_of, self.phase = [], {}, {}, {}
        self.bucket_elems = max(1, bucket_bytes // 4)
        off = 0
        for k in sorted(groups):
            plist = groups[k]
            ordered = []
            for ph in sorted({phase_of.get(id(q), 0) for q in plist}):
                sub = [q for q in plist if phase_of.get(id(q), 0) == ph]
                ordered += [q for q in sub if q.numel() >= self.SMALL] + [q for q in sub if q.numel() < self.SMALL]
            for p in ordered:
                if p.numel() < self.SMALL:
                    p.__dict__['_goat_prezero'] = True     # cleared by zero(): the kernels only ever accumulate into it
                else:
                    p.__dict__.pop('_goat_prezero', None)
                self.params.append(p)
                self.offsets[id(p)] = off
                self.tasks_of[id(p)] = frozenset(k) if usage is not None else None
                self.phase[id(p)] = phase_of.get(id(p), 0)
                off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        self.views = {id(p): self.flat[self.offsets[id(p)]:self.offsets[id(p)] + p.numel()].view_as(p) for p in self.params}
        self._ranges, self._owned, self._fills, self._cur = {}, {}, {}, None
        self.no_zero = {}       # task -> ids of parameters whose slice zero() leaves alone (written wholesale elsewhere)
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self.wire_dtype = wire_dtype
        self._wire = {}         # staging buffers of the reduced-precision exchange, by chunk length (fixed addresses: capturable)

    # -- binding ---------------------------------------------------------------------------------
    def attach(self):
        for p in self.params:
            v = self.views[id(p)]
            p.grad = v
            p.__dict__['_goat_sink'] = v
        return self

    def detach(self):
        for p in self.params:
            p.grad = None
            p.__dict__.pop('_goat_sink', None)

    def bind(self, task):
        """Call before the backward of a step when an optimizer consumes .grad: .grad = arena view (sink re-bound) for the
        parameters `task` uses, None for the others — what the reference's optimizer sees after zero_grad + a step
        of that task.  Replayed hipGraphs write into the arena regardless of this host-side binding."""
        key = task.split('_')[0]
        for p in self.params:
            t = self.tasks_of[id(p)]
            used = t is None or key in t
            p.grad = self.views[id(p)] if used else None
            if used:
                p.__dict__['_goat_sink'] = self.views[id(p)]

    # -- ranges ----------------------------------------------------------------------------------
    def ranges(self, task=None, phase=None, exclude=frozenset()):
        key = task.split('_')[0] if task is not None else None
        r = self._ranges.get((key, phase, exclude))
        if r is None:
            r = []
            for p in self.params:
                t = self.tasks_of[id(p)]
                if key is not None and t is not None and key not in t:
                    continue
                if (phase is not None and self.phase[id(p)] != phase) or id(p) in exclude:
                    continue
                a = self.offsets[id(p)]
                b = a + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                if r and r[-1][1] == a:
                    r[-1][1] = b
                else:
                    r.append([a, b])
            r = self._ranges[(key, phase, exclude)] = [tuple(x) for x in r]
        return r

    def zero(self, task=None):
        """Start of a step.  The HIP Functions clear (or overwrite) a slice themselves the first time they write it in a
        step — right before the kernel that accumulates into it, so the lines are still in the Infinity Cache — and
        this call only has to zero the slices that are NOT written that way (gradients that arrive through plain
        autograd).  Which slices the Functions own is learned per task from the previous step of that task (per-task
        data flow is static; the first step of a task clears everything)."""
        from . import hipops
        key = task.split('_')[0] if task is not None else None
        if self._cur is not None:                       # close the previous step: remember what its Functions wrote
            prev, epoch = self._cur
            self._owned[prev] = {id(p) for p in self.params
                                 if p.__dict__.get('_goat_epoch') == epoch and not p.__dict__.get('_goat_prezero')}
        hipops.WgradQueue.reset()
        hipops.ARENA_EPOCH[0] += 1
        self._cur = (key, hipops.ARENA_EPOCH[0])
        owned = self._owned.get(key)
        if owned is None:
            self._zero([self.flat[a:b] for a, b in self.ranges(task, None, frozenset(self.no_zero.get(key, ())))])
            return
        fills = self._fills.get(key)
        if fills is None or fills[0] != frozenset(owned):
            r = []
            for p in self.params:
                t = self.tasks_of[id(p)]
                if (key is not None and t is not None and key not in t) or id(p) in owned or id(p) in self.no_zero.get(key, ()):
                    continue
                a = self.offsets[id(p)]
                b = a + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                if r and r[-1][1] == a:
                    r[-1][1] = b
                else:
                    r.append([a, b])
            fills = self._fills[key] = (frozenset(owned), r)
        self._zero([self.flat[a:b] for a, b in fills[1]])

    def _zero(self, views):
        if views and views[0].is_cuda:
            from . import hipops
            hipops.zero_ranges(views)          # one launch for the step's fills (csrc/glue.hip)
        else:
            for v in views:
                v.zero_()

    # -- communication ---------------------------------------------------------------------------
    def close_step(self):
        """End of an eager backward pass: zero() skipped the slices the HIP Functions wrote last time (they clear / overwrite
        them on first touch).  If one of those Functions did NOT run this time (e.g. pretrain_model's early return when no
        token is masked), its slice still holds the previous step's gradient: clear it now, before anything reads .grad.
        Host-side check of ~300 dictionary entries; replayed hipGraphs have a static data flow and do not need it."""
        if self._cur is None:
            return 0
        key, epoch = self._cur
        owned = self._owned.get(key, ())
        n = 0
        for p in self.params:
            if id(p) in owned and p.__dict__.get('_goat_epoch') != epoch:
                self.views[id(p)].zero_()
                n += 1
        return n

    _avg_ok = None      # RCCL averages in the collective (ReduceOp.AVG); gloo and old stacks: sum, then one divide pass

    def _reduce_mean(self, c, W):
        _note_collective(c)
        cls = type(self)
        if cls._avg_ok is None:
            cls._avg_ok = False
            if c.is_cuda and dist.get_backend() == 'nccl':
                try:
                    probe = torch.ones(8, dtype=torch.float32, device=c.device)
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG)
                    cls._avg_ok = True
                except (RuntimeError, ValueError):
                    cls._avg_ok = False
        if cls._avg_ok:
            dist.all_reduce(c, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(c)
            c.div_(W)

    def _reduce_mean_wire(self, c, W):
        """Mean over ranks of the float32 chunk `c` with `wire_dtype` (bfloat16) on the wire and float32 accumulation:
            1. every rank rounds its chunk to the wire dtype and sends shard j of it to rank j      (all-to-all, (W-1)/W of the chunk)
            2. rank j adds the W shards it received in float32, divides by W, rounds the mean ONCE
            3. the W averaged shards are all-gathered and widened back into the arena            (all-gather, (W-1)/W of the chunk)
        Same bytes per rank as a ring all-reduce in the wire dtype, but one rounding of the inputs and one of the result instead
        of one per ring hop, every rank ends with bit-identical gradients (they all widen the same gathered shards), and on the
        xGMI full mesh both collectives use all links at once.  gloo (CPU tests) lacks all-to-all: all-gather + local sum, same
        arithmetic.  Inside a hipGraph capture step 1-2 is a reduce-scatter in the wire dtype instead (RCCL's all-to-all does not
        survive capture on this stack — scripts/rccl_capture_probe.py: it crashes hipStreamEndCapture; all-reduce, reduce-scatter,
        all-gather and broadcast capture fine): one rounding per ring hop rather than one per input, same bytes, same 2e-2 bound."""
        _note_collective(c)
        n = c.numel()
        per = (n + W - 1) // W
        buf = self._wire.get(n)
        if buf is None:
            wd = self.wire_dtype
            buf = self._wire[n] = (torch.zeros(W * per, dtype=wd, device=c.device), torch.empty(W * per, dtype=wd, device=c.device),
                                   torch.empty(per, dtype=wd, device=c.device), torch.empty(W * per, dtype=wd, device=c.device))
        send, recv, shard, out = buf
        send[:n].copy_(c)
        gloo = dist.get_backend() == 'gloo'
        if gloo:
            r = _rank()
            recv.copy_(_gloo_gather_rows(send)[:, r * per:(r + 1) * per].reshape(-1))
            shard.copy_(recv.view(W, per).float().sum(0).div_(W))
        elif c.is_cuda and torch.cuda.is_current_stream_capturing():
            dist.reduce_scatter_tensor(shard, send)
            if W > 1:
                shard.copy_(shard.float().div_(W))
        else:
            dist.all_to_all_single(recv, send)
            shard.copy_(recv.view(W, per).float().sum(0).div_(W))
        if gloo:
            out.copy_(_gloo_gather_rows(shard).reshape(-1))
        else:
            dist.all_gather_into_tensor(out, shard)
        c.copy_(out[:n])

    def all_reduce_mean(self, task=None, phase=None, wait=True, exclude=frozenset(), extra=()):
        """Average the task's gradient ranges (of one backward phase, or all) over ranks, in place, on the communication
        stream.  wait=False: return without making the caller's stream wait (call wait_comm() before the gradients are
        read) — this is how the phase-0 all-reduce overlaps the phase-1 backward computation.  Capturable: inside a
        hipops.graph() the communication stream joins the capture as a parallel branch (wait=False branches must be joined
        by wait_comm() before the capture ends)."""
        W = _world()
        if not _comm_on():
            return
        reduce = self._reduce_mean if self.wire_dtype is None else self._reduce_mean_wire
        chunks = []
        for a, b in tuple(self.ranges(task, phase, exclude)) + tuple(extra):        # extra: explicit (begin, end) element ranges
            while a < b:
                e = min(b, a + self.bucket_elems)
                chunks.append(self.flat[a:e])
                a = e
        if self.comm_stream is not None:
            from . import hipops
            hipops.note_parallel_branch()          # (a capture in progress gets a parallel branch: see hipops' hipGraph lifetime note)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for c in chunks:
                    reduce(c, W)
            if wait:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for c in chunks:
                reduce(c, W)

    def wait_comm(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class GoatDataParallel(torch.nn.Module):
    """model wrapper: forward = model.forward; `reduce_gradients(task)` averages grads over ranks."""

    def __init__(self, model, bucket_bytes=64 << 20, wire_dtype=None, share_cfp_negatives=True):
        super().__init__()
        self.module = model
        self.bucket_bytes, self.wire_dtype = bucket_bytes, wire_dtype
        self._buckets = {}
        self._usage = {}
        self.arena = None
        if share_cfp_negatives and hasattr(model, 'cfp_gather'):
            model.cfp_gather = CfpGather()
        if _world() > 1:
            from . import hipops
            hipops.RngState.rank_salt = _rank()      # every rank draws its own dropout masks (the reference seeds seed + rank,
            counter = hipops.RngState.counter        # P/utils/misc.py:12-16 via train_r2r_goat.py:72)
            hipops.manual_seed(hipops.RngState.base)
            hipops.RngState.counter = counter
            # DDP constructor semantics: rank-0 parameters/buffers broadcast to all (P/utils/misc.py:58)
            for t in list(model.parameters()) + list(model.buffers()):
                _note_collective(t.data)
                dist.broadcast(t.data, 0)

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def record_usage(self, task):
        """Call after an ordinary (arena-less) backward of `task`: remembers which parameters it produced gradients for."""
        key = task.split('_')[0]
        for p in self.module.parameters():
            if p.grad is not None:
                self._usage.setdefault(id(p), set()).add(key)

    def build_arena(self, bucket_bytes=128 << 20, late_prefixes=(), phase_prefixes=None):
        """Flat gradient arena over the parameters seen by record_usage (all parameters if it was never called).
        phase_prefixes: [prefixes of phase 1, prefixes of phase 2, ...] — parameter-name prefixes grouped by the backward
        phase in which their gradients become final (phase 0 = everything else = the part of the backward pass that runs
        first).  late_prefixes=(...) is shorthand for one late phase.  See backward_phase()."""
        if phase_prefixes is None:
            phase_prefixes = [tuple(late_prefixes)] if late_prefixes else []

        def phase(name):
            for k, pres in enumerate(phase_prefixes):
                if any(name.startswith(pre) for pre in pres):
                    return k + 1
            return 0
        phase_of = {id(p): phase(n) for n, p in self.module.named_parameters()}
        self.n_phases = len(phase_prefixes) + 1
        self._phase_params = [[p for p in self.module.parameters() if phase_of[id(p)] == k and p.requires_grad]
                              for k in range(self.n_phases)]
        self.arena = GradArena(self.module.parameters(), self._usage or None, bucket_bytes, phase_of, wire_dtype=self.wire_dtype).attach()
        return self.arena

    # Phased backward: the gradient all-reduce of the parameters that finish first runs on the communication stream while the
    # rest of the backward pass is still computing.  The model's forward is cut at `boundaries`: identity views (t.view_as(t),
    # substituted for t in the forward pass by a forward hook) of the outputs of the sub-networks that belong to later phases
    # — autograd executes the grad_fn of a tensor listed in `inputs=`, which must therefore be a side-effect-free node.
    #     w.backward_phase(0, [loss], None, [txt, pano]);          w.reduce_gradients(task, phase=0, wait=False)
    #     w.backward_phase(1, [txt, pano], 'grad', [mid]);         w.reduce_gradients(task, phase=1, wait=False)
    #     w.backward_phase(2, [mid], 'grad', []);                  w.reduce_gradients(task, phase=2)      # waits for all
    def backward_phase(self, k, roots, grads, boundaries):
        """Back-propagate from `roots` (with `grads`: None for a scalar loss, 'grad' = each root's .grad filled by the previous
        phase, or explicit tensors) into the phase-k parameters and the .grad of `boundaries`."""
        from . import hipops
        if grads == 'grad':
            pairs = [(r, r.grad) for r in roots if r is not None and r.grad is not None]
            roots, grads = [r for r, _ in pairs], [g for _, g in pairs]
        boundaries = [b for b in boundaries if b is not None]
        for b in boundaries:
            b.grad = None
        # retain_graph (all but the last phase): the engine would otherwise release saved tensors the later phases need
        torch.autograd.backward(roots, grad_tensors=grads, inputs=self._phase_params[k] + boundaries,
                                retain_graph=k + 1 < self.n_phases)
        hipops.WgradQueue.flush()
        hipops.Branch.join_all()

    def backward_phase_a(self, loss, boundary, grad_tensors=None):
        """two-phase shorthand: loss -> boundary + phase-0 parameters."""
        self.backward_phase(0, [loss], None if grad_tensors is None else [grad_tensors], [boundary])

    def backward_phase_b(self, boundary):
        self.backward_phase(1, [boundary], 'grad', [])

    # -- sparse exchange of the word-embedding gradient -----------------------------------------------------------
    def enable_sparse_embedding(self, table, tasks, mixed_tasks=(), dense_phase=0):
        """`table`: the word-embedding Parameter; `tasks`: the tasks in which its gradient comes from lookups only
        (not mlm: the tied decoder makes it dense).  `mixed_tasks` (phased backward only): tasks in which the table gets a
        DENSE contribution early (mlm: the tied decoder's weight gradient, produced in backward phase `dense_phase`) and the
        lookup contribution in its own, last phase.  There the dense part is all-reduced with phase `dense_phase` (overlapping
        the rest of the backward pass) and the lookup part is exchanged sparsely and added afterwards, instead of one
        154 MB all-reduce after the last phase with nothing left to overlap it.
        Call after build_arena(); then begin_step(task) before each forward."""
        from . import hipops
        if _world() == 1 or self.arena is None or id(table) not in self.arena.views:
            return
        hipops.SparseEmbedGrad.params.add(id(table))
        mixed = {t.split('_')[0] for t in mixed_tasks} if self.arena.phase[id(table)] != dense_phase else set()
        self._sparse = (table, {t.split('_')[0] for t in tasks}, mixed, dense_phase)
        for t in self._sparse[1]:
            self.arena.no_zero.setdefault(t, set()).add(id(table))
        self._stash = {}

    def begin_step(self, task):
        from . import hipops
        key = task.split('_')[0]
        sp = getattr(self, '_sparse', None)
        if sp is not None and (key in sp[1] or key in sp[2]):
            lst = self._stash[key] = []
            hipops.SparseEmbedGrad.sink_list = lst
        else:
            hipops.SparseEmbedGrad.sink_list = None

    def _reduce_sparse(self, key, add=False):
        """all-gather (rows, ids) of every rank and rebuild the averaged table gradient locally (add=True: on top of the
        already averaged dense part)."""
        from . import hipops
        table = self._sparse[0]
        view = self.arena.views[id(table)]
        W = _world()
        if not add:
            view.zero_()
        for rows, ids, _tab, pad in self._stash.get(key, ()):
            ids = ids.reshape(-1).contiguous()
            # The collate pads txt_ids to the per-batch maximum length, so B*L differs between ranks in real training: the ranks
            # first agree on the largest row count and pad with zero rows (id 0: a scatter-add of zeros changes nothing).
            # sparse_uniform_rows = True (fixed-shape synthetic batches, bench.py) skips the exchange and its host sync.
            if not getattr(self, 'sparse_uniform_rows', False):
                cnt = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
                _note_collective(cnt)
                dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
                nmax = int(cnt.item())
                if rows.shape[0] < nmax:
                    fill = nmax - rows.shape[0]
                    rows = torch.cat([rows, rows.new_zeros((fill, rows.shape[1]))], 0)
                    ids = torch.cat([ids, ids.new_zeros(fill)], 0)
            if dist.get_backend() == 'gloo' and rows.is_cuda:        # (single-GPU self-tests: gloo has no GPU all-gather)
                all_rows = torch.zeros((W,) + tuple(rows.shape), dtype=torch.float32, device=rows.device)
                all_ids = torch.zeros((W,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device)
                all_rows[_rank()] = rows.float()
                all_ids[_rank()] = ids
                dist.all_reduce(all_rows)
                dist.all_reduce(all_ids)
                all_rows = all_rows.to(rows.dtype)
            else:
                all_rows = torch.empty((W * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
                all_ids = torch.empty(W * ids.shape[0], dtype=ids.dtype, device=ids.device)
                _note_collective(rows)
                dist.all_gather_into_tensor(all_rows, rows.contiguous())
                dist.all_gather_into_tensor(all_ids, ids)
            all_rows = all_rows.reshape(-1, rows.shape[1])
            hipops.embedding_scatter_add(view, all_rows * (1.0 / W), all_ids.reshape(-1), pad)

    def reduce_gradients(self, task, grads=None, phase=None, wait=True):
        if self.arena is not None and grads is None:
            key = task.split('_')[0]
            sp = getattr(self, '_sparse', None)
            if sp is not None and key in sp[1] and _world() > 1:
                tid = id(sp[0])
                if phase is None or phase == self.arena.phase[tid]:
                    self._reduce_sparse(key)            # small all-gather + local scatter on the caller's stream
                return self.arena.all_reduce_mean(task, phase, wait, frozenset((tid,)))
            if sp is not None and key in sp[2] and _world() > 1:
                if phase is None:
                    raise RuntimeError('mixed dense/sparse exchange of the embedding table needs the phased backward pass')
                tid = id(sp[0])
                if phase == sp[3]:                      # dense (tied decoder) part: travels with this phase's ranges
                    a = self.arena.offsets[tid]
                    return self.arena.all_reduce_mean(task, phase, wait, extra=((a, a + sp[0].numel()),))
                if phase == self.arena.phase[tid]:      # lookup part: after every dense all-reduce has landed
                    self.arena.all_reduce_mean(task, phase, True, frozenset((tid,)))
                    return self._reduce_sparse(key, add=True)
                return self.arena.all_reduce_mean(task, phase, wait, frozenset((tid,)))
            return self.arena.all_reduce_mean(task, phase, wait)
        key = task.split('_')[0]
        gb = self._buckets.get(key)
        if gb is None:
            if grads is not None:
                params = [p for p, g in zip(self.module.parameters(), grads) if g is not None]
            else:
                params = [p for p in self.module.parameters() if p.grad is not None]
            gb = GradBuckets(params, self.bucket_bytes, self.wire_dtype)
            self._buckets[key] = gb
        if grads is not None:
            grads = [g for g in grads if g is not None]
        if gb.params:                      # (a model that took no part in the loss — the critic outside RL training — has nothing to exchange)
            gb.all_reduce_mean(grads)


_EAGER_SINCE_QUIESCE = [0]      # eager RCCL collectives of the gradient arena issued since the last quiesce_collectives()


def _note_collective(t):
    """Bookkeeping behind quiesce_collectives (nothing used to enforce the call).  An arena collective that is being CAPTURED while
    eager RCCL collectives issued earlier may still sit with the process group's watchdog raises here, with the remedy in the message — instead
    of the 1-in-3 process abort (hipErrorCapturedEvent in the watchdog thread) that the missing call used to produce."""
    if not (t.is_cuda and dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
        return
    if torch.cuda.is_current_stream_capturing():
        if _EAGER_SINCE_QUIESCE[0] > 0:
            raise RuntimeError('a gradient collective is being captured into a hipGraph while %d eager collective(s) issued before the capture '
                               'may still be polled by the RCCL watchdog: call vln_goat_amd.dp.quiesce_collectives() right before the capture '
                               'begins' % _EAGER_SINCE_QUIESCE[0])
    else:
        _EAGER_SINCE_QUIESCE[0] += 1


def quiesce_if_needed():
    """called by hipops.graph right BEFORE a capture begins (a device synchronise is still legal there): retire the eager RCCL collectives
    issued since the last quiesce, if any — so a capture that contains collectives never meets a watchdog that still polls earlier ones, and
    a capture long after the last eager collective costs nothing (the guard used to raise on a stale count)."""
    if _EAGER_SINCE_QUIESCE[0] > 0:
        quiesce_collectives()


def quiesce_collectives(seconds=0.3):
    """Call right before a hipGraph capture that will contain collectives.  ProcessGroupNCCL's watchdog thread polls the end events of the
    eager collectives issued so far (every 100 ms); on this HIP runtime `hipEventQuery` fails with hipErrorCapturedEvent once the STREAM an
    event was recorded on has entered capture mode — even though the event itself was recorded before — and the watchdog then takes the
    process down.  After a device synchronise every earlier collective is complete, and one watchdog period later it has retired them all:
    nothing is left to poll while the communication stream is capturing.  (Found as a 1-in-3 flake of scripts/in_graph_comm_check.py.)"""
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized():
        time.sleep(seconds)
    _EAGER_SINCE_QUIESCE[0] = 0


def wrap_finetune_models(vln_bert, critic=None, **kw):
    """The fine-tuning agent's two models under data parallelism, as M/r2r/agent_base.py:100-102 (and M/reverie/agent_base.py:114-115)
    wraps them in DDP(find_unused_parameters=True): rank 0's parameters and buffers broadcast at construction, one process per GPU,
    every rank rolls out its own shard of the batch (M/r2r/env.py shards the instruction list by rank), the losses of the iteration's
    rollouts (teacher + sampled, M/r2r/agent.py:414-437) are summed, ONE backward, gradients averaged over ranks.  The wrappers pass
    `w(mode, batch)` through to the models; the gradient usage key of the whole iteration is 'nav':
        w, wc = dp.wrap_finetune_models(vln_bert, critic)
        ... first iteration: loss.backward(); w.record_usage('nav'); arena = w.build_arena() ...
        arena.zero('nav'); loss = rollouts(w); loss.backward(); dp.reduce_finetune_gradients((w, wc))
    The critic receives no gradient outside RL training (train_rl is commented out upstream): its exchange is empty, which is what
    find_unused_parameters=True makes of it in the reference."""
    w = GoatDataParallel(vln_bert, share_cfp_negatives=False, **kw)
    wc = GoatDataParallel(critic, share_cfp_negatives=False, **kw) if critic is not None else None
    return w, wc


def reduce_finetune_gradients(wrappers, key='nav', wait=True):
    """gradient average of one fine-tuning iteration over ranks, for every wrapped model that produced gradients."""
    for w in wrappers:
        if w is not None:
            if w.arena is not None:
                w.arena.all_reduce_mean(key, None, wait)
            else:
                w.reduce_gradients(key)


def broadcast_task(task_names, chosen_index, device):
    """rank 0 picks the task, everyone follows (P/data/loader.py:56-59)."""
    t = torch.tensor([chosen_index], dtype=torch.int64, device=device)
    if _world() > 1:
        _note_collective(t)
        dist.broadcast(t, 0)
    return task_names[int(t.item())]
