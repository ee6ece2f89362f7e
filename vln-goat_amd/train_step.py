"""The pre-training step around the hot path: what the reference's trainer does per iteration, as a
harness of this package (the trainer itself — data loaders, logging, checkpoints — is out of scope).

Reference behaviour reproduced here (P = /root/reference/pretrain_src):
  * task choice     P/data/loader.py:54-61    one multinomial draw over the mix ratios every `accum_steps` iterations,
                                               rank 0's draw broadcast so every rank trains the same task
  * forward/backward P/train_r2r_goat.py:301-327  task = name.split('_')[0]; loss_vec = model(batch, task, True);
                                               loss = loss_vec.mean() / gradient_accumulation_steps; loss.backward()
  * update          P/train_r2r_goat.py:330-363  every `accum_steps` iterations: learning rate from the schedule,
                                               clip_grad_norm_(grad_norm) unless -1, optimizer.step(), zero_grad()
The reference averages gradients over ranks inside DDP's backward hooks; here the average is one call after the last
backward of the accumulation window (GoatDataParallel.reduce_gradients with the gradient arena, GradBuckets without):
the all-reduce is linear, so the result is the same.
"""
import torch
import torch.distributed as dist

from . import dp


class TaskSampler:
    """Indefinite task-name stream of the reference's MetaLoader (without its data loaders)."""

    def __init__(self, names, ratios, accum_steps=1, device='cpu', generator=None):
        if len(names) != len(ratios) or not names:
            raise ValueError('one sampling ratio per task name')
        self.names = list(names)
        self.ratios = torch.tensor([float(r) for r in ratios], dtype=torch.float32)
        self.accum_steps = max(1, int(accum_steps))
        self.device = torch.device(device)
        self.generator = generator
        self.step = 0
        self._task_id = None

    def next(self):
        if self.step % self.accum_steps == 0:
            tid = torch.multinomial(self.ratios, 1, generator=self.generator).to(self.device)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.broadcast(tid, 0)                      # every rank follows rank 0's draw
            self._task_id = int(tid.cpu().item())
        self.step += 1
        return self.names[self._task_id]


class PretrainStep:
    """One iteration of the reference's pre-training loop on a model of this package (or any module with the
    `model(batch, task, compute_loss)` contract).

        step = PretrainStep(model, optimizer, grad_accum=1, grad_norm=5.0, wrapper=GoatDataParallel(model) or None)
        info = step(name, batch)      # {'task', 'loss', 'n_loss_units', 'updated', 'grad_norm'}  (grad_norm: a float, or with the
                                      #  fused optimizer a callable that reads the device scalar on request)
    """

    def __init__(self, model, optimizer=None, grad_accum=1, grad_norm=5.0, wrapper=None, lr_schedule=None):
        self.model, self.optimizer, self.wrapper = model, optimizer, wrapper
        self.grad_accum = max(1, int(grad_accum))
        self.grad_norm = grad_norm
        self.lr_schedule = lr_schedule            # callable(global_step) -> learning rate, or None
        self.micro_step = 0
        self.global_step = 0
        self._buckets = {}

    def _arena(self):
        return getattr(self.wrapper, 'arena', None) if self.wrapper is not None else None

    def _zero(self, task):
        arena = self._arena()
        if arena is not None:
            arena.bind(task)                      # .grad = arena view for the parameters `task` uses, None for the others: the
            arena.zero(task)                      # optimizer skips them, as after the reference's zero_grad + DDP unused-parameter step
        elif self.optimizer is not None:
            self.optimizer.zero_grad()
        else:
            for p in self.model.parameters():
                p.grad = None

    def _average(self, task):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        if self.wrapper is not None:
            self.wrapper.reduce_gradients(task)
            return
        params = [p for p in self.model.parameters() if p.grad is not None]
        key = (task, tuple(id(p) for p in params))
        gb = self._buckets.get(key)
        if gb is None:
            gb = self._buckets[key] = dp.GradBuckets(params)
        gb.all_reduce_mean()

    def __call__(self, name, batch):
        task = name.split('_')[0]
        if self.micro_step % self.grad_accum == 0:
            self._zero(task)
            if self.wrapper is not None:
                self.wrapper.begin_step(task)
        loss_vec = self.model(batch, task, True)
        n_units = int(loss_vec.shape[0])
        loss = loss_vec.mean()                    # the model returns un-reduced losses
        if self.grad_accum > 1:
            loss = loss / self.grad_accum
        loss.backward()
        self.micro_step += 1
        info = {'task': task, 'loss': float(loss.detach()), 'n_loss_units': n_units, 'updated': False, 'grad_norm': None}
        if self.micro_step % self.grad_accum != 0:
            return info
        arena = self._arena()
        if arena is not None:
            arena.close_step()                    # slices a kernel "owned" last time but did not write this time must not keep old values
        self._average(task)
        self.global_step += 1
        if self.optimizer is not None:
            if self.lr_schedule is not None:
                lr = self.lr_schedule(self.global_step)
                for g in self.optimizer.param_groups:
                    g['lr'] = lr
            if hasattr(self.optimizer, 'arena'):       # optim.FusedAdamW: norm, clip and update in two kernels on the arena
                self.optimizer.step(task, max_norm=self.grad_norm if (self.grad_norm is not None and self.grad_norm != -1) else None)
                info['grad_norm'] = self.optimizer.last_grad_norm
            else:
                if self.grad_norm is not None and self.grad_norm != -1:
                    info['grad_norm'] = float(torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm))
                self.optimizer.step()
        info['updated'] = True
        return info


def collate_indices(config, batch, tasks=('mlm', 'sap', 'cfp')):
    """Everything a pre-training step derives from the id STRINGS and label positions of a batch, built on the host from the
    CPU batch (what pretrain_model.GlocalTextPathCMTPreTraining otherwise builds lazily on first use and caches in
    batch['_goat_cache']): the graph-map / local-branch gather indices (graphmap.py; P/model/vilmodel_goat.py:377-391,430-468),
    the MLM row selection (P/model/pretrain_goat.py:196-206) and the SAP fusion matrix (:329-345).  R2R batches (views only)."""
    from . import graphmap
    if batch.get('traj_obj_img_fts') is not None:
        raise NotImplementedError('collate_indices: object batches (REVERIE/SOON) build their indices lazily')
    V = batch['traj_view_img_fts'].shape[1]
    G = batch['gmap_step_ids'].shape[1]
    lens = batch['traj_vp_view_lens']
    out = {}
    out['gmap'] = graphmap.build_gmap_index(batch['traj_step_lens'], lens, batch['traj_vpids'], batch['traj_cand_vpids'],
                                            batch['gmap_vpids'], G, V, bool(config.adaptive_pano_fusion))
    out['vp'] = graphmap.build_vp_index(batch['traj_step_lens'], lens, V)
    n_rows = int(batch['traj_view_img_fts'].shape[0])
    fused = bool(config.adaptive_pano_fusion)
    out['gmap_inv'] = graphmap.inverse_index(out['gmap'][0], out['gmap'][1], out['gmap'][2], n_rows * V + (n_rows if fused else 0))
    out['vp_inv'] = tuple(t for t in graphmap.inverse_index(out['vp'][0], out['vp'][1], None, n_rows * V) if t is not None)
    W = out['vp'][3]
    if 'mlm' in tasks:
        labels = batch['txt_labels'].reshape(-1)
        idx = (labels != -1).nonzero().squeeze(1)
        out['mlm_idx'], out['mlm_tgt'] = idx, labels[idx]
    if 'sap' in tasks:
        last = torch.as_tensor(batch['traj_step_lens']).cumsum(0) - 1
        nav = batch['traj_nav_types'][last] != 1
        nav = torch.cat([torch.zeros(nav.shape[0], 1, dtype=torch.bool), nav], 1)[:, :W]
        out['sap'] = (nav, graphmap.build_sap_fusion(batch['traj_cand_vpids'], batch['gmap_vpids'], batch['gmap_visited_masks'], G, W))
    return out


class StaticBatch:
    """A device batch at FIXED addresses behind a captured (hipGraph) step, fed with a new host batch per step.

    Every tensor of the batch and every index tensor of `collate_indices` is a view into ONE device buffer; a host batch is
    packed into a pinned buffer of the same layout (`pack`: the collate step of a loader, P/data/loader.py:78-120), moved with one
    asynchronous H2D copy into a staging buffer on a side stream (`stage`: overlaps the running step) and swapped in with
    one D2D copy on the compute stream (`commit`), after which the memoised masks are refreshed in place.  The shapes are
    those of the batch the object was built from: a host batch with any other shape (ragged T / L, another map size) raises
    ValueError — such batches go through the eager path (or a StaticBatch of their own shape bucket).

        sb = StaticBatch(model.config, host_batch, tasks)       # sb.gb: the device batch to capture the step on
        ... capture model(sb.gb, task) ...
        buf = sb.pack(next_host_batch); sb.stage(buf); sb.commit(); graph.replay()
    """
    ALIGN = 256

    def __init__(self, config, host_batch, tasks=('mlm', 'sap', 'cfp'), device='cuda'):
        self.config, self.tasks, self.device = config, tuple(tasks), torch.device(device)
        idx = collate_indices(config, host_batch, self.tasks)
        self.layout = []                   # (key path, offset, shape, dtype)
        off = 0
        for path, t in self._tensors(host_batch, idx):
            n = t.numel() * t.element_size()
            self.layout.append((path, off, tuple(t.shape), t.dtype))
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.nbytes = max(off, self.ALIGN)
        self.flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.staging = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.vp_width = idx['vp'][3]
        self.gb = self._views(self.flat, host_batch)
        self.side = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._staged = torch.cuda.Event() if self.side is not None else None
        self._consumed = torch.cuda.Event() if self.side is not None else None
        first = self.pack(host_batch, _idx=idx)
        self.flat.copy_(first)
        self._pending = None

    @staticmethod
    def _tensors(batch, idx):
        for k in sorted(batch):
            if torch.is_tensor(batch[k]):
                yield ('batch', k), batch[k]
        for k in sorted(idx):
            v = idx[k]
            for i, t in enumerate(v if isinstance(v, tuple) else (v,)):
                if torch.is_tensor(t):
                    yield ('idx', k, i), t

    def _views(self, flat, host_batch):
        gb = {k: v for k, v in host_batch.items() if not torch.is_tensor(v) and k != '_goat_cache'}
        parts = {}
        for path, off, shape, dtype in self.layout:
            n = 1
            for s in shape:
                n *= s
            v = flat[off:off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(shape)
            if path[0] == 'batch':
                gb[path[1]] = v
            else:
                parts.setdefault(path[1], {})[path[2]] = v
        cache = {}
        for k, d in parts.items():
            if k == 'vp':
                cache[k] = (d[0], d[1], d[2], self.vp_width)
            elif k in ('mlm_idx', 'mlm_tgt'):
                cache[k] = d[0]
            else:
                cache[k] = tuple(d[i] for i in sorted(d))
        gb['_goat_cache'] = cache
        return gb

    def new_pinned(self):
        t = torch.empty(self.nbytes, dtype=torch.uint8)
        return t.pin_memory() if self.device.type == 'cuda' else t

    def pack(self, host_batch, out=None, _idx=None, tensors=True):
        """host batch -> flat (pinned) buffer in the device layout; builds the batch's index tensors on the way.
        tensors=False: only the index tensors are (re)built and written — for a loader that collated the batch's tensors
        straight into `out` (its previous pack)."""
        idx = _idx if _idx is not None else collate_indices(self.config, host_batch, self.tasks)
        if idx['vp'][3] != self.vp_width:
            raise ValueError('StaticBatch: local-branch width %d != %d of the captured shape' % (idx['vp'][3], self.vp_width))
        out = out if out is not None else self.new_pinned()
        got = dict(self._tensors(host_batch, idx))
        dst = out.numpy()
        if len(got) != len(self.layout):
            raise ValueError('StaticBatch: the batch has a different set of tensors than the captured one')
        for path, off, shape, dtype in self.layout:
            if not tensors and path[0] == 'batch':
                continue
            t = got.get(path)
            if t is None or tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError('StaticBatch: %s is %s %s, the captured shape is %s %s'
                                 % ('/'.join(map(str, path)), None if t is None else tuple(t.shape), None if t is None else t.dtype, shape, dtype))
            n = t.numel() * t.element_size()
            # (plain memcpy through numpy: torch's threaded copy_ costs more in thread wake-ups than it saves on 27 MB)
            dst[off:off + n] = t.contiguous().view(-1).view(torch.uint8).numpy() if n else dst[off:off]
        return out

    def stage(self, packed):
        """asynchronous H2D of a packed batch into the staging buffer (side stream; waits until the previous commit has read it).
        -> event that completes when `packed` has been read (None on CPU): synchronise on it before rewriting the buffer."""
        if self.side is None:
            self._pending = packed
            return None
        self.side.wait_event(self._consumed)
        with torch.cuda.stream(self.side):
            self.staging.copy_(packed, non_blocking=True)
            self._staged.record(self.side)
            done = torch.cuda.Event()
            done.record(self.side)
        self._pending = packed
        return done

    def commit(self):
        """compute stream: staged batch -> the static buffer (one D2D copy), then the memoised masks are recomputed in place."""
        from . import layers
        if self._pending is None:
            raise RuntimeError('StaticBatch.commit without a staged batch')
        if self.side is None:
            self.flat.copy_(self._pending)
        else:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._staged)
            self.flat.copy_(self.staging, non_blocking=True)
            self._consumed.record(cur)
        self._pending = None
        layers.refresh_masks()
