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
                from . import dp
                dp._note_collective(tid)
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


import numpy as np      # noqa: E402
_NP_OF = {torch.float32: np.float32, torch.bfloat16: np.uint16, torch.float16: np.float16, torch.int64: np.int64, torch.int32: np.int32,
          torch.bool: np.bool_, torch.uint8: np.uint8}
_INT_VIEW = {torch.bfloat16: torch.int16}        # numpy has no bfloat16: move the bits as 16-bit integers

PAD_FILL = {'txt_labels': -1, 'traj_vp_view_lens': 1}          # every other tensor is padded with zeros


BIG_TENSORS = ('traj_view_img_fts', 'traj_obj_img_fts')


def pad_batch(batch, L=None, N=None, G=None, W=None, shape_only=()):
    """shape_only: keys whose padded tensor is only needed for its SHAPE (a zero-stride view, no memory) — StaticBatch.pack writes
    the real rows of the 27 MB feature tensor straight into the pinned buffer instead of padding a copy first.
    Pad a host pre-training batch (the *_collate schema of P/data/tasks.py:110,392,618, which pads to the per-batch maxima) up to
    a SHAPE BUCKET: text length L, panorama count N (= sum of trajectory lengths), map size G, local width W (views + [stop]).
    Padding panoramas are appended after the last real one with view length 1 (no index ever refers to them), padded text slots
    carry label -1, padded map slots lie beyond gmap_lens.  -> a new dict (tensors re-allocated only when they grow)."""
    out = dict(batch)

    def grow(key, dims, fill=None):
        t = batch.get(key)
        if t is None or not torch.is_tensor(t):
            return
        shape = list(t.shape)
        for d, n in dims:
            if n is not None:
                if shape[d] > n:
                    raise ValueError('pad_batch: %s has %d along dim %d, the bucket holds %d' % (key, shape[d], d, n))
                shape[d] = n
        if shape == list(t.shape):
            return
        if key in shape_only:
            out[key] = torch.zeros((1,) * len(shape), dtype=t.dtype).expand(shape)
            return
        fill_v = PAD_FILL.get(key, 0) if fill is None else fill
        # numpy, not torch: a torch copy of a few hundred KB wakes the intra-op thread pool (128 threads on the GPU box's host), which
        # costs 80-90 ms every few calls — measured: 1.6 ms per pack with numpy, 26 ms on average with torch indexing
        src = t.view(_INT_VIEW[t.dtype]).numpy() if t.dtype in _INT_VIEW else t.numpy()
        arr = np.zeros(shape, dtype=src.dtype) if not fill_v else np.full(shape, fill_v, dtype=src.dtype)
        arr[tuple(slice(0, k) for k in t.shape)] = src
        new = torch.from_numpy(arr)
        out[key] = new.view(t.dtype) if t.dtype in _INT_VIEW else new
    for k in ('txt_ids', 'txt_labels'):
        grow(k, [(1, L)])
    for k in ('traj_view_img_fts', 'traj_loc_fts', 'traj_nav_types', 'traj_vp_view_lens', 'traj_obj_img_fts', 'traj_vp_obj_lens',
              'traj_reverie_obj_names'):
        grow(k, [(0, N)])
    for k in ('gmap_step_ids', 'gmap_pos_fts', 'gmap_visited_masks'):
        grow(k, [(1, G)])
    grow('gmap_pair_dists', [(1, G), (2, G)])
    grow('vp_pos_fts', [(1, W)])
    # the batch's own padded widths: the reference's un-masked CFP pooling runs over exactly these (collate_indices -> cfp_*_mask)
    out['_own'] = batch.get('_own') or {'L': int(batch['txt_ids'].shape[1]), 'G': int(batch['gmap_step_ids'].shape[1]),
                                        'W': int(batch['vp_pos_fts'].shape[1])}
    return out


POSITION_FEATURES = ('traj_loc_fts', 'gmap_pos_fts', 'vp_pos_fts', 'loc_fts')      # ('loc_fts': the fine-tuning model's panorama input)


def prepare_position_features(batch, dtype=None):
    """The 7- / 14-wide angle / distance features (P/data/tasks.py collates: traj_loc_fts, gmap_pos_fts, vp_pos_fts) in the form their
    first Linear consumes: compute dtype, last dimension zero-padded to the GEMM's 16-byte K chunk (8 bf16 / 4 float32 elements).  Done
    once per batch where the batch is built (host side), so the step launches neither the cast nor the pad (2 launches per Linear and step
    before).  hipops.linear accepts both forms; the padded columns meet zero weight columns.  -> a new dict."""
    from . import layers
    dt = dtype or layers.compute_dtype()
    e = 8 if dt == torch.bfloat16 else 4
    out = dict(batch)
    for k in POSITION_FEATURES:
        t = batch.get(k)
        if not torch.is_tensor(t) or not t.is_floating_point():
            continue
        pad = (-t.shape[-1]) % e
        if t.shape[-1] > 16 or (pad == 0 and t.dtype == dt):
            continue
        out[k] = _cast_pad_host(t, dt, pad)
    return out


def _cast_pad_host(t, dt, pad):
    """float32 host tensor -> `dt` with `pad` zero columns appended, through numpy: torch's own cast / pad of a few hundred KB wakes the
    intra-op thread pool of the GPU boxes' 128-256-core hosts (measured on the fresh-batch leg: 6.1 -> 16.9 ms per step with torch ops
    here).  bfloat16 = round-to-nearest-even of the float32 bits, as torch's .to(torch.bfloat16)."""
    if t.is_cuda or t.dtype != torch.float32 or dt not in (torch.float32, torch.bfloat16):
        t = t.to(dt)
        return torch.nn.functional.pad(t, (0, pad)) if pad else t
    src = np.ascontiguousarray(t.numpy())
    shape = src.shape[:-1] + (src.shape[-1] + pad,)
    if dt == torch.float32:
        arr = np.zeros(shape, np.float32)
        arr[..., :src.shape[-1]] = src
        return torch.from_numpy(arr)
    u = src.view(np.uint32)
    nan = np.isnan(src)
    r = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)).astype(np.uint16)
    if nan.any():
        r[nan] = 0x7FC0
    arr = np.zeros(shape, np.uint16)
    arr[..., :src.shape[-1]] = r
    return torch.from_numpy(arr.view(np.int16)).view(torch.bfloat16)


def index_capacities(batch, tasks, mlm_rate=0.25):
    """upper bounds of the variable-length index tensors of collate_indices for every batch that fits the shapes of `batch`."""
    Nn, V = batch['traj_view_img_fts'].shape[:2]
    B, Lb = batch['txt_ids'].shape
    W = batch['vp_pos_fts'].shape[1]
    O = batch['traj_obj_img_fts'].shape[1] if torch.is_tensor(batch.get('traj_obj_img_fts')) else 0
    # the MRC selections draw from the B last panoramas' views / objects, not from the text: capacities of their own
    return {'nnz_gmap': Nn * V + Nn, 'nnz_vp': B * W, 'mlm': max(8, int(mlm_rate * B * Lb)),
            'mrc_view': max(8, B * V), 'mrc_obj': max(8, B * O)}


def _pad1(t, n, fill):
    if t.shape[0] > n:
        raise ValueError('index tensor of %d entries exceeds the bucket capacity %d' % (t.shape[0], n))
    if t.shape[0] == n:
        return t
    out = torch.full((n,) + tuple(t.shape[1:]), fill, dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


def collate_indices(config, batch, tasks=('mlm', 'sap', 'cfp'), caps=None, vp_width=None):
    """Everything a pre-training step derives from the id STRINGS and label positions of a batch, built on the host from the
    CPU batch (what pretrain_model.GlocalTextPathCMTPreTraining otherwise builds lazily on first use and caches in
    batch['_goat_cache']): the graph-map / local-branch gather indices (graphmap.py; P/model/vilmodel_goat.py:377-391,430-468), the
    view + object row assembly of REVERIE / SOON batches (:331-341), the MLM row selection (P/model/pretrain_goat.py:196-206), the
    SAP fusion matrix (:329-345) and the OG logit gather (:356-391).
    caps (index_capacities): pad every variable-length index tensor to a fixed capacity — entries past the last segment are never
    read, MLM padding rows select row 0 with target -100 (ignored by goat_ce_*) and `mlm_scale` = capacity / real rows restores the
    mean — so that batches of one shape bucket share ONE device layout.  vp_width: local width of the bucket (default: the batch's own)."""
    from . import graphmap
    V = batch['traj_view_img_fts'].shape[1]
    G = batch['gmap_step_ids'].shape[1]
    lens = batch['traj_vp_view_lens']
    fused = bool(config.adaptive_pano_fusion)
    n_rows = int(batch['traj_view_img_fts'].shape[0])
    out = {}
    n_real = int(sum(batch['traj_step_lens']))            # (a padded batch carries dummy panoramas after the real ones)
    if batch.get('traj_obj_img_fts') is not None:
        obj = batch['traj_vp_obj_lens']
        O = batch['traj_obj_img_fts'].shape[1]
        Wp = batch['traj_nav_types'].shape[1]
        ci = graphmap.build_obj_concat_index(lens, obj, V, O, Wp)
        out['objcat'] = ci
        out['objcat_inv'] = tuple(t for t in graphmap.inverse_index(ci[0], ci[1], None, n_rows * V + n_rows * O) if t is not None)
        out['view_lens_cpu'], out['obj_lens_cpu'] = lens.clone(), obj.clone()
        lens, V = lens + obj, Wp
    g = graphmap.build_gmap_index(batch['traj_step_lens'], lens[:n_real], batch['traj_vpids'], batch['traj_cand_vpids'],
                                  batch['gmap_vpids'], G, V, False) if not fused else None
    if fused:
        # the fused rows sit behind ALL view rows of the (possibly padded) batch: row n_rows * V + n
        g = _gmap_index_fused(batch, lens[:n_real], G, V, n_rows)
    v = graphmap.build_vp_index(batch['traj_step_lens'], lens[:n_real], V)
    if vp_width is not None and v[3] != vp_width:
        v = _vp_index_width(batch['traj_step_lens'], lens[:n_real], V, vp_width)
    out['gmap'], out['vp'] = g, v
    out['gmap_inv'] = graphmap.inverse_index(g[0], g[1], g[2], n_rows * V + (n_rows if fused else 0))
    out['vp_inv'] = tuple(t for t in graphmap.inverse_index(v[0], v[1], None, n_rows * V) if t is not None)
    W = v[3]
    if 'mlm' in tasks:
        labels = batch['txt_labels'].reshape(-1)
        idx = (labels != -1).nonzero().squeeze(1)
        out['mlm_idx'], out['mlm_tgt'] = idx, labels[idx]
    if 'sap' in tasks:
        last = torch.as_tensor(batch['traj_step_lens']).cumsum(0) - 1
        nav = batch['traj_nav_types'][last] != 1
        nav = torch.cat([torch.zeros(nav.shape[0], 1, dtype=torch.bool), nav], 1)[:, :W]
        if nav.shape[1] < W:
            nav = torch.cat([nav, torch.ones(nav.shape[0], W - nav.shape[1], dtype=torch.bool)], 1)
        out['sap'] = (nav, graphmap.build_sap_fusion(batch['traj_cand_vpids'], batch['gmap_vpids'], batch['gmap_visited_masks'], G, W))
        # goat_sap_fuse skips a label outside its logit row (it must: -100 is the ignore value); F.cross_entropy, which it replaces, raises
        # on such targets — so a mislabelled or bucket-mismatched batch is refused here, on the host, before it trains on silence
        for key, width in (('global_act_labels', G), ('local_act_labels', W)):
            lab = batch.get(key)
            if torch.is_tensor(lab):
                bad = (lab != -100) & ((lab < 0) | (lab >= width))
                if bool(bad.any()):
                    raise ValueError('collate_indices: %s holds %d outside [0, %d) (and not the ignore value -100)' % (key, int(lab[bad][0]), width))
    has_obj = batch.get('traj_obj_img_fts') is not None
    last = torch.as_tensor(batch['traj_step_lens']).cumsum(0) - 1
    if 'og' in tasks and has_obj:
        # OG logits gathered from the local tokens (pretrain_model.forward_og): object slots follow the [stop] token and the views
        vl, ol = out['view_lens_cpu'][last].tolist(), out['obj_lens_cpu'][last].tolist()
        O = batch['traj_obj_img_fts'].shape[1] if caps is not None else max(1, max(ol))
        oi, om = torch.zeros(len(vl), O, dtype=torch.int64), torch.zeros(len(vl), O, dtype=torch.bool)
        for b, (v_, o_) in enumerate(zip(vl, ol)):
            oi[b, :o_] = torch.arange(1 + v_, 1 + v_ + o_)
            om[b, :o_] = True
        out['og_idx'] = (oi, om)
    if 'mrc' in tasks:
        vl = (out['view_lens_cpu'] if has_obj else batch['traj_vp_view_lens'])[last].tolist()
        ol = out['obj_lens_cpu'][last].tolist() if has_obj else [0] * len(vl)
        for which, mkey in (('view', 'vp_view_mrc_masks'), ('obj', 'vp_obj_mrc_masks')):
            mask = batch.get(mkey)
            if mask is None:
                continue
            rows = []
            for b in range(mask.shape[0]):
                n = vl[b] if which == 'view' else ol[b]
                off = 1 if which == 'view' else 1 + vl[b]
                for j in mask[b].nonzero().squeeze(1).tolist():
                    if j >= n:
                        raise ValueError('MRC mask selects a padded %s slot (sample %d, slot %d)' % (which, b, j))
                    rows.append(b * W + off + j)
            out['mrc_' + which] = (torch.tensor(rows, dtype=torch.int64), mask.reshape(-1).nonzero().squeeze(1))
    if caps is not None:
        c = caps
        for which in ('view', 'obj'):
            if 'mrc_' + which in out:
                rows, sel = out['mrc_' + which]
                n, cap = int(rows.shape[0]), c.get('mrc_' + which, c['mlm'])
                if n == 0:
                    raise ValueError('collate_indices: a bucketed batch needs at least one masked %s region' % which)
                w = torch.zeros(cap, dtype=torch.float32)
                w[:n] = 1.0
                out['mrc_' + which] = (_pad1(rows, cap, int(rows[0])), _pad1(sel, cap, int(sel[0])))
                out['mrc_%s_w' % which] = w
        out['gmap'] = (_pad1(g[0], c['nnz_gmap'], -1), g[1], g[2])
        gi = out['gmap_inv']
        out['gmap_inv'] = (_pad1(gi[0], c['nnz_gmap'], -1), gi[1], _pad1(gi[2], c['nnz_gmap'], 0.0))
        out['vp'] = (_pad1(v[0], c['nnz_vp'], -1), v[1], v[2], v[3])
        vi = out['vp_inv']
        out['vp_inv'] = (_pad1(vi[0], c['nnz_vp'], -1),) + tuple(vi[1:])
        if 'cfp' in tasks:
            # the CFP heads pool over ALL slots of the batch's own padded width, padding included (reference quirk):
            # bucket padding beyond that width must not take part
            own = batch.get('_own') or {'L': batch['txt_ids'].shape[1], 'G': G}
            B = batch['txt_ids'].shape[0]
            tm = torch.zeros(B, batch['txt_ids'].shape[1], dtype=torch.float32)
            tm[:, own['L']:] = float('-inf')
            gm = torch.zeros(B, G, dtype=torch.float32)
            gm[:, own['G']:] = float('-inf')
            vm = torch.zeros(B, W, dtype=torch.float32)
            vm[:, own.get('W', W):] = float('-inf')
            out['cfp_txt_mask'], out['cfp_gmap_mask'], out['cfp_vp_mask'] = tm, gm, vm
        if 'mlm' in tasks:
            n = int(out['mlm_idx'].shape[0])
            if n == 0:
                raise ValueError('collate_indices: a bucketed batch needs at least one masked token')
            out['mlm_idx'] = _pad1(out['mlm_idx'], c['mlm'], 0)
            out['mlm_tgt'] = _pad1(out['mlm_tgt'], c['mlm'], -100)
            out['mlm_scale'] = torch.tensor([float(c['mlm']) / n], dtype=torch.float32)
    return out


def _gmap_index_fused(batch, lens, G, V, n_rows):
    from . import graphmap
    n_real = len(lens)
    idx, start, scale = graphmap.build_gmap_index(batch['traj_step_lens'], lens, batch['traj_vpids'], batch['traj_cand_vpids'],
                                                  batch['gmap_vpids'], G, V, True)
    if n_rows != n_real:                 # fused row n was addressed as n_real * V + n: move it behind the padded view rows
        idx = idx.clone()
        m = idx >= n_real * V
        idx[m] += (n_rows - n_real) * V
    return idx, start, scale


def _vp_index_width(traj_step_lens, lens, V, width):
    """graphmap.build_vp_index for a fixed local width (>= the batch's own): the extra slots are empty segments."""
    import numpy as np
    step_lens = np.asarray(list(traj_step_lens), dtype=np.int64)
    view_lens = np.asarray(lens.tolist() if torch.is_tensor(lens) else list(lens), dtype=np.int64)
    B = len(step_lens)
    last = np.cumsum(step_lens) - 1
    vp_lens = view_lens[last] + 1
    own = int(vp_lens.max())
    if own > width:
        raise ValueError('local width %d exceeds the bucket width %d' % (own, width))
    if width - 1 > V:
        raise ValueError('bucket width %d needs %d view slots, the panoramas have %d' % (width, width - 1, V))
    idx = (last[:, None] * V + np.arange(width - 1, dtype=np.int64)[None, :]).reshape(-1)
    per_tok = np.ones((B, width), dtype=np.int64)
    per_tok[:, 0] = 0
    start = np.concatenate([[0], np.cumsum(per_tok.reshape(-1))])
    return (torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(start.astype(np.int32)), torch.from_numpy(vp_lens.astype(np.int64)), width)


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

    def __init__(self, config, host_batch, tasks=('mlm', 'sap', 'cfp'), device='cuda', bucket=None):
        """bucket: None — the batch's own shapes, every later batch must match them exactly (round-2 behaviour) — or a dict
        {'L': text length, 'N': panoramas, 'G': map size, 'W': local width} (any subset; missing entries = the first batch's own):
        the object then accepts every RAGGED batch that fits (`pad_batch` + capacity-padded index tensors), so that one captured
        step serves all batches of the bucket instead of falling back to the eager path."""
        self.config, self.tasks, self.device = config, tuple(tasks), torch.device(device)
        from . import layers
        self.pos_dtype = layers.compute_dtype()       # the position features are stored cast + K-padded (prepare_position_features)
        host_batch = prepare_position_features(host_batch, self.pos_dtype)
        self.bucket = None
        if bucket is not None:
            self.bucket = {'L': bucket.get('L', host_batch['txt_ids'].shape[1]), 'N': bucket.get('N', host_batch['traj_view_img_fts'].shape[0]),
                           'G': bucket.get('G', host_batch['gmap_step_ids'].shape[1]), 'W': bucket.get('W', host_batch['vp_pos_fts'].shape[1])}
            host_batch = pad_batch(host_batch, **self.bucket)
            self.caps = index_capacities(host_batch, self.tasks)
        idx = self._collate(host_batch)
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

    def _collate(self, host_batch):
        if self.bucket is None:
            return collate_indices(self.config, host_batch, self.tasks)
        return collate_indices(self.config, host_batch, self.tasks, caps=self.caps, vp_width=self.bucket['W'])

    def fits(self, host_batch):
        """can `host_batch` be packed into this object's layout?"""
        shape = (host_batch['txt_ids'].shape[1], host_batch['traj_view_img_fts'].shape[0], host_batch['gmap_step_ids'].shape[1],
                 host_batch['vp_pos_fts'].shape[1])
        if self.bucket is None:
            want = (self.gb['txt_ids'].shape[1], self.gb['traj_view_img_fts'].shape[0], self.gb['gmap_step_ids'].shape[1],
                    self.gb['vp_pos_fts'].shape[1])
            return shape == want
        b = self.bucket
        return shape[0] <= b['L'] and shape[1] <= b['N'] and shape[2] <= b['G'] and shape[3] <= b['W'] and \
            host_batch['txt_ids'].shape[0] == self.gb['txt_ids'].shape[0] and \
            host_batch['traj_view_img_fts'].shape[1:] == self.gb['traj_view_img_fts'].shape[1:]

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
        gb = {k: v for k, v in host_batch.items() if not torch.is_tensor(v) and k not in ('_goat_cache', '_own')}
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
            elif k in ('mlm_idx', 'mlm_tgt', 'mlm_scale', 'view_lens_cpu', 'obj_lens_cpu', 'mrc_view_w', 'mrc_obj_w', 'cfp_txt_mask', 'cfp_gmap_mask', 'cfp_vp_mask'):
                cache[k] = d[0]
            else:
                cache[k] = tuple(d[i] for i in sorted(d))
        for k in ('view_lens_cpu', 'obj_lens_cpu'):          # host-side bookkeeping of the object branch stays on the host
            cache.pop(k, None)
        gb['_goat_cache'] = cache
        return gb

    def new_pinned(self):
        t = torch.empty(self.nbytes, dtype=torch.uint8)
        return t.pin_memory() if self.device.type == 'cuda' else t

    def pack(self, host_batch, out=None, _idx=None, tensors=True):
        """host batch -> flat (pinned) buffer in the device layout; builds the batch's index tensors on the way.
        tensors=False: only the index tensors are (re)built and written — for a loader that collated the batch's tensors
        straight into `out` (its previous pack)."""
        if _idx is None:
            host_batch = prepare_position_features(host_batch, self.pos_dtype)
        raw = host_batch
        if self.bucket is not None and _idx is None:
            host_batch = pad_batch(host_batch, **self.bucket, shape_only=BIG_TENSORS)
        idx = _idx if _idx is not None else self._collate(host_batch)
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
            if self.bucket is not None and path[0] == 'batch' and path[1] in BIG_TENSORS and t is not None and t.dtype == dtype:
                # the real rows go straight into the corner of the padded slot, the padding slabs are zeroed
                src = raw[path[1]]
                if src.dim() == len(shape) and all(a <= b for a, b in zip(src.shape, shape)):
                    n = 1
                    for d_ in shape:
                        n *= d_
                    view = dst[off:off + n * t.element_size()].view(np.uint8).view(_NP_OF[dtype]).reshape(shape)
                    view[tuple(slice(0, k) for k in src.shape)] = src.view(_INT_VIEW.get(dtype, dtype)).numpy().view(_NP_OF[dtype]) \
                        if dtype in _INT_VIEW else src.numpy()
                    for d_ in range(len(shape)):
                        if src.shape[d_] < shape[d_]:
                            view[tuple(slice(0, src.shape[i]) for i in range(d_)) + (slice(src.shape[d_], None),)] = 0
                    continue
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
