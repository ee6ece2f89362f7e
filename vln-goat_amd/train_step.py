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
