"""SURVEY §8 a-18 — the pre-training step harness (vln_goat_amd.train_step) at world size 2 (gloo, CPU), on a toy module with
the `model(batch, task, compute_loss)` contract: every rank trains rank 0's task draw, gradients are the rank average,
clipping / learning-rate schedule / optimizer step follow P/train_r2r_goat.py:301-363, and the result equals a
single-process step on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Toy(torch.nn.Module):
    """two task heads on a shared trunk; returns un-reduced per-sample losses like the GOAT pre-training model"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.trunk = torch.nn.Linear(6, 8)
        self.heads = torch.nn.ModuleDict({'mlm': torch.nn.Linear(8, 1), 'sap': torch.nn.Linear(8, 1)})

    def forward(self, batch, task, compute_loss=True):
        return (self.heads[task](torch.tanh(self.trunk(batch['x']))).squeeze(-1) - batch['y']).pow(2)


def _batch(seed, n=4):
    g = torch.Generator().manual_seed(seed)
    return {'x': torch.randn(n, 6, generator=g) * 3, 'y': torch.randn(n, generator=g)}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vln_goat_amd import train_step
    # different generators per rank: only rank 0's draw may count
    sampler = train_step.TaskSampler(['mlm', 'sap_r2r'], [1, 1], accum_steps=2, generator=torch.Generator().manual_seed(10 + 77 * rank))
    names = [sampler.next() for _ in range(12)]
    model = Toy()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    step = train_step.PretrainStep(model, opt, grad_accum=2, grad_norm=0.05, lr_schedule=lambda s: 0.1 * s)
    infos = [step(names[i], _batch(100 + 10 * i + rank)) for i in range(2)]           # one accumulation window
    q.put((rank, names, [i['updated'] for i in infos], infos[-1]['grad_norm'], [i['n_loss_units'] for i in infos],
           {k: v.detach().numpy() for k, v in model.state_dict().items()}, step.global_step, opt.param_groups[0]['lr']))
    dist.barrier()
    dist.destroy_process_group()


def test_pretrain_step_two_ranks_equals_single_process_on_the_concatenated_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    r0, r1 = res
    assert r0[1] == r1[1]                                    # same task on every rank ...
    assert all(r0[1][i] == r0[1][i + 1] for i in range(0, 12, 2))      # ... held for a whole accumulation window
    assert len(set(r0[1])) == 2                              # and both tasks do occur
    assert r0[2] == [False, True] and r0[4] == [4, 4]
    assert r0[6] == 1 and abs(r0[7] - 0.1) < 1e-9            # one optimizer step, learning rate from the schedule
    for k in r0[5]:
        assert np.array_equal(r0[5][k], r1[5][k]), k         # ranks stay in lockstep

    # single process, same window, both ranks' samples in each micro-batch
    model = Toy()
    task = r0[1][0].split('_')[0]
    for i in range(2):
        b = [_batch(100 + 10 * i + r) for r in range(2)]
        cat = {k: torch.cat([b[0][k], b[1][k]]) for k in b[0]}
        (model(cat, task, True).mean() / 2).backward()
    gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.05))
    assert gn > 0.05                                         # the clip is active in this case
    assert abs(gn - r0[3]) / gn < 1e-5
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p -= 0.1 * p.grad
    for k, v in model.state_dict().items():
        assert np.allclose(v.numpy(), r0[5][k], rtol=1e-5, atol=1e-7), k
