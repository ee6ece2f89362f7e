"""world_size-2 tests (gloo, CPU) of the data-parallel engine: bucketed gradient averaging and the CFP
all-gather whose result must equal the single-process loss/gradients on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker_buckets(rank, world, port, q):
    _init(rank, world, port)
    from vln_goat_amd import dp
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Linear(32, 8), torch.nn.Linear(8, 4))
    unused = torch.nn.Linear(3, 3)          # never receives a gradient (task-dependent parameter subsets)
    holder = torch.nn.ModuleList([model, unused])
    w = dp.GoatDataParallel(holder, bucket_bytes=1024)     # tiny buckets -> several all-reduces
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 16)
    model(x).pow(2).mean().backward()
    local = [p.grad.clone() for p in model.parameters()]
    w.reduce_gradients('sap')
    w.reduce_gradients('sap')   # second call reuses the cached bucket plan; averaging an average is a no-op
    got = [p.grad.clone() for p in model.parameters()]
    # numpy payloads are pickled by value (torch tensors travel as shared-memory fds that die with this process)
    q.put((rank, [t.numpy() for t in local], [t.numpy() for t in got], [p.grad is None for p in unused.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_allreduce_mean():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_buckets, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    mean = [(torch.from_numpy(a) + torch.from_numpy(b)) / 2 for a, b in zip(res[0][1], res[1][1])]
    for r in range(world):
        for g, m in zip(res[r][2], mean):
            assert torch.allclose(torch.from_numpy(g), m, atol=1e-6)
        assert all(res[r][3])


def _worker_arena(rank, world, port, q):
    _init(rank, world, port)
    from vln_goat_amd import dp
    torch.manual_seed(0)
    shared = torch.nn.Linear(16, 32)
    head_a, head_b = torch.nn.Linear(32, 8), torch.nn.Linear(32, 4)
    unused = torch.nn.Linear(3, 3)
    holder = torch.nn.ModuleList([shared, head_a, head_b, unused])
    w = dp.GoatDataParallel(holder)
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 16)

    def run(task):
        h = shared(x)
        return (head_a(h) if task == 'sap' else head_b(h)).pow(2).mean()
    for task in ('sap', 'mlm'):                  # discovery: ordinary backward per task
        for p in holder.parameters():
            p.grad = None
        run(task).backward()
        w.record_usage(task)
    for p in holder.parameters():
        p.grad = None
    arena = w.build_arena(bucket_bytes=256)      # tiny buckets -> several all-reduces per range
    out = {}
    for task in ('sap', 'mlm', 'sap'):
        arena.bind(task)                         # .grad = arena view for this task's parameters, None for the others
        arena.zero(task)
        run(task).backward()
        local = {n: p.grad.clone().numpy() for n, p in holder.named_parameters() if p.grad is not None}
        w.reduce_gradients(task)
        got = {n: (p.grad.clone().numpy() if p.grad is not None else None) for n, p in holder.named_parameters()}
        out[task] = (local, got)
    q.put((rank, out, [len(arena.ranges(t)) for t in ('sap', 'mlm')], arena.numel))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_arena_allreduce_and_task_binding():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_arena, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    import numpy as np
    for task, used_head, other_head in (('sap', '1.', '2.'), ('mlm', '2.', '1.')):
        l0, g0 = res[0][1][task]
        l1, g1 = res[1][1][task]
        for n in g0:
            if n.startswith('3.') or n.startswith(other_head):
                assert g0[n] is None and g1[n] is None, (task, n)      # unused / other task's parameters: .grad None
                continue
            want = (l0[n] + l1[n]) / 2
            assert np.allclose(g0[n], want, atol=1e-6) and np.allclose(g1[n], want, atol=1e-6), (task, n)
    assert res[0][2] == res[1][2] and max(res[0][2]) <= 2       # a task's slices form at most 2 contiguous ranges here


def _cfp_inputs(n, h=32):
    g = torch.Generator().manual_seed(7)
    return [torch.tanh(torch.randn(n, h, generator=g)) for _ in range(4)]


def _worker_cfp(rank, world, port, q):
    _init(rank, world, port)
    from vln_goat_amd import dp
    from vln_goat_amd.pretrain_model import cfp_losses
    B = 3
    full = _cfp_inputs(world * B)
    loc = [t[rank * B:(rank + 1) * B].clone().requires_grad_(True) for t in full]
    loss = cfp_losses(loc[0], loc[1], loc[2], loc[3], 0.7, dp.CfpGather())
    loss.mean().backward()
    grads = [t.grad.clone() for t in loc]
    for g in grads:                      # what GoatDataParallel.reduce_gradients does for parameters
        dist.all_reduce(g)
        g /= world
    q.put((rank, loss.detach().numpy(), [t.grad.numpy() for t in loc]))
    dist.barrier()
    dist.destroy_process_group()


def test_cfp_gather_equals_single_process_on_concatenated_batch():
    from vln_goat_amd.pretrain_model import cfp_losses
    world, port, B = 2, _free_port(), 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_cfp, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    full = [t.clone().requires_grad_(True) for t in _cfp_inputs(world * B)]
    ref = cfp_losses(full[0], full[1], full[2], full[3], 0.7, None)     # reference formula, one process
    ref.mean().backward()
    got = torch.cat([torch.from_numpy(r[1]) for r in res])
    assert torch.allclose(got, ref.detach(), atol=1e-5)
    # d(mean over the global batch)/d(local embeddings) = (1/W) * local grads of the local-mean loss
    for k in range(4):
        g = torch.cat([torch.from_numpy(r[2][k]) for r in res]) / world
        assert torch.allclose(g, full[k].grad, atol=1e-6), k


def test_world_size_one_is_the_reference_formula():
    from vln_goat_amd import dp
    from vln_goat_amd.pretrain_model import cfp_losses
    x = _cfp_inputs(4)
    a = cfp_losses(x[0], x[1], x[2], x[3], 1.0, None)
    b = cfp_losses(x[0], x[1], x[2], x[3], 1.0, dp.CfpGather())
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------- bf16 wire format of the arena exchange
def _worker_wire(rank, world, port, q):
    _init(rank, world, port)
    from vln_goat_amd import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 7))
    torch.manual_seed(200 + rank)
    x = torch.randn(9, 24)
    out = {}
    for name, wire in (('f32', None), ('bf16', torch.bfloat16)):
        w = dp.GoatDataParallel(net, wire_dtype=wire)
        arena = w.build_arena(bucket_bytes=1000)                 # chunks of 250 floats: not a multiple of the world size
        arena.zero('nav')
        net(x).pow(2).mean().backward()
        w.reduce_gradients('nav')
        out[name] = arena.flat.clone().numpy()
        arena.detach()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_exchange_matches_the_float32_exchange():
    """GradArena(wire_dtype=bfloat16): all-to-all of bf16 shards, float32 accumulation on receipt, all-gather of the averaged shards —
    within bf16 rounding (2e-2 of the tensor's scale, north_star's bf16 bound) of the float32 all-reduce, and BIT-identical on both ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_wire, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    import numpy as np
    a0, a1 = res[0][1], res[1][1]
    assert np.array_equal(a0['bf16'], a1['bf16'])                # ranks stay in lockstep
    assert np.allclose(a0['f32'], a1['f32'], atol=1e-7)
    scale = np.abs(a0['f32']).max()
    assert scale > 0
    err = np.abs(a0['bf16'] - a0['f32']).max()
    assert err <= 2e-2 * scale, (err, scale)
    assert err > 0                                               # (the reduced-precision path did run)
    rel = np.abs(a0['bf16'] - a0['f32'])[np.abs(a0['f32']) > 0.05 * scale] / np.abs(a0['f32'])[np.abs(a0['f32']) > 0.05 * scale]
    assert rel.max() < 2 ** -7                                   # two roundings to 8 mantissa bits


# ---------------------------------------------------------------------------------------------- fine-tuning iteration (BASELINE configs[3])
class _ToyVLNBert(torch.nn.Module):
    """the VLNBert call contract of M/models/model.py:12-50: forward(mode, batch), several calls per rollout, BPTT through a carried state"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.txt = torch.nn.Linear(6, 8)
        self.pano = torch.nn.Linear(5, 8)
        self.nav = torch.nn.Linear(16, 4)
        self.mem = torch.nn.Linear(4, 8)
        self.never = torch.nn.Linear(2, 2)          # a parameter no mode uses (find_unused_parameters=True upstream)

    def forward(self, mode, batch):
        if mode == 'language':
            return torch.tanh(self.txt(batch['txt']))
        if mode == 'panorama':
            return torch.tanh(self.pano(batch['views']))
        h = torch.cat([batch['txt_embeds'], batch['pano'] + (0 if batch['mem'] is None else self.mem(batch['mem']))], 1)
        return self.nav(h)


def _toy_iteration(model, data, ml_weight=0.2):
    """M/r2r/agent.py:414-437 (dagger): a teacher rollout weighted ml_weight and a second rollout weighted 1, losses summed over samples
    and steps, divided by the batch size (agent.py:708), ONE backward."""
    B = data['txt'].shape[0]
    total = 0.0
    for wgt, tgt in ((ml_weight, data['tgt_a']), (1.0, data['tgt_b'])):
        txt = model('language', {'txt': data['txt']})
        mem, loss = None, 0.0
        for t in range(data['views'].shape[1]):
            pano = model('panorama', {'views': data['views'][:, t]})
            logits = model('navigation', {'txt_embeds': txt, 'pano': pano, 'mem': mem})
            mem = logits
            loss = loss + torch.nn.functional.cross_entropy(logits, tgt[:, t], reduction='sum')
        total = total + loss * wgt / B
    return total


def _toy_data(seed, n):
    g = torch.Generator().manual_seed(seed)
    return {'txt': torch.randn(n, 6, generator=g), 'views': torch.randn(n, 3, 5, generator=g),
            'tgt_a': torch.randint(0, 4, (n, 3), generator=g), 'tgt_b': torch.randint(0, 4, (n, 3), generator=g)}


def _worker_finetune(rank, world, port, q):
    _init(rank, world, port)
    from vln_goat_amd import dp
    model = _ToyVLNBert()
    if rank == 1:                                    # rank 1 starts from other weights: the wrapper must broadcast rank 0's
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    critic = torch.nn.Linear(8, 1)
    w, wc = dp.wrap_finetune_models(model, critic)
    data = _toy_data(300 + rank, 3)
    loss = _toy_iteration(w, data)
    loss.backward()
    w.record_usage('nav')
    for p in model.parameters():
        p.grad = None
    arena = w.build_arena()
    losses = []
    for _ in range(2):                               # two iterations on the arena: the second must not see the first's gradients
        arena.zero('nav')
        loss = _toy_iteration(w, data)
        loss.backward()
        dp.reduce_finetune_gradients((w, wc))
        losses.append(float(loss))
    q.put((rank, losses, {n: (None if p.grad is None else p.grad.clone().numpy()) for n, p in model.named_parameters()},
           [p.grad is None for p in critic.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_finetune_iteration_two_ranks_equals_single_process_on_the_concatenated_batch():
    """dp.wrap_finetune_models / reduce_finetune_gradients: vln_bert + critic as M/r2r/agent_base.py:100-102 wraps them; the iteration of
    M/r2r/agent.py:414-445 (two rollouts, one backward) on 2 ranks x 3 samples gives the gradients of one process on the 6 samples."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_finetune, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    import numpy as np
    model = _ToyVLNBert()
    d0, d1 = _toy_data(300, 3), _toy_data(301, 3)
    cat = {k: torch.cat([d0[k], d1[k]]) for k in d0}
    loss = _toy_iteration(model, cat)
    loss.backward()
    assert abs((res[0][1][1] + res[1][1][1]) / 2 - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert res[0][1][0] == res[0][1][1]
    for n, p in model.named_parameters():
        g0, g1 = res[0][2][n], res[1][2][n]
        if p.grad is None:
            assert g0 is None and g1 is None, n
            continue
        assert np.allclose(g0, p.grad.numpy(), rtol=1e-5, atol=1e-6), n
        assert np.array_equal(g0, g1), n
    assert all(res[0][3]) and all(res[1][3])         # the critic took no part: no gradient, no exchange
