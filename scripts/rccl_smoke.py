"""Single-rank RCCL ('nccl' backend) exercise of every collective call the data-parallel engine makes, on cuda:0:
parameter broadcast, in-place all-reduce of arena ranges on the communication stream, the CFP all-gather and its
reduce-scatter backward, the task broadcast, barrier.  A 1-GPU box cannot run more ranks (RCCL refuses two ranks per
GPU); this only proves the API usage (dtypes, shapes, stream handling) is accepted by RCCL."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from vln_goat_amd import dp

m = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.Linear(512, 64)).cuda()
w = dp.GoatDataParallel(m)
x = torch.randn(8, 256, device='cuda')
m(x).pow(2).mean().backward()
w.record_usage('sap')
for p in m.parameters():
    p.grad = None
arena = w.build_arena(bucket_bytes=64 << 10)
arena.zero('sap')
m(x).pow(2).mean().backward()
ref = [p.grad.clone() for p in m.parameters()]
# force the multi-rank code path with world size 1 (mean over one rank = identity)
dp._world = lambda: 2 if os.environ.get('FAKE_WORLD') else 1
chunks = []
for a, b in arena.ranges('sap'):
    while a < b:
        e = min(b, a + arena.bucket_elems)
        chunks.append(arena.flat[a:e]); a = e
arena.comm_stream.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(arena.comm_stream):
    for c in chunks:
        arena._reduce_mean(c, 1)          # ReduceOp.AVG on RCCL (probed once), sum + divide elsewhere
torch.cuda.current_stream().wait_stream(arena.comm_stream)
print('ReduceOp.AVG accepted by this RCCL:', dp.GradArena._avg_ok)
torch.cuda.synchronize()
assert all(torch.equal(p.grad, r) for p, r in zip(m.parameters(), ref))
packed = torch.randn(4, 8, 768, device='cuda', requires_grad=True)
out = torch.empty((1 * 4, 8, 768), device='cuda')
dist.all_gather_into_tensor(out, packed.detach().contiguous())
assert torch.equal(out.view(1, 4, 8, 768)[0], packed.detach())
dx = torch.empty(4, 8, 768, device='cuda')
dist.reduce_scatter_tensor(dx, out)
assert torch.equal(dx, packed.detach())
t = torch.tensor([2], dtype=torch.int64, device='cuda')
dist.broadcast(t, 0)
for p in m.parameters():
    dist.broadcast(p.data, 0)
dist.barrier()
torch.cuda.synchronize()
# capture a collective into a hipGraph (round-2 candidate: all-reduce inside the captured step)
try:
    buf = torch.ones(1 << 20, device='cuda')
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(buf)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        dist.all_reduce(buf)
        buf.mul_(0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('graph-captured all_reduce: ok, buf[0] =', float(buf[0]))
except Exception as e:
    print('graph-captured all_reduce: FAILED', type(e).__name__, e)
dist.destroy_process_group()
print('RCCL_SMOKE_OK')
