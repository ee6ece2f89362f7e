"""Which RCCL collectives survive hipGraph capture on this stack (one-rank group)?  python scripts/rccl_capture_probe.py <name>"""
import os, sys
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29535')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
name = sys.argv[1]
x = torch.arange(1 << 16, dtype=torch.float32, device='cuda').to(torch.bfloat16)
y = torch.empty_like(x)
comm = torch.cuda.Stream()
def op():
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        if name == 'all_reduce': dist.all_reduce(x)
        elif name == 'all_to_all_single': dist.all_to_all_single(y, x)
        elif name == 'all_gather_into_tensor': dist.all_gather_into_tensor(y, x)
        elif name == 'reduce_scatter_tensor': dist.reduce_scatter_tensor(y, x)
        elif name == 'broadcast': dist.broadcast(x, 0)
        elif name == 'all_gather_list': dist.all_gather([y], x)
    torch.cuda.current_stream().wait_stream(comm)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    op()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    op()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print('CAPTURE_OK', name, flush=True)
dist.destroy_process_group()
