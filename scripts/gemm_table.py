"""Per-shape GEMM timing table of one eager mlm+sap+cfp cycle (HIP events around every goat_gemm_nt launch)."""
import sys, os, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vln_goat_amd import hipops

class A: pass
args = A(); args.batch = int(sys.argv[1]) if len(sys.argv) > 1 else 48; args.dtype = 'bf16'; args.layers = '6,3,2'
torch.cuda.set_device(0)
cfg, model, batch, gb, _static = bench.build(args, 0)
hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')
hipops.AUTOTUNE = True
for rep in range(3):
    hipops.PROFILE = [] if rep == 2 else None
    for task in bench.TASKS:
        for p in model.parameters():
            p.grad = None
        model(gb, task, compute_loss=True).mean().backward()
    torch.cuda.synchronize()
recs = hipops.PROFILE
hipops.PROFILE = None
# time every recorded launch individually with a cache flush in front (in-step conditions), 3 reps, median
from vln_goat_amd import _lib
L = _lib.lib()
flush = torch.empty(320 << 20, dtype=torch.uint8, device='cuda')
agg = collections.OrderedDict()
for rec in recs:
    fl, key = rec[2], rec[3]
    name, cargs, keep = rec[4]
    ts = []
    for _ in range(3):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = getattr(L, name)(torch.cuda.current_stream().cuda_stream, *cargs); e1.record(); e1.synchronize()
        assert rc == 0
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    t = ts[1] * 1e3
    a = agg.setdefault(key, [0, 0.0, fl])
    a[0] += 1; a[1] += t
tot = sum(a[1] for a in agg.values())
print('total gemm us per cycle: %.1f   launches %d' % (tot, len(recs)))
print('%-52s %5s %10s %9s %8s %6s' % ('(M,N,K,epi,split,dtype)', 'n', 'total_us', 'avg_us', 'TF/s', 'pct'))
for key, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-52s %5d %10.1f %9.2f %8.1f %6.2f' % (str(key), n, t, t / n, fl / (t / n * 1e-6) / 1e12, 100 * t / tot))
