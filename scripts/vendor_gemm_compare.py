"""How far is goat_gemm_bf16 from the vendor library (torch.mm -> hipBLASLt) on the GOAT shapes?  Information only:
the product path never calls the vendor GEMM.  Operands rotate through 6 buffers (cold caches, as inside a step)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops

torch.cuda.set_device(0)
ROT = 6
def bench(fn, n=40):
    for _ in range(6):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
hipops.AUTOTUNE = True
print('%-28s %10s %10s %8s' % ('shape (M,N,K) y = x W^T', 'ours us', 'vendor us', 'ratio'))
for M, N, K in ((3840, 768, 768), (3840, 2304, 768), (3840, 3072, 768), (3840, 768, 3072), (8640, 3072, 768), (8640, 768, 3072),
                (1776, 768, 768), (1056, 768, 768), (8192, 8192, 8192)):
    xs = [torch.randn(M, K, device='cuda').to(torch.bfloat16) for _ in range(ROT)]
    ws = [(torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
    os_ = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
    i = [0]
    def ours():
        k = i[0] % ROT; i[0] += 1
        hipops.gemm(xs[k], ws[k], os_[k])
    def vendor():
        k = i[0] % ROT; i[0] += 1
        torch.mm(xs[k], ws[k].t(), out=os_[k])
    ours(); torch.cuda.synchronize()
    to, tv = bench(ours), bench(vendor)
    print('%-28s %10.1f %10.1f %8.2f   (%.0f vs %.0f TFLOP/s)' % (str((M, N, K)), to, tv, to / tv, 2.0 * M * N * K / to / 1e6, 2.0 * M * N * K / tv / 1e6))
