"""Producer -> consumer timing of GEMM pairs as they occur in a step (the consumer reads what the producer just wrote), to judge
store policies (plain / nt / sc1 via GOAT_HIP_LIB variants) and tile choices in context rather than on isolated launches.
    python scripts/gemm_chain.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops
from vln_goat_amd._lib import EPI_GELU

torch.cuda.set_device(0)
T = hipops.tile
ROT = 4


def bench(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M in (3840, 8640):
    xs = [torch.randn(M, 768, device='cuda').to(torch.bfloat16) for _ in range(ROT)]
    w1 = [(torch.randn(3072, 768, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
    w2 = [(torch.randn(768, 3072, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
    wq = [(torch.randn(2304, 768, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
    hs = [torch.empty(M, 3072, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
    us = [torch.empty(M, 3072, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
    ys = [torch.empty(M, 768, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
    qs = [torch.empty(M, 2304, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
    b1, b2 = torch.zeros(3072, device='cuda'), torch.zeros(768, device='cuda')
    i = [0]
    for t1, s1 in ((T(192, 256), 2), (T(256, 256), 2), (128, 0x102), (T(128, 256), 3)):
        for t2, s2 in ((128, 0x104), (T(128, 256), 3), (256, 3)):
            def pair():
                k = i[0] % ROT
                i[0] += 1
                hipops._launch_gemm_bf16(xs[k], w1[k], hs[k], False, False, M, 3072, 768, b1, EPI_GELU, us[k], 1, t1, s1, None)
                hipops._launch_gemm_bf16(hs[k], w2[k], ys[k], False, False, M, 768, 3072, b2, 0, None, 1, t2, s2, None)

            def first():
                k = i[0] % ROT
                i[0] += 1
                hipops._launch_gemm_bf16(xs[k], w1[k], hs[k], False, False, M, 3072, 768, b1, EPI_GELU, us[k], 1, t1, s1, None)

            def second():
                k = i[0] % ROT
                i[0] += 1
                hipops._launch_gemm_bf16(hs[k], w2[k], ys[k], False, False, M, 768, 3072, b2, 0, None, 1, t2, s2, None)
            tp, ta, tb = bench(pair), bench(first), bench(second)
            print('M=%d FFN1(gelu+aux) %s/s%d -> FFN2 %s/s%d : pair %.1f us | alone %.1f + %.1f = %.1f' % (
                M, hipops.tile_name(t1), s1 & 0xFF, hipops.tile_name(t2), s2 & 0xFF, tp, ta, tb, ta + tb), flush=True)
