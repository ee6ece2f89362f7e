"""Where do the operands of a GOAT GEMM have to be for the kernel to run at its micro-benchmark speed?  Times goat_gemm_bf16 on the
text-layer shapes with (a) both operands re-used every launch (L2 / Infinity-Cache warm), (b) the WEIGHT rotating through enough
copies to be HBM-cold (what a training step sees: 400 MB of weights per step against a 256 MB Infinity Cache), (c) the ACTIVATION
rotating, (d) both.  (Round 3 also tried a software prefetch: a small kernel on a parallel graph branch that reads the NEXT launch's
weight while the current GEMM runs.  The fork / join of the branch costs more than the cold start it removes — +5...+7 us per launch on
the text shapes, profiles/round3_gemm_cold_operands.txt — so the entry point was removed again.)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
L = _lib.lib()


def T(bm, bn):
    return bm | (bn << 16)


CASES = [  # name, ta, tb, M, N, K, epi, tile, nstage
    ('qkv fwd', 0, 0, 3840, 2304, 768, 0, T(192, 192), 2), ('ffn-up fwd', 0, 0, 3840, 3072, 768, 1, T(192, 256), 2),
    ('ffn-down fwd', 0, 0, 3840, 768, 3072, 0, 96, 4), ('out-proj fwd', 0, 0, 3840, 768, 768, 0, 96, 4),
    ('ffn-down dgrad', 0, 1, 3840, 3072, 768, 3, T(192, 256), 2), ('ffn-up dgrad', 0, 1, 3840, 768, 3072, 0, 96, 4),
    ('pano ffn-up', 0, 0, 8640, 3072, 768, 1, T(256, 256), 2), ('pano ffn-down', 0, 0, 8640, 768, 3072, 0, T(128, 256), 3),
]
NC = 72          # copies of the rotating operand (72 x 4.7 MB = 340 MB of weights)
main = torch.cuda.current_stream()


def bench(fn, n=NC * 2):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, ta, tb, M, N, K, epi, tile, ns in CASES:
    na = NC if M * K * 2 * NC < 3e9 else 24
    As = [torch.randn((M, K), device='cuda').to(torch.bfloat16) for _ in range(na)]
    Bs = [(torch.randn((K, N) if tb else (N, K), device='cuda') * 0.05).to(torch.bfloat16) for _ in range(NC)]
    out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16) if epi else None

    def gemm(a, b, stream=None):
        st = (stream or main).cuda_stream
        rc = L.goat_gemm_bf16(st, ta, tb, 1, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N, M, N, K, None, epi,
                              aux.data_ptr() if epi else None, N if epi else 0, 1, tile, ns, None)
        assert rc == 0, rc
    warm = bench(lambda i: gemm(As[0], Bs[0]))
    wcold = bench(lambda i: gemm(As[0], Bs[i % NC]))
    acold = bench(lambda i: gemm(As[i % na], Bs[0]))
    both = bench(lambda i: gemm(As[i % na], Bs[i % NC]))

    print('%-16s M=%5d N=%5d K=%5d | warm %6.2f | W cold %6.2f | A cold %6.2f | both cold %6.2f us   (%.0f -> %.0f TFLOP/s)'
          % (name, M, N, K, warm, wcold, acold, both, 2.0 * M * N * K / warm / 1e6, 2.0 * M * N * K / both / 1e6))
