"""Fixed vs per-K-tile cost of every tile configuration: y = x W^T on M x N with K swept (HIP events, warm and cold operands).
    python scripts/gemm_ksweep2.py [M N]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops

torch.cuda.set_device(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
KS = [256, 768, 1536, 3072, 6144]
T = hipops.tile
CFGS = [(128, 0x102), (128, 0x103), (256, 2), (256, 3), (T(128, 256), 2), (T(128, 256), 3), (T(192, 256), 2), (T(256, 192), 2), (T(256, 256), 2)]


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


for rot in (1, 6):
    print('--- M=%d N=%d  %s operands; us per launch at K = %s; slope = us per 64-deep K-tile (K 768 -> 6144), TF at that slope' % (
        M, N, 'warm (same buffers)' if rot == 1 else 'cold (rotating over 6 sets)', KS))
    for bm, ns in CFGS:
        ts = []
        for K in KS:
            As = [torch.randn(M, K, device='cuda').to(torch.bfloat16) for _ in range(rot)]
            Bs = [(torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(rot)]
            Os = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(rot)]
            i = [0]

            def run():
                k = i[0] % rot
                i[0] += 1
                hipops._launch_gemm_bf16(As[k], Bs[k], Os[k], False, False, M, N, K, None, 0, None, 1, bm, ns, None)
            ts.append(bench(run))
        slope = (ts[4] - ts[1]) / ((KS[4] - KS[1]) / 64)
        print('%-8s s%d | %s | slope %.3f us/ktile = %5.0f TF | fixed(K->0) %.1f us' % (
            hipops.tile_name(bm), ns & 0xFF, ' '.join('%7.1f' % t for t in ts), slope, 2.0 * M * N * 64 / slope / 1e6, ts[1] - slope * KS[1] / 64), flush=True)
