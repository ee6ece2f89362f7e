"""Micro-benchmark of goat_gemm_bf16 / goat_gemm_nt on the GOAT shapes (random data, HIP events, 50 reps)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
SHAPES = [  # (ta, tb, M, N, Kc, epi, split)
    (0, 0, 3840, 768, 768, 0, 1), (0, 0, 3840, 2304, 768, 0, 1), (0, 0, 3840, 3072, 768, 1, 1), (0, 0, 3840, 768, 3072, 0, 1),
    (0, 1, 3840, 768, 768, 0, 1), (0, 1, 3840, 3072, 768, 3, 1), (0, 1, 3840, 768, 3072, 0, 1),
    (1, 1, 768, 768, 3840, 0, 11), (1, 1, 3072, 768, 3840, 0, 3), (1, 1, 768, 3072, 3840, 0, 3),
    (0, 0, 8640, 2304, 768, 0, 1), (0, 0, 8640, 768, 3072, 0, 1), (0, 0, 1056, 768, 768, 0, 1), (0, 0, 8192, 8192, 8192, 0, 1),
]
bms = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [128, 64]
stages = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4]
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
ROT = int(os.environ.get('ROT', '1'))   # >1: rotate over ROT operand sets (cold caches, as inside a training step)
for ta, tb, M, N, Kc, epi, split in SHAPES:
    if ROT > 1 and M * N * Kc > 2e11:
        continue
    As = [torch.randn((Kc, M) if ta else (M, Kc), device='cuda').to(torch.bfloat16) for _ in range(ROT)]
    Bs = [(torch.randn((Kc, N) if tb else (N, Kc), device='cuda') * 0.1).to(torch.bfloat16) for _ in range(ROT)]
    Os = [torch.zeros(M, N, device='cuda', dtype=torch.float32 if split > 1 else torch.bfloat16) for _ in range(ROT)]
    a, b, out = As[0], Bs[0], Os[0]
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16) if epi else None
    bias = torch.zeros(N, device='cuda') if (split == 1) else None
    res = []
    for bm, ns in [(b_, n_) for b_ in bms for n_ in stages]:
        cnt = [0]
        def run():
            i = cnt[0] % ROT
            cnt[0] += 1
            a, b, out = As[i], Bs[i], Os[i]
            s_ = L.goat_gemm_bf16(st, ta, tb, hipops._dt(out), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                 out.data_ptr(), out.stride(0), M, N, Kc, bias.data_ptr() if bias is not None else None, epi,
                                 aux.data_ptr() if aux is not None else None, N if aux is not None else 0, split, bm, ns if 'ns' in dir() else 2, None)
            assert s_ == 0, s_
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        res.append('bm%d/s%d %7.2f us %6.1f TF' % (bm, ns, us, 2.0 * M * N * Kc / us / 1e6))
    print('t%d%d M=%5d N=%5d Kc=%5d epi=%d split=%2d | %s' % (ta, tb, M, N, Kc, epi, split, ' | '.join(res)))
