"""Sweep (bm, split_k) of the TN (wgrad) GEMM on GOAT's weight-gradient shapes; prints the best per shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib
torch.cuda.set_device(0)
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
SHAPES = [(768, 768, 3840), (3072, 768, 3840), (768, 3072, 3840), (2304, 768, 3840), (1536, 768, 3840),
          (768, 768, 8640), (3072, 768, 8640), (768, 3072, 8640), (2304, 768, 8640),
          (768, 768, 1776), (3072, 768, 1776), (2304, 768, 1776), (768, 768, 1056), (3072, 768, 1056)]
for M, N, Kc in SHAPES:
    a = torch.randn(Kc, M, device='cuda').to(torch.bfloat16)
    b = torch.randn(Kc, N, device='cuda').to(torch.bfloat16)
    res = []
    for bm in (64, 128):
        for split in (1, 2, 3, 4, 6, 8, 12):
            if split > (Kc + 63) // 64:
                continue
            out = torch.zeros(M, N, device='cuda')
            cs = torch.zeros(M, device='cuda')
            def run():
                s_ = L.goat_gemm_bf16(st, 1, 1, 0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N,
                                      M, N, Kc, None, 0, None, 0, split, bm, 2, cs.data_ptr())
                assert s_ == 0
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 1e3 / 30, bm, split))
    res.sort()
    print('out[%4d,%4d] Kc=%5d  best: %s' % (M, N, Kc, '  '.join('bm%d/s%d %.1fus(%.0fTF)' % (bm, sp, t, 2.0*M*N*Kc/t/1e6) for t, bm, sp in res[:4])))
