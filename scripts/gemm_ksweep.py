"""Fixed cost vs per-k-tile cost of goat_gemm_bf16: time as a function of the contraction length at fixed M,N."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 3072)
ROT = int(os.environ.get('ROT', '6'))
for epi in (0, 1):
    for bm, ns in ((128, 2), (128, 3), (64, 2), (64, 3)):
        row = []
        for Kc in (64, 128, 256, 512, 768, 1536, 3072):
            As = [torch.randn(M, Kc, device='cuda').to(torch.bfloat16) for _ in range(ROT)]
            Bs = [(torch.randn(N, Kc, device='cuda') * 0.1).to(torch.bfloat16) for _ in range(ROT)]
            Os = [torch.zeros(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)]
            aux = [torch.zeros(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(ROT)] if epi else None
            bias = torch.zeros(N, device='cuda')
            cnt = [0]
            def run():
                i = cnt[0] % ROT; cnt[0] += 1
                s_ = L.goat_gemm_bf16(st, 0, 0, hipops._dt(Os[i]), As[i].data_ptr(), Kc, Bs[i].data_ptr(), Kc, Os[i].data_ptr(), N,
                                     M, N, Kc, bias.data_ptr(), epi, aux[i].data_ptr() if epi else None, N if epi else 0, 1, bm, ns, None)
                assert s_ == 0
            for _ in range(6):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                run()
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 1e3 / 60)
        print('M=%d N=%d epi=%d bm%d s%d | ' % (M, N, epi, bm, ns) + ' '.join('K%d:%6.1fus' % (k, t) for k, t in zip((64, 128, 256, 512, 768, 1536, 3072), row)))
