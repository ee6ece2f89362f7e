"""Phase timing of the forward attention kernel (needs attention2.hip built with -DGOAT_ATTN_TIMING=1; GOAT_HIP_LIB):
cycles between the stamps of wave 0 of every block: stage-issue, barrier, S = K Q^T, softmax, dropout, P V, store."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import _lib
torch.cuda.set_device(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
NH, H = 12, 768
for B, Lq, Lk in ((48, 80, 80), (1, 80, 80), (240, 36, 36), (48, 22, 80)):
    for p in (0.1, 0.0):
        Lm = max(Lq, Lk)
        qkv = torch.randn(B, Lm, 3 * H, device='cuda').to(torch.bfloat16)
        o = torch.empty(B, Lq, H, device='cuda', dtype=torch.bfloat16)
        lse = torch.zeros(B * NH * Lq + B * NH * 8, device='cuda')
        km = torch.zeros(B, Lk, device='cuda')
        rs, bs = 3 * H, Lm * 3 * H
        for _ in range(3):
            rc = L.goat_attn_fwd(st, 1, qkv.data_ptr(), rs, bs, qkv.data_ptr() + H * 2, rs, bs, qkv.data_ptr() + 2 * H * 2, rs, bs, o.data_ptr(), H, Lq * H,
                                 km.data_ptr(), None, lse.data_ptr(), B, NH, Lq, Lk, 0.125, p, 1, 0, None)
            assert rc == 0
        torch.cuda.synchronize()
        t = lse[B * NH * Lq:].view(torch.int32).view(B * NH, 8).cpu().numpy().astype('int64')
        d = (t[:, 1:] - t[:, :-1]) & 0xFFFFFFFF
        span = (t[:, 7].max() - t[:, 0].min()) & 0xFFFFFFFF
        names = ['stage', 'barrier', 'S', 'softmax', 'dropout', 'PV', 'store']
        print('B=%3d Lq=%2d Lk=%2d p=%.1f | per-block mean cycles: %s | block total %.0f | first start -> last end %d cycles' % (
            B, Lq, Lk, p, '  '.join('%s %.0f' % (n, d[:, i].mean()) for i, n in enumerate(names)), (t[:, 7] - t[:, 0]).mean(), span))
