"""Phase timing of the shared-dS attention backward kernel (needs attention2.hip built with -DGOAT_ATTN_TIMING=1; GOAT_HIP_LIB):
s_memtime stamps of wave 0 of every block, written to the (otherwise unused) dbias buffer: staging -> barrier -> phase 1 (S, dP, dS,
dK / dV) -> barrier -> dK / dV stores -> phase 2 (dQ) + store.   VARIANT_SRC=attention2 scripts/build_variants.sh attnt:"-DGOAT_ATTN_TIMING=1"
GOAT_HIP_LIB=vln-goat_amd/csrc/ab/libgoat_attnt.so python scripts/attn_bwd_timing.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import _lib
torch.cuda.set_device(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
NH, H = 12, 768
for B, Lq, Lk in ((48, 80, 80), (1, 80, 80), (240, 36, 36), (48, 37, 80), (48, 80, 37)):
    for p in (0.1, 0.0):
        Lm = max(Lq, Lk)
        qkv = torch.randn(B, Lm, 3 * H, device='cuda').to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        o = torch.randn(B, Lq, H, device='cuda').to(torch.bfloat16)
        do = torch.randn(B, Lq, H, device='cuda').to(torch.bfloat16)
        lse = torch.zeros(B * NH * Lq, device='cuda')
        stamps = torch.zeros(B * NH * 8, device='cuda', dtype=torch.int32)
        km = torch.zeros(B, Lk, device='cuda')
        rs, bs = 3 * H, Lm * 3 * H
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + H * 2, qkv.data_ptr() + 2 * H * 2
        dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + H * 2, dqkv.data_ptr() + 2 * H * 2
        for _ in range(3):
            rc = L.goat_attn_bwd(st, 1, q, rs, bs, k, rs, bs, v, rs, bs, o.data_ptr(), H, Lq * H, do.data_ptr(), H, Lq * H, dq, rs, bs, dk, rs, bs, dv, rs, bs,
                                 km.data_ptr(), None, lse.data_ptr(), stamps.data_ptr(), B, NH, Lq, Lk, 0.125, p, 1, 0, None)
            assert rc == 0
        torch.cuda.synchronize()
        t = stamps.view(B * NH, 8).cpu().numpy().astype('int64')
        d = (t[:, 1:7] - t[:, 0:6]) & 0xFFFFFFFF
        span = (t[:, 6].max() - t[:, 0].min()) & 0xFFFFFFFF
        print('      staging split: loads issued after %.0f cycles, landed + LDS writes + D after another %.0f' % (((t[:, 7] - t[:, 0]) & 0xFFFFFFFF).mean(), ((t[:, 1] - t[:, 7]) & 0xFFFFFFFF).mean()))
        names = ['staging', 'barrier', 'phase1', 'barrier', 'dKdV store', 'phase2+dQ store']
        print('B=%3d Lq=%2d Lk=%2d p=%.1f | mean cycles per block (wave 0): %s | block total %.0f | first start -> last end %d cycles (%.1f us at 2.4 GHz... s_memtime is a 100 MHz counter: see the README)' % (
            B, Lq, Lk, p, '  '.join('%s %.0f' % (n, d[:, i].mean()) for i, n in enumerate(names)), ((t[:, 6] - t[:, 0]) & 0xFFFFFFFF).mean(), span, span / 2400.0), flush=True)
