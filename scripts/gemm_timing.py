"""Where do a K-tile's cycles go?  Needs a libgoat_hip.so built with -DGOAT_G2_TIMING=1 (GOAT_HIP_LIB): every wave sums, over
its K-tiles, the cycles (s_memtime) spent waiting for its DMA pieces / last fragments, at the barrier, and in the compute
phase between barriers.    GOAT_HIP_LIB=.../ab/libgoat_timing.so python scripts/gemm_timing.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops

torch.cuda.set_device(0)
T = hipops.tile
CASES = [(0, 0, 3840, 3072, 768, T(192, 256), 2), (0, 0, 3840, 3072, 768, 128, 0x102), (0, 0, 3840, 768, 3072, 128, 0x104),
         # weight-gradient layout (dW = dY^T X): a lone 36-tile problem, and problems that fill the chip
         (1, 1, 3072, 768, 3840, T(256, 256), 2), (1, 1, 3072, 3072, 3840, T(256, 256), 2), (1, 1, 6144, 3072, 3840, T(256, 256), 2),
         (1, 1, 6144, 3072, 3840, 256, 3), (1, 1, 6144, 3072, 3840, 128, 0x102), (1, 1, 6144, 3072, 3840, T(128, 256), 3),
         (0, 0, 8192, 8192, 8192, T(256, 256), 2)]
for ta, tb, M, N, K, bm, ns in CASES:
    a = torch.randn((K, M) if ta else (M, K), device='cuda').to(torch.bfloat16)
    b = (torch.randn((K, N) if tb else (N, K), device='cuda') * 0.05).to(torch.bfloat16)
    o = torch.empty(M, N, device='cuda', dtype=torch.float32 if ta else torch.bfloat16)
    rows, cols = bm & 0xFFFF, (bm >> 16) or 128
    nblk = ((M + rows - 1) // rows) * ((N + cols - 1) // cols)
    nw = 4 if (cols == 128 and rows <= 128 and not (ns & 0x100)) else 8
    aux = torch.zeros(nblk * nw * 4 + 64, dtype=torch.int32, device='cuda')
    for _ in range(3):
        hipops._launch_gemm_bf16(a, b, o, bool(ta), bool(tb), M, N, K, None, 0, aux, 1, bm, ns, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hipops._launch_gemm_bf16(a, b, o, bool(ta), bool(tb), M, N, K, None, 0, aux, 1, bm, ns, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    r = aux[:nblk * nw * 4].view(nblk, nw, 4).float()
    nkt = (K + 63) // 64
    w, br, cp, tot = [r[:, :, i] for i in range(4)]
    print('t%d%d %dx%dx%d tile %s s%d: %d blocks x %d waves, %d K-tiles, %.1f us (%.0f TF) | per K-tile: wait %.0f  barrier %.0f  compute %.0f  (sum %.0f) cycles | '
          'main loop %.0f cyc (min %.0f max %.0f) | ideal MFMA cycles per K-tile %d' % (
              ta, tb, M, N, K, hipops.tile_name(bm), ns & 0xFF, nblk, nw, nkt, us, 2.0 * M * N * K / us / 1e6, w.mean() / nkt, br.mean() / nkt, cp.mean() / nkt,
              (w + br + cp).mean() / nkt, tot.mean(), tot.min(), tot.max(), rows * cols * 64 * 2 // (4 * 1024)))
    print('   block 0 waves: wait', [int(x / nkt) for x in w[0].tolist()], 'barrier', [int(x / nkt) for x in br[0].tolist()], 'compute', [int(x / nkt) for x in cp[0].tolist()])
