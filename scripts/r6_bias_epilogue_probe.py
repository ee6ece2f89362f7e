"""round 6: what the bias vector costs at the head of the forward GEMM epilogues (one 16-byte load per column group instead of four 4-byte ones).
Forward shapes of the headline batch on their tuned configurations, cold random operands, with and without a bias; run once per library (GOAT_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops as H, tuning
H.load_tuned()
dev = 'cuda'
torch.manual_seed(0)
for (M, N, K, epi) in [(3840, 3072, 768, H.EPI_GELU), (3840, 3072, 768, H.EPI_NONE), (3840, 2304, 768, H.EPI_NONE), (3840, 768, 768, H.EPI_NONE), (3840, 768, 3072, H.EPI_NONE),
                       (8640, 3072, 768, H.EPI_GELU), (8640, 768, 3072, H.EPI_NONE), (20480, 3072, 768, H.EPI_GELU), (20480, 768, 768, H.EPI_NONE)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 0.1).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.empty_like(out) if epi != H.EPI_NONE else None
    bias = torch.rand(N, device=dev)
    res = []
    for b in (None, bias):
        key = (False, False, M, N, K, epi, False, 1, b is not None)
        cfg = tuning._TUNED.get(key) or tuning._TUNED.get((False, False, M, N, K, epi, False, 1, True)) or (H._heuristic_cfg(False, False, M, N, K, 1) + (1,))
        bm, ns, _ = cfg
        fn = lambda: H._launch_gemm_bf16(a, w, out, False, False, M, N, K, b, epi, aux, 1, bm, ns, None)
        fn(); torch.cuda.synchronize()
        res.append((H._time_cfg(fn, reps=11) * 1e3, H.tile_name(bm), H.stage_name(ns)))
    print('%6d x %5d x %5d epi %d  %-8s %-5s  no bias %7.2f us   bias %7.2f us   (+%.2f us)' % (M, N, K, epi, res[1][1], res[1][2], res[0][0], res[1][0], res[1][0] - res[0][0]), flush=True)
