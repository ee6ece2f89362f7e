"""Launch the main GOAT GEMM shapes REPS times each, in a fixed order, for a rocprofv3 --pmc pass
(scripts/gemm_pmc_report.py joins the counter CSV with this order).  Operands are flushed from the caches
before every launch (a 320 MB fill), as inside a training step."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
REPS = 3
SHAPES = [  # (ta, tb, M, N, Kc, epi, split, f32out)
    (0, 0, 3840, 3072, 768, 1, 1, 0), (0, 0, 3840, 768, 3072, 0, 1, 0), (0, 0, 3840, 2304, 768, 0, 1, 0), (0, 0, 3840, 768, 768, 0, 1, 0),
    (0, 1, 3840, 3072, 768, 3, 1, 0), (0, 1, 3840, 768, 3072, 0, 1, 0), (0, 1, 3840, 768, 2304, 0, 1, 0), (0, 1, 3840, 768, 768, 0, 1, 0),
    (1, 1, 3072, 768, 3840, 0, 2, 1), (1, 1, 768, 3072, 3840, 0, 1, 1), (1, 1, 2304, 768, 3840, 0, 2, 1), (1, 1, 768, 768, 3840, 0, 3, 1),
    (0, 0, 8640, 3072, 768, 1, 1, 0), (0, 0, 8640, 768, 3072, 0, 1, 0), (1, 1, 3072, 768, 8640, 0, 3, 1),
]
flush = torch.empty(320 << 20, dtype=torch.uint8, device='cuda')
order = []
for ta, tb, M, N, Kc, epi, split, f32 in SHAPES:
    a = torch.randn((Kc, M) if ta else (M, Kc), device='cuda').to(torch.bfloat16)
    b = (torch.randn((Kc, N) if tb else (N, Kc), device='cuda') * 0.1).to(torch.bfloat16)
    out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16)
    aux = torch.randn(M, N, device='cuda').to(torch.bfloat16) if epi else None
    key = (bool(ta), bool(tb), M, N, Kc, epi, bool(f32), split, False)
    cfg = hipops._TUNED.get(key)
    if cfg is None:
        cands = [v for k, v in hipops._TUNED.items() if k[:7] == key[:7]]
        cfg = cands[0] if cands else (128, 2, split)
    bm, ns, sp = cfg
    for _ in range(REPS):
        flush.zero_()
        s_ = L.goat_gemm_bf16(st, ta, tb, hipops._dt(out), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N,
                             M, N, Kc, None, epi, aux.data_ptr() if aux is not None else None, N if aux is not None else 0, sp, bm, ns, None)
        assert s_ == 0, s_
    torch.cuda.synchronize()
    rd = (M * Kc + N * Kc) * 2 + (M * N * 2 if epi in (3, 4) else 0)
    wr = M * N * (4 if f32 else 2) * (2 if epi in (1, 2) else 1)
    order.append({'shape': [ta, tb, M, N, Kc, epi, sp, bm, ns], 'reps': REPS, 'algo_read_bytes': rd, 'algo_write_bytes': wr,
                  'flops': 2.0 * M * N * Kc})
json.dump(order, open(os.environ.get('GEMM_PMC_ORDER', 'gpurun_out/gemm_pmc_order.json'), 'w'))
