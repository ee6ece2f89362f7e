"""One GEMM shape, hot operands, REPS launches — the workload for `rocprofv3 --pmc <SQ/TA/TCP counters>` passes that ask
where the main loop of goat_gemm_bf16 stalls (scripts/pmc_summary.py prints the per-launch averages).

    python scripts/gemm_stall_pmc.py M N K [ta tb bm nstage]
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

M, N, K = (int(v) for v in sys.argv[1:4])
ta, tb, bm, ns = (int(v) for v in sys.argv[4:8]) if len(sys.argv) > 7 else (0, 0, 128, 2)
torch.cuda.set_device(0)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
a = torch.randn((K, M) if ta else (M, K), device='cuda').to(torch.bfloat16)
b = (torch.randn((K, N) if tb else (N, K), device='cuda') * 0.1).to(torch.bfloat16)
out = torch.zeros(M, N, device='cuda', dtype=torch.float32 if (ta and tb) else torch.bfloat16)     # wgrad layout: f32 result
for _ in range(5):
    rc = L.goat_gemm_bf16(st, ta, tb, hipops._dt(out), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N,
                          M, N, K, None, 0, None, 0, 1, bm, ns, None)
    assert rc == 0, rc
torch.cuda.synchronize()
