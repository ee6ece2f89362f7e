"""Every tile configuration of goat_gemm_bf16 on the GOAT shapes (all three operand layouts) next to the vendor library
(torch.mm -> hipBLASLt; information only).  Operands rotate through ROT buffer sets: cold caches, as inside a step.
    python scripts/gemm_tiles.py [fwd|dgrad|wgrad|all]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib
from vln_goat_amd._lib import EPI_GELU, EPI_MUL_DGELU, EPI_NONE

torch.cuda.set_device(0)
ROT = 6
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
FWD = [(3840, 768, 768, 0), (3840, 2304, 768, 0), (3840, 3072, 768, EPI_GELU), (3840, 768, 3072, 0),
       (8640, 768, 768, 0), (8640, 2304, 768, 0), (8640, 3072, 768, EPI_GELU), (8640, 768, 3072, 0),
       (1776, 768, 768, 0), (1776, 3072, 768, EPI_GELU), (1056, 768, 768, 0), (1056, 2304, 768, 0)]
DGRAD = [(3840, 768, 768, 0), (3840, 768, 2304, 0), (3840, 3072, 768, EPI_MUL_DGELU), (3840, 768, 3072, 0),
         (8640, 768, 2304, 0), (8640, 3072, 768, 0), (8640, 768, 3072, 0), (1776, 768, 3072, 0)]
WGRAD = [(768, 768, 3840), (2304, 768, 3840), (3072, 768, 3840), (768, 3072, 3840), (3072, 768, 8640), (768, 3072, 8640)]


def bench(fn, n=30):
    for _ in range(4):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def sweep(ta, tb, M, N, Kc, epi, f32out=False, splits=(1,)):
    As = [torch.randn((Kc, M) if ta else (M, Kc), device='cuda').to(torch.bfloat16) for _ in range(ROT)]
    Bs = [(torch.randn((Kc, N) if tb else (N, Kc), device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
    Os = [torch.zeros(M, N, device='cuda', dtype=torch.float32 if f32out else torch.bfloat16) for _ in range(ROT)]
    aux = [torch.randn(M, N, device='cuda').to(torch.bfloat16) for _ in range(ROT)] if epi else None
    bias = torch.zeros(N, device='cuda') if not f32out else None
    res = []
    for split in splits:
        for bm, ns in hipops._tile_candidates(ta, tb, M, N):
            i = [0]

            def run():
                k = i[0] % ROT
                i[0] += 1
                hipops._launch_gemm_bf16(As[k], Bs[k], Os[k], ta, tb, M, N, Kc, bias if split == 1 else None, epi if split == 1 else 0,
                                         aux[k] if aux else None, split, bm, ns, None)
            try:
                run()
            except RuntimeError:
                continue
            res.append((bench(run), hipops.tile_name(bm), ns & 0xFF, 8 if (ns & 0x100 or bm >= 256) else 4, split))
    i = [0]

    def vendor():
        k = i[0] % ROT
        i[0] += 1
        a = As[k].t() if ta else As[k]
        b = Bs[k] if tb else Bs[k].t()
        torch.mm(a, b, out=Os[k]) if not f32out else torch.mm(a, b)
    tv = bench(vendor)
    res.sort()
    fl = 2.0 * M * N * Kc
    best = res[0]
    print('t%d%d M=%5d N=%5d K=%5d epi=%d | best %-8s s%d w%d k%d %7.1f us %6.0f TF | vendor %7.1f us %6.0f TF | ratio %.2f | next: %s' % (
        ta, tb, M, N, Kc, epi, best[1], best[2], best[3], best[4], best[0], fl / best[0] / 1e6, tv, fl / tv / 1e6, best[0] / tv,
        '  '.join('%s/s%d/w%d/k%d %.1f' % (r[1], r[2], r[3], r[4], r[0]) for r in res[1:6])), flush=True)


if which in ('fwd', 'all'):
    for M, N, K, epi in FWD:
        sweep(False, False, M, N, K, epi)
if which in ('dgrad', 'all'):
    for M, N, K, epi in DGRAD:
        sweep(False, True, M, N, K, epi)
if which in ('wgrad', 'all'):
    for M, N, K in WGRAD:
        sweep(True, True, M, N, K, 0, f32out=True, splits=(1, 2, 4))
