"""Every tile candidate of goat_gemm_bf16 on the full-size GOAT shapes against torch (fp32 on the same bf16 operands)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops
from vln_goat_amd._lib import EPI_GELU, EPI_MUL_DGELU
torch.cuda.set_device(0)
bad = 0
for (ta, tb, M, N, K, epi) in ((0, 0, 8640, 3072, 768, EPI_GELU), (0, 0, 8640, 768, 3072, 0), (0, 0, 3840, 3072, 768, EPI_GELU), (0, 1, 8640, 3072, 768, EPI_MUL_DGELU),
                               (0, 0, 8640, 2304, 768, 0), (0, 1, 3840, 768, 3072, 0), (0, 0, 576, 50304, 768, 0)):
    g = torch.Generator(device='cuda').manual_seed(M + N)
    a = torch.randn((M, K), device='cuda', generator=g).to(torch.bfloat16)
    b = (torch.randn((K, N) if tb else (N, K), device='cuda', generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device='cuda', generator=g) * 0.1 if not epi == EPI_MUL_DGELU else None
    ref = a.float() @ (b.float() if tb else b.float().T) + (bias if bias is not None else 0)
    auxin = torch.randn(M, N, device='cuda', generator=g).to(torch.bfloat16) if epi == EPI_MUL_DGELU else None
    if epi == EPI_GELU:
        want = torch.nn.functional.gelu(ref)
    elif epi == EPI_MUL_DGELU:
        u = auxin.float().requires_grad_(True)
        torch.nn.functional.gelu(u).sum().backward()
        want = ref * u.grad
    else:
        want = ref
    f32out = N > 40000
    for bm, ns in hipops._tile_candidates(ta, tb, M, N):
        out = torch.full((M, N), float('nan'), device='cuda', dtype=torch.float32 if f32out else torch.bfloat16)
        aux = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16) if epi == EPI_GELU else auxin
        try:
            for _rep in range(4):          # (rare hazards / races: several launches, the last one is checked)
                hipops._launch_gemm_bf16(a, b, out, bool(ta), bool(tb), M, N, K, bias, epi, aux, 1, bm, ns, None)
        except RuntimeError as e:
            continue
        torch.cuda.synchronize()
        err = float((out.float() - want).abs().max() / want.abs().max())
        nan = int(torch.isnan(out.float()).sum())
        nana = int(torch.isnan(aux.float()).sum()) if epi == EPI_GELU else 0
        flag = 'BAD' if (nan or nana or not err < 2e-2) else 'ok'
        bad += flag == 'BAD'
        if flag == 'BAD' or '-v' in sys.argv:
            print('%s t%d%d %dx%dx%d epi=%d tile %s s%d%s: rel err %.3e nan %d aux-nan %d' % (flag, ta, tb, M, N, K, epi, hipops.tile_name(bm), ns & 0xFF, ' 8w' if ns & 0x100 else '', err, nan, nana))
            if nan:
                rows = torch.isnan(out.float()).any(1).nonzero().flatten()
                cols = torch.isnan(out.float()).any(0).nonzero().flatten()
                print('    nan rows %d..%d (%d), cols %d..%d (%d)' % (int(rows[0]), int(rows[-1]), rows.numel(), int(cols[0]), int(cols[-1]), cols.numel()))
print('BAD configurations:', bad)
