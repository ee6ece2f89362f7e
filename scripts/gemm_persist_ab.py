"""A/B of the persistent ping-pong GEMM (nstage | GOAT_GEMM_PERSIST) against one workgroup per tile, same box, same process.
Every line: 48 dependent launches in one hipGraph, operands rotating through enough buffer sets to be HBM-cold (> 400 MB), random
bf16 data (DVFS: constant operands clock ~30 % higher on the dense tiles).   python scripts/gemm_persist_ab.py [B48|B256|all]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import _lib, hipops


def _goat_graph(g, **kw):
    """torch.cuda.graph through vln_goat_amd.hipops.graph: a graph whose capture forked one of the package's parallel branches is kept
    alive (ROCm 7.2 graph-destruction bug; see hipops.graph)."""
    from vln_goat_amd import hipops
    return hipops.graph(g, **kw)

torch.cuda.set_device(0)
L = _lib.lib()
PP, PERSIST = 0x200, 0x400
which = sys.argv[1] if len(sys.argv) > 1 else 'all'


def tile(r, c):
    return r | (c << 16)


def graph_time(fn, n=48):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


SHAPES = []
if which in ('B48', 'all'):
    SHAPES += [('pano FFN-up   B48', 8640, 3072, 768, 1), ('pano QKV      B48', 8640, 2304, 768, 0), ('pano FFN-up dgrad B48', 8640, 3072, 768, 3),
               ('text FFN-up   B48', 3840, 3072, 768, 1)]
if which in ('B256', 'all'):
    SHAPES += [('text FFN-up   B256', 20480, 3072, 768, 1), ('text FFN-up dgrad B256', 20480, 3072, 768, 3), ('text QKV      B256', 20480, 2304, 768, 0),
               ('text out-proj B256', 20480, 768, 768, 0), ('text FFN-down B256', 20480, 768, 3072, 0), ('pano FFN-up   B256', 46080, 3072, 768, 1),
               ('pano QKV      B256', 46080, 2304, 768, 0), ('pano FFN-down B256', 46080, 768, 3072, 0)]
for name, M, N, K, epi in SHAPES:
    setb = (M * K + N * K + M * N * (2 if epi else 1)) * 2
    rot = max(2, min(48, int(420e6 / setb) + 1))
    A = [torch.randn(M, K, device='cuda').to(torch.bfloat16) for _ in range(rot)]
    W = [(torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16) for _ in range(rot)]
    C = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(rot)]
    X = [torch.randn(M, N, device='cuda').to(torch.bfloat16) for _ in range(rot)] if epi else None
    bias = torch.zeros(N, device='cuda') if epi in (0, 1) else None
    print('%s  %d x %d x %d  epi %d  (%d buffer sets)' % (name, M, N, K, epi, rot), flush=True)
    for t in (tile(256, 256), tile(192, 256), tile(128, 256), tile(256, 128), tile(128, 128)):
        rows, cols = t & 0xFFFF, t >> 16
        ntiles = ((M + rows - 1) // rows) * ((N + cols - 1) // cols)
        res = []
        for ns in (PP | 2, PP | PERSIST | 2):
            def fn(i, ns=ns):
                j = i % rot
                hipops._launch_gemm_bf16(A[j], W[j], C[j], False, False, M, N, K, bias, epi, X[j] if epi else None, 1, t, ns, None)
            try:
                res.append(graph_time(fn))
            except RuntimeError as e:
                res.append(float('nan'))
        tf = 2.0 * M * N * K / 1e6
        print('   %-8s %5d tiles (%.2f per CU)   one WG per tile %7.2f us %5.0f TF/s | persistent %7.2f us %5.0f TF/s   %+5.1f %%' % (
            hipops.tile_name(t), ntiles, ntiles / 256.0, res[0], tf / res[0], res[1], tf / res[1], 100.0 * (res[1] / res[0] - 1.0)), flush=True)
    del A, W, C, X
    torch.cuda.empty_cache()
