"""GPU-side time of goat_ln_fwd / goat_ln_bwd (C ABI, hipGraph of 40 launches, operands rotated) on the GOAT row counts."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import _lib


def _goat_graph(g, **kw):
    """torch.cuda.graph through vln_goat_amd.hipops.graph: a graph whose capture forked one of the package's parallel branches is kept
    alive (ROCm 7.2 graph-destruction bug; see hipops.graph)."""
    from vln_goat_amd import hipops
    return hipops.graph(g, **kw)
torch.cuda.set_device(0)
L = _lib.lib()
H, ROT = 768, 4
st_holder = [torch.cuda.current_stream().cuda_stream]


def timeit(fn, n=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = st_holder[0]
    with _goat_graph(g):
        st_holder[0] = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            fn()
    st_holder[0] = keep
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for M in (3840, 8640, 1776):
    for det in (0, 1, 2):        # 0 atomics, 1 per-call partials + reduction launch, 2 partials left behind (one batched reduction per backward pass)
        sets = []
        for _ in range(ROT):
            x, r = (torch.randn(M, H, device='cuda').to(torch.bfloat16) for _ in range(2))
            y, z, dy, dx, dres = (torch.empty(M, H, device='cuda', dtype=torch.bfloat16) for _ in range(5))
            dy.normal_()
            sets.append((x, r, y, z, dy, dx, dres, torch.empty(M, device='cuda'), torch.empty(M, device='cuda')))
        gamma, beta = torch.ones(H, device='cuda'), torch.zeros(H, device='cuda')
        dg, db = torch.zeros(H, device='cuda'), torch.zeros(H, device='cuda')
        ws = torch.empty(L.goat_ln_bwd_ws_floats(H), device='cuda') if det else None
        acc = 2 if det == 2 else 1
        i = [0]

        def fwd():
            x, r, y, z, dy, dx, dres, mean, rstd = sets[i[0] % ROT]; i[0] += 1
            rc = L.goat_ln_fwd(st_holder[0], 1, x.data_ptr(), r.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-12, 0.1, 1, 0, None,
                               y.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, H)
            assert rc == 0

        def bwd():
            x, r, y, z, dy, dx, dres, mean, rstd = sets[i[0] % ROT]; i[0] += 1
            rc = L.goat_ln_bwd(st_holder[0], 1, dy.data_ptr(), None, z.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0.1, 1, 0, None,
                               dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr() if ws is not None else None, M, H, acc, None)
            assert rc == 0
        for _ in range(ROT):
            fwd()
        tf, tb = timeit(fwd), timeit(bwd)
        print('M=%5d %s | fwd %5.1f us (%.1f MB -> %.2f TB/s) | bwd %5.1f us (%.1f MB -> %.2f TB/s)' % (
            M, ['atomics                          ', 'deterministic (partials + reduce)', 'partials only (deferred reduce)  '][det], tf, 4 * M * H * 2 / 1e6, 4 * M * H * 2 / 1e6 / tf,
            tb, 4 * M * H * 2 / 1e6, 4 * M * H * 2 / 1e6 / tb), flush=True)
