"""GPU-side time of goat_attn_fwd / goat_attn_bwd on the GOAT shapes: the C ABI called directly, 100 launches between two HIP
events (no autograd / allocator overhead), operands rotated through 4 buffer sets.   python scripts/attn_kernel_bench.py"""
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
ROT, NH, H = 4, 12, 768
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=48):
    """GPU time per launch: n launches captured into one hipGraph (no host gaps), replayed 5 times."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    global st
    keep = st
    g = torch.cuda.CUDAGraph()
    with _goat_graph(g):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            fn()
    st = keep
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for name, B, Lq, Lk, self_attn in (('text self', 48, 80, 80, True), ('pano self', 240, 36, 36, True), ('gmap<-text', 48, 22, 80, False),
                                   ('vp<-text', 48, 37, 80, False), ('text<-gmap', 48, 80, 22, False), ('text<-vp', 48, 80, 37, False),
                                   ('text160 self', 32, 160, 160, True)):
    for p, use_mask in ((0.1, True), (0.0, True)):
        sets = []
        for _ in range(ROT):
            if self_attn:
                qkv = torch.randn(B, Lq, 3 * H, device='cuda').to(torch.bfloat16)
                q, k, v = (qkv, 0, 3 * H, Lq * 3 * H), (qkv, H, 3 * H, Lq * 3 * H), (qkv, 2 * H, 3 * H, Lq * 3 * H)
                dqkv = torch.empty_like(qkv)
                dq, dk, dv = (dqkv, 0, 3 * H, Lq * 3 * H), (dqkv, H, 3 * H, Lq * 3 * H), (dqkv, 2 * H, 3 * H, Lq * 3 * H)
            else:
                qq = torch.randn(B, Lq, H, device='cuda').to(torch.bfloat16)
                kv = torch.randn(B, Lk, 2 * H, device='cuda').to(torch.bfloat16)
                q, k, v = (qq, 0, H, Lq * H), (kv, 0, 2 * H, Lk * 2 * H), (kv, H, 2 * H, Lk * 2 * H)
                dqq, dkv = torch.empty_like(qq), torch.empty_like(kv)
                dq, dk, dv = (dqq, 0, H, Lq * H), (dkv, 0, 2 * H, Lk * 2 * H), (dkv, H, 2 * H, Lk * 2 * H)
            o = torch.empty(B, Lq, H, device='cuda', dtype=torch.bfloat16)
            do = torch.randn(B, Lq, H, device='cuda').to(torch.bfloat16)
            lse = torch.empty(B * NH * Lq, device='cuda')
            km = torch.zeros(B, Lk, device='cuda') if use_mask else None
            sets.append((q, k, v, o, do, dq, dk, dv, lse, km))
        ptr = lambda t: t[0].data_ptr() + t[1] * 2
        i = [0]

        def fwd():
            q, k, v, o, do, dq, dk, dv, lse, km = sets[i[0] % ROT]
            i[0] += 1
            rc = L.goat_attn_fwd(st, 1, ptr(q), q[2], q[3], ptr(k), k[2], k[3], ptr(v), v[2], v[3], o.data_ptr(), H, Lq * H,
                                 km.data_ptr() if km is not None else None, None, lse.data_ptr(), B, NH, Lq, Lk, 0.125, p, 1, 0, None)
            assert rc == 0, rc

        def bwd():
            q, k, v, o, do, dq, dk, dv, lse, km = sets[i[0] % ROT]
            i[0] += 1
            rc = L.goat_attn_bwd(st, 1, ptr(q), q[2], q[3], ptr(k), k[2], k[3], ptr(v), v[2], v[3], o.data_ptr(), H, Lq * H,
                                 do.data_ptr(), H, Lq * H, ptr(dq), dq[2], dq[3], ptr(dk), dk[2], dk[3], ptr(dv), dv[2], dv[3],
                                 km.data_ptr() if km is not None else None, None, lse.data_ptr(), None, B, NH, Lq, Lk, 0.125, p, 1, 0, None)
            assert rc == 0, rc
        for _ in range(ROT):
            fwd()
        tf, tb = timeit(fwd), timeit(bwd)
        mb_f = (B * (Lq + 2 * Lk) * H * 2 + B * Lq * H * 2) / 1e6
        mb_b = (B * (Lq + 2 * Lk) * H * 2 * 2 + 2 * B * Lq * H * 2) / 1e6
        print('%-12s B=%3d Lq=%3d Lk=%3d p=%.1f mask=%d | fwd %5.1f us (%.1f MB -> %.2f TB/s) | bwd %5.1f us (%.1f MB -> %.2f TB/s)' % (
            name, B, Lq, Lk, p, use_mask, tf, mb_f, mb_f / tf, tb, mb_b, mb_b / tb), flush=True)
