"""Attention kernel timings on the GOAT shapes (HIP events; operands rotated through 6 buffers = cold caches)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops

torch.cuda.set_device(0)
ROT = 6
def bench(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for name, B, Lq, Lk, self_attn in (('text self', 48, 80, 80, True), ('pano self', 240, 36, 36, True), ('gmap<-text', 48, 22, 80, False),
                                   ('vp<-text', 48, 37, 80, False), ('text<-gmap', 48, 80, 22, False), ('text<-vp', 48, 80, 37, False),
                                   ('text self B=96', 96, 80, 80, True)):
    H = 768
    if self_attn:
        xs = [torch.randn(B, Lq, 3 * H, device='cuda').to(torch.bfloat16).requires_grad_(True) for _ in range(ROT)]
        args = [(x, None) for x in xs]
    else:
        qs = [torch.randn(B, Lq, H, device='cuda').to(torch.bfloat16).requires_grad_(True) for _ in range(ROT)]
        kvs = [torch.randn(B, Lk, 2 * H, device='cuda').to(torch.bfloat16).requires_grad_(True) for _ in range(ROT)]
        args = list(zip(qs, kvs))
    km = torch.zeros(B, Lk, device='cuda')
    i = [0]
    def fwd():
        a, b = args[i[0] % ROT]; i[0] += 1
        with torch.no_grad():
            return hipops.attention(a, b, km, None, 12, 0.1)
    tf = bench(fwd)
    outs = [hipops.attention(a, b, km, None, 12, 0.1) for a, b in args]
    dys = [torch.randn_like(o) for o in outs]
    j = [0]
    def bwd():
        k = j[0] % ROT; j[0] += 1
        a, b = args[k]
        a.grad = None
        if b is not None:
            b.grad = None
        outs[k].backward(dys[k], retain_graph=True)
    tb = bench(bwd)
    byt = (B * Lq * H * 2 + B * Lk * H * 2) * 2 * 1
    print('%-16s B=%3d Lq=%3d Lk=%3d  fwd %6.1f us   bwd(dq+dkv, incl. autograd) %6.1f us   min-bytes fwd %.1f MB' % (name, B, Lq, Lk, tf, tb, byt / 1e6))
