"""Where the contraction-balanced grouped launch (goat_wgrad_grouped_balanced) spends its time, against the one-workgroup-per-tile launch
on the same operands: groups with no cut tiles at all (A, B: what the three inlined tile copies and the loop cost), a uniform group with
a tail (C, D) and a group that mixes contraction lengths (E).      python scripts/wgrad_balanced_probe.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
L = _lib.lib()
T = hipops.tile
LAYER = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
GROUPS = {
    'A 256 whole tiles (3840 x 4096 x 4096)': [(3840, 4096, 4096)],
    'B 512 whole tiles (2 x A)': [(3840, 4096, 4096)] * 2,
    'C 432 tiles, two problems (A + 3840 x 4096 x 2816)': [(3840, 4096, 4096), (3840, 4096, 2816)],
    'D text x4 layers (432 tiles, 16 problems)': [(3840, o, i) for _ in range(4) for (o, i) in LAYER],
    'E pano x2 + text x2': [(8640, o, i) for _ in range(2) for (o, i) in LAYER] + [(3840, o, i) for _ in range(2) for (o, i) in LAYER],
}
bm = T(256, 256)
nb = L.goat_wgrad_balanced_ws_bytes(bm)
ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for name, probs in GROUPS.items():
    sets = []
    for _ in range(3):
        arr = (_lib.WgradProblem * len(probs))()
        keep = []
        for k, (rows, n_out, n_in) in enumerate(probs):
            dy = (torch.randn(rows, n_out, device='cuda') * 0.1).to(torch.bfloat16)
            x = torch.randn(rows, n_in, device='cuda').to(torch.bfloat16)
            dw = torch.empty(n_out, n_in, device='cuda')
            q = arr[k]
            q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), n_out, x.data_ptr(), n_in, dw.data_ptr(), n_in, None
            q.rows, q.n_out, q.n_in, q.accumulate = rows, n_out, n_in, 0
            keep.append((dy, x, dw))
        sets.append((arr, keep))
    fl = sum(2.0 * r * o * i for r, o, i in probs)
    out = []
    for what in ('pp', 'bal'):
        i = [0]

        def run():
            arr = sets[i[0] % 3][0]
            i[0] += 1
            if what == 'pp':
                rc = L.goat_wgrad_grouped(st, ctypes.addressof(arr), len(probs), bm, 0x202)
            else:
                rc = L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), len(probs), bm, ws.data_ptr(), nb)
            assert rc == 0, rc
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(12):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 12)
    print('%-52s per-tile %7.1f us %5.0f TF   balanced %7.1f us %5.0f TF   %+5.1f %%' % (name, out[0], fl / out[0] / 1e6, out[1], fl / out[1] / 1e6, 100 * (out[1] / out[0] - 1)))
