"""goat_wgrad_grouped on the weight-gradient problems of GOAT's layers (16 problems per launch as hipops.WgradQueue issues them):
every tile configuration, cold operands.    python scripts/wgrad_group_bench.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib

torch.cuda.set_device(0)
T = hipops.tile
LAYER = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]          # (n_out, n_in): QKV, attention output, FFN up, FFN down
GROUPS = {'text x4 layers (rows 3840)': [(3840, o, i) for _ in range(4) for (o, i) in LAYER],
          'pano x2 layers (rows 8640) + text x2': [(8640, o, i) for _ in range(2) for (o, i) in LAYER] + [(3840, o, i) for _ in range(2) for (o, i) in LAYER],
          'cross-modal (rows 1776 / 1056) x16': [(r, o, i) for r in (1776, 1056) for _ in range(2) for (o, i) in LAYER]}
CFGS = [(64, 3), (128, 2), (128, 0x102), (128, 0x103), (128, 0x104), (256, 2), (256, 3), (T(128, 256), 2), (T(128, 256), 3), (T(256, 256), 2),
        (256, 0x202), (T(128, 256), 0x202), (T(256, 256), 0x202)]          # 0x200: ping-pong main loop
GROUPS['text x2 layers (rows 3840), 8 problems'] = [(3840, o, i) for _ in range(2) for (o, i) in LAYER]
if os.environ.get('WG_GROUP'):        # (PMC passes: one group, one configuration)
    k = list(GROUPS)[int(os.environ['WG_GROUP'])]
    GROUPS = {k: GROUPS[k]}
if os.environ.get('WG_CFG'):
    a, b = os.environ['WG_CFG'].split(',')
    CFGS = [(int(a, 0), int(b, 0))]
L = _lib.lib()
ROT = int(os.environ.get('WG_ROT', '3'))      # operand sets rotated through (1: operands stay cache-resident where they fit)
for gname, probs in GROUPS.items():
    fl = sum(2.0 * r * o * i for r, o, i in probs)
    sets = []
    for _ in range(ROT):
        arr = (_lib.WgradProblem * len(probs))()
        keep = []
        for k, (rows, n_out, n_in) in enumerate(probs):
            dy = (torch.randn(rows, n_out, device='cuda') * 0.1).to(torch.bfloat16)
            x = torch.randn(rows, n_in, device='cuda').to(torch.bfloat16)
            dw = torch.empty(n_out, n_in, device='cuda')
            db = torch.zeros(n_out, device='cuda')
            q = arr[k]
            q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), n_out, x.data_ptr(), n_in, dw.data_ptr(), n_in, db.data_ptr()
            q.rows, q.n_out, q.n_in, q.accumulate = rows, n_out, n_in, 0
            keep.append((dy, x, dw, db))
        sets.append((arr, keep))
    res = []
    for bm, ns in CFGS:
        st = torch.cuda.current_stream().cuda_stream
        i = [0]

        def run():
            arr = sets[i[0] % ROT][0]
            i[0] += 1
            rc = L.goat_wgrad_grouped(st, ctypes.addressof(arr), len(probs), bm, ns)
            assert rc == 0, rc
        try:
            run()
        except AssertionError:
            continue
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        rows_, cols_ = bm & 0xFFFF, (bm >> 16) or 128
        tiles = sum(((o + rows_ - 1) // rows_) * ((i_ + cols_ - 1) // cols_) for _, o, i_ in probs)
        res.append((us, hipops.tile_name(bm), ns, tiles))
    for bm in (T(256, 256), T(128, 256), 256):            # contraction-balanced launch (goat_wgrad_grouped_balanced): one workgroup per CU
        nb = L.goat_wgrad_balanced_ws_bytes(bm)
        ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        i = [0]

        def run():
            arr = sets[i[0] % ROT][0]
            i[0] += 1
            rc = L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), len(probs), bm, ws.data_ptr(), nb)
            assert rc == 0, rc
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) * 1e3 / 10, hipops.tile_name(bm), 0x402, 256))
    print('%s: %.1f GFLOP' % (gname, fl / 1e9))
    for us, name, ns, tiles in sorted(res):
        print('   %-8s s%d%s  tiles %4d  %7.1f us  %6.0f TF' % (name, ns & 0xFF, ' 8w' if ns & 0x100 else (' pp' if ns & 0x200 else (' bal' if ns & 0x400 else '   ')), tiles, us, fl / us / 1e6))
