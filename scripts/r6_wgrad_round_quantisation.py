"""round 6: is the grouped weight gradient's rate set by the ROUND QUANTISATION of its tiles over the 256 CUs?  Groups of n identical FFN-up problems
(dW [3072, 768] over 3840 rows: 36 tiles of 256 x 256 each) on the ping-pong 256 x 256 tile, cold operands: n = 7 / 14 / 21 fill 0.98 / 1.97 / 2.95
rounds, n = 12 is the text group's 432 tiles (1.69 rounds of work in 2), n = 10 / 17 sit just past a round boundary."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib
torch.cuda.set_device(0)
L = _lib.lib()
cfg = (hipops.tile(256, 256), hipops.PINGPONG | 2)
ROT = 3
BIAS = os.environ.get('WG_NO_BIAS', '0') != '1'
for rows in (3840, 20480):
    for n in ([int(v) for v in os.environ['WG_N'].split(',')] if 'WG_N' in os.environ else (7, 8, 10, 12, 14, 17, 21)):
        probs = [(rows, 3072, 768)] * n
        fl = sum(2.0 * r * o * i for r, o, i in probs)
        sets = []
        for _ in range(ROT):
            arr = (_lib.WgradProblem * n)()
            keep = []
            for k, (r, n_out, n_in) in enumerate(probs):
                dy = (torch.randn(r, n_out, device='cuda') * 0.1).to(torch.bfloat16)
                x = torch.randn(r, n_in, device='cuda').to(torch.bfloat16)
                dw = torch.empty(n_out, n_in, device='cuda')
                db = torch.zeros(n_out, device='cuda')
                q = arr[k]
                q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), n_out, x.data_ptr(), n_in, dw.data_ptr(), n_in, (db.data_ptr() if BIAS else None)
                q.rows, q.n_out, q.n_in, q.accumulate = r, n_out, n_in, 0
                keep.append((dy, x, dw, db))
            sets.append((arr, keep))
        i = [0]

        def run():
            arr = sets[i[0] % ROT][0]
            i[0] += 1
            assert L.goat_wgrad_grouped(torch.cuda.current_stream().cuda_stream, ctypes.addressof(arr), n, cfg[0], cfg[1]) == 0
        t = hipops._time_cfg(run, reps=9) * 1e-3
        tiles = n * 36
        print('rows %5d  %2d problems  %4d tiles = %.2f rounds of 256   %8.1f us   %6.0f TFLOP/s   (%.0f per FILLED round-equivalent)' % (
            rows, n, tiles, tiles / 256.0, t * 1e6, fl / t / 1e12, fl / t / 1e12 * (-(-tiles // 256)) / (tiles / 256.0)), flush=True)
        del sets
        torch.cuda.empty_cache()
