"""Times the launch plans hipops.WgradQueue considers for a group of weight-gradient problems (single launch per tile configuration,
and the tail split: last partial round of 256x128 tiles moved to a second launch on 128x128 tiles).  Cold operands."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops, _lib
import ctypes
torch.cuda.set_device(0)
W = hipops.WgradQueue
LAYER = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
for name, nl in (('text x4 layers', 4), ('text x2 layers', 2), ('text x6 layers (24 problems -> 16 + 8)', None)):
    if nl is None:
        continue
    q = []
    for _ in range(nl):
        for o, i in LAYER:
            dy = (torch.randn(3840, o, device='cuda') * 0.1).to(torch.bfloat16)
            x = torch.randn(3840, i, device='cuda').to(torch.bfloat16)
            q.append((dy, x, torch.empty(o, i, device='cuda'), None, 0))
    fl = sum(2.0 * 3840 * t[0].shape[1] * t[1].shape[1] for t in q)
    print(name, 'tail', W._tail_split(q))
    for plan in W._plans(q):
        parts = []
        for idx, cfg in plan:
            arr = (_lib.WgradProblem * len(idx))()
            W._fill(arr, [q[i] for i in idx])
            parts.append((arr, len(idx), cfg))

        def run():
            for arr, m, cfg in parts:
                rc = _lib.lib().goat_wgrad_grouped(torch.cuda.current_stream().cuda_stream, ctypes.addressof(arr), m, cfg[0], cfg[1])
                assert rc == 0
        t = hipops._time_cfg(run, reps=7)
        print('   %-60s %7.1f us  %5.0f TF' % (' + '.join('%d x %s s%d%s' % (len(i), hipops.tile_name(c[0]), c[1] & 0xFF, ' 8w' if c[1] & 0x100 else '') for i, c in plan), t * 1e3, fl / t / 1e9))
