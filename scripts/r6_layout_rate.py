"""round 6: rate of the 256 x 256 ping-pong tile by operand layout at a weight-gradient-like shape (long contraction, >= 2 rounds of tiles):
TN (both operands through ds_read_b64_tr_b16: today's weight gradient), NN (A K-contiguous, B transposed: a weight gradient whose dY was
transposed beforehand), NT (neither).  Random operands, L2 / Infinity Cache flushed before every repetition."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vln_goat_amd import hipops as H

torch.manual_seed(0)
dev = 'cuda'
for (M, N, K) in [(8192, 4096, 3840), (6144, 6144, 3840), (8192, 4096, 8640)]:
    a_k = (torch.rand(M, K, device=dev) * 4 - 2).bfloat16()       # A, K-contiguous
    a_t = a_k.t().contiguous()                                     # A as [K, M]
    b_k = (torch.rand(N, K, device=dev) * 4 - 2).bfloat16()       # B, K-contiguous
    b_t = b_k.t().contiguous()                                     # B as [K, N]
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    ref = None
    for name, a, b, ta, tb in [('NT', a_k, b_k, False, False), ('NN', a_k, b_t, False, True), ('TN', a_t, b_t, True, True)]:
        for tl, ns in [(H.tile(256, 256), H.PINGPONG | 2), (H.tile(256, 256), 2), (256, 3)]:
            try:
                fn = lambda: H._launch_gemm_bf16(a, b, out, ta, tb, M, N, K, None, H.EPI_NONE, None, 1, tl, ns, None)
                fn(); torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max() / ref.abs().max())
                t = H._time_cfg(fn, reps=7)
                print('%5dx%5dx%5d %s %-8s %-4s %8.1f us %7.0f TF/s  (max rel diff to NT %.1e)' % (M, N, K, name, H.tile_name(tl), H.stage_name(ns), t * 1e3, 2.0 * M * N * K / t / 1e9, err), flush=True)
            except RuntimeError as e:
                print(name, H.tile_name(tl), H.stage_name(ns), 'failed', str(e)[:80])
