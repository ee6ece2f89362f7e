"""A/B of libgoat_hip.so builds on fixed (shape, tile config) cases: every library named on the command line is timed
in its own subprocess (GOAT_HIP_LIB), cases interleaved over several rounds; operands rotate through 6 buffers.

    python scripts/gemm_ab.py libA.so libB.so ...          (driver)
"""
import json
import os
import subprocess
import sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)

def T(bm, bn):
    return bm | (bn << 16)


CASES = [  # ta, tb, M, N, K, epi, f32out, split, tile (rows | cols << 16), nstage (| 0x100: eight waves on 128x128)
    (0, 0, 3840, 2304, 768, 0, 0, 1, T(192, 192), 2), (0, 0, 3840, 2304, 768, 0, 0, 1, T(192, 192), 3), (0, 0, 3840, 1536, 768, 0, 0, 1, T(192, 192), 3),
    (0, 0, 3840, 1536, 768, 0, 0, 1, T(256, 192), 2), (0, 0, 3840, 1536, 768, 0, 0, 1, 128, 0x102),
    (0, 0, 3840, 768, 3072, 0, 0, 1, 96, 2), (0, 0, 3840, 768, 3072, 0, 0, 1, 96, 3), (0, 0, 3840, 768, 3072, 0, 0, 1, 96, 4),
    (0, 0, 3840, 768, 768, 0, 0, 1, 96, 3), (0, 0, 3840, 768, 768, 0, 0, 1, 96, 4), (0, 1, 3840, 768, 3072, 0, 0, 1, 96, 4),
    (0, 1, 3840, 768, 768, 0, 0, 1, 96, 4), (0, 1, 3840, 768, 768, 0, 0, 1, 128, 0x104), (0, 1, 3840, 768, 2304, 0, 0, 1, 96, 4),
    (0, 1, 3840, 768, 2304, 0, 0, 1, 128, 0x104), (0, 0, 8640, 768, 768, 0, 0, 1, 96, 4), (0, 0, 8640, 768, 768, 0, 0, 1, 128, 0x104),
    (0, 0, 3840, 3072, 768, 0, 0, 1, T(192, 256), 2), (0, 0, 3840, 3072, 768, 1, 0, 1, T(192, 256), 2), (0, 0, 3840, 3072, 768, 0, 0, 1, T(256, 256), 2),
    (0, 0, 3840, 3072, 768, 0, 0, 1, 128, 0x102), (0, 0, 3840, 2304, 768, 0, 0, 1, T(256, 192), 2), (0, 0, 3840, 2304, 768, 0, 0, 1, 128, 0x102),
    (0, 0, 3840, 768, 3072, 0, 0, 1, 128, 0x104), (0, 0, 3840, 768, 3072, 0, 0, 1, T(128, 256), 3), (0, 0, 3840, 768, 768, 0, 0, 1, 128, 0x104),
    (0, 0, 3840, 768, 768, 0, 0, 1, 64, 2),
    (0, 1, 3840, 3072, 768, 3, 0, 1, T(192, 256), 2), (0, 1, 3840, 3072, 768, 3, 0, 1, 128, 0x102), (0, 1, 3840, 768, 3072, 0, 0, 1, 128, 0x104),
    (0, 1, 3840, 768, 3072, 0, 0, 1, T(128, 256), 3),
    (1, 1, 3072, 768, 3840, 0, 1, 1, 128, 0x104), (1, 1, 3072, 768, 3840, 0, 1, 1, T(256, 256), 2), (1, 1, 3072, 768, 3840, 0, 1, 1, T(128, 256), 3),
    (1, 1, 768, 768, 3840, 0, 1, 4, 128, 0x103),
    (0, 0, 8640, 3072, 768, 0, 0, 1, T(256, 256), 2), (0, 0, 8640, 3072, 768, 0, 0, 1, 128, 0x102), (0, 0, 8640, 768, 3072, 0, 0, 1, T(128, 256), 3),
    (0, 0, 1776, 768, 768, 0, 0, 1, 64, 4), (0, 0, 8192, 8192, 8192, 0, 0, 1, T(256, 256), 2), (0, 0, 8192, 8192, 8192, 0, 0, 1, T(128, 256), 3),
]


def worker():
    import torch
    from vln_goat_amd import hipops, _lib
    torch.cuda.set_device(0)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    ROT = 6
    res = []
    for ta, tb, M, N, K, epi, f32, split, bm, ns in CASES:
        a = [torch.randn((K, M) if ta else (M, K), device='cuda').to(torch.bfloat16) for _ in range(ROT)]
        b = [(torch.randn((K, N) if tb else (N, K), device='cuda') * 0.05).to(torch.bfloat16) for _ in range(ROT)]
        o = [torch.zeros(M, N, device='cuda', dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(ROT)]
        aux = torch.randn(M, N, device='cuda').to(torch.bfloat16) if epi else None
        n = 8 if M * N * K > 1e11 else 48

        def run(k):
            rc = L.goat_gemm_bf16(st, ta, tb, hipops._dt(o[k]), a[k].data_ptr(), a[k].stride(0), b[k].data_ptr(), b[k].stride(0),
                                  o[k].data_ptr(), N, M, N, K, None, epi, aux.data_ptr() if epi else None, N if epi else 0,
                                  split, bm, ns, None)
            assert rc == 0, rc
        for k in range(ROT):
            run(k)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                run(i % ROT)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        res.append(best)
    print('RESULT ' + json.dumps(res))


if __name__ == '__main__':
    if os.environ.get('GOAT_AB_WORKER'):
        worker()
        sys.exit(0)
    libs = sys.argv[1:]
    out = {l: [] for l in libs}
    for rnd in range(2):
        for l in libs:
            env = dict(os.environ, GOAT_AB_WORKER='1', GOAT_HIP_LIB=os.path.abspath(l))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            line = [x for x in r.stdout.splitlines() if x.startswith('RESULT ')]
            if not line:
                print(l, 'FAILED', r.stdout[-600:])
                continue
            out[l].append(json.loads(line[0][7:]))
    print('%-46s' % 'case (ta,tb,M,N,K,epi,f32,split,bm,ns)' + ''.join('%14s' % os.path.basename(l)[:13] for l in libs))
    for i, c in enumerate(CASES):
        row = '%-46s' % str(c)
        for l in libs:
            ts = [r[i] for r in out[l]]
            row += '%14.2f' % min(ts) if ts else '%14s' % 'n/a'
        print(row)
