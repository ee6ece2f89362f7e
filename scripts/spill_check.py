"""Register-spill audit of every kernel of the library: compiles each csrc/*.hip for gfx950 with
-Rpass-analysis=kernel-resource-usage and lists the kernels with spilled VGPRs/SGPR->memory or scratch.  No GPU needed.
    python scripts/spill_check.py [> profiles/roundN_gemm_spill_check.txt]"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(src):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-I', os.path.join(ROOT, 'include'),
                            '-o', os.path.join(td, 'o.o'), '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, cwd=td)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    rows = []
    for b in blocks:
        f = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        rows.append((b.split()[0], f('VGPRs'), f('AGPRs'), f('VGPRs Spill'), f(r'ScratchSize \[bytes/lane\]'), f(r'LDS Size \[bytes/block\]')))
    return os.path.basename(src), r.returncode, rows


def main():
    srcs = sorted(glob.glob(os.path.join(ROOT, 'vln-goat_amd', 'csrc', '*.hip')))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(audit, srcs))
    bad = 0
    for name, rc, rows in res:
        sp = [r for r in rows if r[3] or r[4]]
        print('%-16s rc %d  kernels %3d  max VGPRs %3d  spilling/scratch %d' % (name, rc, len(rows), max([r[1] for r in rows] or [0]), len(sp)))
        for r in sp:
            out = subprocess.run(['c++filt', r[0]], capture_output=True, text=True).stdout.strip()
            print('    %s  VGPRs %d  spilled %d  scratch %d B/lane' % (out[:150], r[1], r[3], r[4]))
        bad += len(sp)
    print('total kernels with spills or scratch: %d' % bad)
    return 0


if __name__ == '__main__':
    sys.exit(main())
