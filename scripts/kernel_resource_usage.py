"""Per-kernel register / LDS / scratch table of the HIP sources (no GPU needed): compiles every csrc/*.hip with the flags of
vln-goat_amd/_lib.build plus -Rpass-analysis=kernel-resource-usage into a scratch directory and writes the table to stdout.
    python scripts/kernel_resource_usage.py > profiles/roundN_kernel_resource_usage.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'vln-goat_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def remarks(src, tmp):
    out = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', os.path.join(tmp, os.path.basename(src) + '.o'),
                          '-Rpass-analysis=kernel-resource-usage'], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode:
        raise RuntimeError(out.stdout.decode(errors='replace')[-2000:])
    return out.stdout.decode(errors='replace')


def main():
    cached = sys.argv[1] if len(sys.argv) > 1 else None       # a directory of <name>.txt remark dumps made earlier
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    with tempfile.TemporaryDirectory() as tmp:
        if cached:
            texts = [open(os.path.join(cached, os.path.basename(s)[:-4] + '.txt'), errors='replace').read() for s in srcs]
        else:
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
                texts = list(ex.map(lambda s: remarks(s, tmp), srcs))
    rows = []
    for s, text in zip(srcs, texts):
        cur = None
        for line in text.splitlines():
            m = re.search(r'remark: Function Name: (\S+)', line)
            if m:
                cur = {'src': os.path.basename(s), 'name': m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass', line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    # (binutils 2.38's c++filt does not know DF16b, the __bf16 builtin: demangled as `half` (Dh) and renamed; builtins are no substitution candidates)
    dem = subprocess.run(['c++filt'], input='\n'.join(r['name'].replace('DF16b', 'Dh') for r in rows), capture_output=True, text=True).stdout.split('\n')
    dem = [re.sub(r'\bhalf\b', 'bf16', d) for d in dem]
    seen, out = set(), []
    for r, d in zip(rows, dem):
        d = d.replace('(anonymous namespace)::', '')
        d = re.sub(r'^void ', '', d)
        depth, cut = 0, len(d)          # drop the trailing parameter list
        if d.endswith(')'):
            for i in range(len(d) - 1, -1, -1):
                depth += d[i] == ')'
                depth -= d[i] == '('
                if depth == 0:
                    cut = i
                    break
        r['dem'] = d[:cut].strip()
        if (r['src'], r['dem']) not in seen:
            seen.add((r['src'], r['dem']))
            out.append(r)
    spill = [r for r in out if r.get('VGPRs Spill', '0') != '0' or r.get('ScratchSize [bytes/lane]', '0') != '0']
    print('Per-kernel register / LDS / scratch usage (hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage: the flags of')
    print('vln-goat_amd/_lib.build; scripts/kernel_resource_usage.py).  LDS is the STATIC size: kernels with extern __shared__ get the rest at launch.')
    print('occ = waves per SIMD the register budget allows (512 VGPR+AGPR per SIMD lane: 1 wave at > 256, 2 at <= 256, 4 at <= 128, 8 at <= 64).')
    print('%d kernels, %d with scratch or VGPR spills%s' % (len(out), len(spill), (': ' + '; '.join('%s (%s B/lane)' % (r['dem'][:70], r.get('ScratchSize [bytes/lane]')) for r in spill)) if spill else ''))
    print()
    print('%-15s %5s %5s %5s %7s %4s %8s  %s' % ('source', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'occ', 'LDS B', 'kernel'))
    for r in out:
        print('%-15s %5s %5s %5s %7s %4s %8s  %s' % (r['src'], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'), r.get('ScratchSize [bytes/lane]', '?'),
                                                     r.get('Occupancy [waves/SIMD]', '?'), r.get('LDS Size [bytes/block]', '?'), r['dem'][:160]))


if __name__ == '__main__':
    main()
