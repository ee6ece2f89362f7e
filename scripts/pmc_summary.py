"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel family, launches / total / average per launch."""
import csv, glob, re, sys, collections
d = sys.argv[1]
fam = lambda n: ('gemm2_kernel' if 'gemm2_' in n else 'pp_kernel (ping-pong GEMM)' if ('pp_kernel' in n or 'pp_group_kernel' in n) else 'gemm_nt_kernel' if 'gemm_nt_kernel' in n else
                 re.sub(r'<.*', '', re.sub(r'^void ', '', n)).replace('(anonymous namespace)::', '')[:60])
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[fam(r['Kernel_Name'])][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
rows = []
for k, cs in agg.items():
    for c, (n, s) in cs.items():
        rows.append((s, k, c, n))
rows.sort(reverse=True)
print('%-62s %-28s %8s %16s %14s' % ('kernel family', 'counter', 'launches', 'sum', 'avg/launch'))
for s, k, c, n in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%-62s %-28s %8d %16.1f %14.2f' % (k, c, n, s, s / n))
