"""Durations of the kernels matching <pattern> in the replayed steps of a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid size):
launches per step, average / min / max duration — which shapes of a family cost what inside the step (LayerNorm, attention ...).
    python scripts/kernel_hist.py <trace dir> <pattern> [window_ms=150] [ms_per_step=6.3]"""
import csv, glob, re, sys, collections
d, pat = sys.argv[1], sys.argv[2]
win = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
msps = float(sys.argv[4]) if len(sys.argv) > 4 else 6.3
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
t_end = max(int(r['End_Timestamp']) for r in rows)
agg = collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp']) < t_end - win * 1e6 or not re.search(pat, r['Kernel_Name']):
        continue
    grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))))
    name = re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1|void ', '', r['Kernel_Name'])[:60]
    agg[(name, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
n = win / msps
marks = sum(1 for r in rows if int(r['Start_Timestamp']) >= t_end - win * 1e6 and 'zero_ranges_kernel' in r['Kernel_Name'])
if marks >= 3:          # one goat_zero_ranges launch per captured step
    n = float(marks)
print('%-62s %8s %10s %9s %9s %9s %10s' % ('kernel', 'blocks', 'per step', 'avg us', 'min us', 'max us', 'us/step'))
for (name, grid), ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-62s %8d %10.1f %9.1f %9.1f %9.1f %10.1f' % (name, grid, len(ds) / n, sum(ds) / len(ds), min(ds), max(ds), sum(ds) / n))
