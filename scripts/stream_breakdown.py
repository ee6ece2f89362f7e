"""Per-queue / per-stream kernel time of a rocprofv3 kernel trace of bench.py: which branch of the captured step is the long one.
    python scripts/stream_breakdown.py <trace dir>"""
import csv, glob, sys, collections, re
d = sys.argv[1]
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
print('columns:', list(rows[0].keys()))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
fam = lambda n: re.sub(r'<.*', '', re.sub(r'^void ', '', n)).replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:44]
tot = collections.Counter(); cnt = collections.Counter(); per = collections.defaultdict(collections.Counter)
for r in rows:
    dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r[key]
    tot[k] += dt; cnt[k] += 1; per[k][fam(r['Kernel_Name'])] += dt
for k, t in tot.most_common():
    print('%s %s: %9.1f us in %6d kernels' % (key, k, t, cnt[k]))
    for n, v in per[k].most_common(8):
        print('       %9.1f us  %s' % (v, n))
