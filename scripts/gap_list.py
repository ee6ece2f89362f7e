"""Every idle interval (no kernel running) of at least <min_us> inside the last <window_ms> of a rocprofv3 --kernel-trace of bench.py, with
the kernel that ended before it and the kernel that starts after it — which dependencies of the replayed step graph cost dispatch latency.
    python scripts/gap_list.py <trace dir> [window_ms=6.5] [min_us=4]"""
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 6.5e6
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 4e3
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in csv.DictReader(open(f)))
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - win]
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', 'aten:').replace('_ZN12_GLOBAL__N_1', '').replace('_ZN7goat_g2', '')[:58]
busy_until, last = rows[0][1], rows[0]
tot = 0
print('kernels %d, window %.2f ms' % (len(rows), win / 1e6))
for r in rows[1:]:
    if r[0] > busy_until:
        gap = r[0] - busy_until
        if gap >= min_ns:
            tot += gap
            print('%7.1f us | after %-58s (q%s) | before %-58s (q%s)' % (gap / 1e3, short(last[2]), last[3], short(r[2]), r[3]))
    if r[1] > busy_until:
        busy_until, last = r[1], r
print('sum of listed gaps: %.1f us' % (tot / 1e3))
