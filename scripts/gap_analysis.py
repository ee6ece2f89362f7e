"""How busy is the GPU inside the timed hipGraph replays?  Reads a rocprofv3 --kernel-trace CSV of bench.py and, over the
last `--steps` replays (the densest window of kernels), reports: wall time of the window, time with >= 1 kernel running
(busy), time with exactly 0 (gaps between dependent graph nodes), the concurrency histogram (parallel branches), and the
idle time attributed to the kernel that FOLLOWS each gap (which launches wait longest).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --no-cpu-baseline --no-roofline
    python scripts/gap_analysis.py <dir> [window_ms]
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
if not f:
    print('no kernel_trace.csv under', d)
    sys.exit(1)
rows = []
for r in csv.DictReader(open(f[0])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_end = max(r[1] for r in rows)
win_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6          # default: the last 150 ms (~20 replays)
t0 = t_end - win_ns
rows = [r for r in rows if r[0] >= t0]
print('kernels in window: %d   window %.2f ms' % (len(rows), win_ns / 1e6))

events = []
for s, e, n in rows:
    events.append((s, 1, n))
    events.append((e, -1, n))
events.sort(key=lambda x: (x[0], x[1]))
depth, last = 0, events[0][0]
hist = defaultdict(int)
gap_after = defaultdict(lambda: [0, 0])
for t, dlt, n in events:
    hist[depth] += t - last
    if depth == 0 and dlt == 1 and t > last:
        ga = gap_after[n.split('(')[0][:70]]
        ga[0] += 1
        ga[1] += t - last
    depth += dlt
    last = t
wall = events[-1][0] - events[0][0]
ksum = sum(e - s for s, e, _ in rows)
print('wall %.2f ms   kernel-time sum %.2f ms   busy (>=1 kernel) %.2f ms (%.1f %%)   idle %.2f ms (%.1f %%)' % (
    wall / 1e6, ksum / 1e6, (wall - hist[0]) / 1e6, 100 * (wall - hist[0]) / wall, hist[0] / 1e6, 100 * hist[0] / wall))
for k in sorted(hist):
    print('  %d kernels running: %8.2f ms  %5.1f %%' % (k, hist[k] / 1e6, 100 * hist[k] / wall))
print('idle time by the kernel that ends the gap (top 15):')
for n, (c, t) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:15]:
    print('  %-70s gaps %5d   total %8.1f us   avg %5.2f us' % (n, c, t / 1e3, t / 1e3 / c))
